cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pf_$C -- python3 tools/probes/proj_one.py > gpurun_out/pf_$C.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pf_$C/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "project_fused_kernel" in row["Kernel_Name"] and "false" in row["Kernel_Name"] and row["Counter_Name"]=="$C": acc[row["Kernel_Name"][:50]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print("$C", k, "KB per launch %.0f" % (sum(v)/len(v)), "launches", len(v))
PY
done
