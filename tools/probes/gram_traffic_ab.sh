# HBM-side reads of the Gram kernel's two launches (rocprofv3 --pmc FETCH_SIZE) and its time.  bash tools/probes/gram_traffic_ab.sh on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/probes/gram_traffic_ab.py | tail -1
for F in 1; do
  ROMTIME_GRAM_FLAGS=$F rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/gram_fetch_$F -- python3 tools/probes/gram_traffic_ab.py > gpurun_out/gram_fetch_$F.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/gram_fetch_$F/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "gram128" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE": acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print("flags $F", k, "GB per launch (2*FETCH_SIZE*1024): %.2f" % (2*1024*sum(v)/len(v)/1e9))
PY
done
