# Fused projection: XCD-contiguous work list (ROMTIME_PROJECT_MAP=1, default) against the round-robin one (0):
# kernel time from a kernel trace and L2<-fabric reads (FETCH_SIZE) for 32 x (N = 1e5, r = 80) and 120 vectors.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/proj_map; mkdir -p $O; rm -f $O/*.log
for M in 1 0; do
 for B in 32 120; do
  ROMTIME_PROJECT_MAP=$M rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${M}_$B -- python3 tools/probes/proj_one.py $B > $O/kt_${M}_$B.log 2>&1 || exit 1
  ROMTIME_PROJECT_MAP=$M rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${M}_$B -- python3 tools/probes/proj_one.py $B > $O/pmc_${M}_$B.log 2>&1 || exit 2
  python3 - <<PY >> $O/summary.log
import csv,glob
f=glob.glob("$O/kt_${M}_$B/*/*_kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    if "project_fused_kernel" in row["Name"] and "false" in row["Name"]: print("map $M B $B", row["Name"][:40], "avg us %.1f" % (float(row["AverageNs"])/1e3), "calls", row["Calls"])
f=glob.glob("$O/pmc_${M}_$B/*/*_counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "project_fused_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("map $M B $B FETCH MB per launch (2*FETCH_SIZE*1024): %.0f" % (2*1024*sum(v)/len(v)/1e6))
PY
  rm -rf $O/kt_${M}_$B $O/pmc_${M}_$B
 done
done
cat $O/summary.log
