R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/solve_trace; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/probes/solve_probe.py > $O/log 2>&1
cat $O/log | tail -2
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$O/t/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "newton" in r["Name"]: print(r["Name"][:50], r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
