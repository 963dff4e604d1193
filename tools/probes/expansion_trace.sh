# device time of expansion_kernel and its ablations (rocprofv3 kernel trace of tools/probes/expansion_probe.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ex_trace; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/probes/expansion_probe.py > $O/log 2>&1
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$O/t/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "expansion" in r["Name"] or "gemm_f64" in r["Name"]: print(r["Name"][:70], r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
PY
