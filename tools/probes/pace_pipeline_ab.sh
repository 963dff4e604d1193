# bench.py's pipeline headline with variants of the paced Gram, alternating, short runs.
# bash tools/probes/pace_pipeline_ab.sh "1:228 33:228" (ROMTIME_GRAM_FLAGS:ROMTIME_GRAM_PACE; flags & 32 = unpaced)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pace_ab; mkdir -p $O; rm -f $O/*.json
cd $R
for i in 1 2 3; do
  for FC in ${1:-1:228 33:228}; do
    F=${FC%%:*}; C=${FC##*:}
    ROMTIME_PIPELINE_GRAM_PACE=${PIPE_PACE:-0} ROMTIME_GRAM_PACE=$C ROMTIME_GRAM_FLAGS=$F timeout -k 10 120 python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-latency > $O/run_${F}_${C}_$i.json 2>/dev/null || exit 1
  done
done
python3 - <<PY
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$O/run_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split("/")[-1].rsplit("_",1)[0]].append((d["ms_per_step"], d["roofline"]["kernel_ms"]))
for k,v in acc.items():
    print(k, "ms_per_step", " ".join("%.3f" % a for a,_ in v), "| gram_ms", " ".join("%.3f" % b for _,b in v))
PY
