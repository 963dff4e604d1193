# bench.py pipeline headline for CU partitions / Gram forms: "E:PACE:FLAGS" = eigensolver CUs per XCD, pipeline pacing, ROMTIME_GRAM_FLAGS
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/part_ab; mkdir -p $O; rm -f $O/*.json
cd $R
for i in 1 2 3; do
  for T in ${1:-4:0:1 3:1:1}; do
    E=${T%%:*}; REST=${T#*:}; P=${REST%%:*}; F=${REST##*:}
    ROMTIME_PIPELINE_EIG_CUS=$E ROMTIME_PIPELINE_GRAM_PACE=$P ROMTIME_GRAM_FLAGS=$F timeout -k 10 120 python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-latency > $O/run_${E}_${P}_${F}_$i.json 2>/dev/null || exit 1
  done
done
python3 - <<PY
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$O/run_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split("/")[-1].rsplit("_",1)[0]].append((d["ms_per_step"], d["roofline"]["kernel_ms"], d["stage_ms"]["eig_chain_ms"]))
for k,v in acc.items():
    print(k, "ms_per_step", " ".join("%.3f" % a for a,_,_ in v), "| gram_ms", " ".join("%.3f" % b for _,b,_ in v), "| eig_chain", " ".join("%.2f" % c for _,_,c in v))
PY
