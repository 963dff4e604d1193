# L2<-fabric reads of the Gram inside bench.py's pipeline (224-CU stream) for pace configurations of the two-launch form.
# bash tools/probes/pace_pipeline_fetch.sh "0:228 1:222 1:223"   (ROMTIME_PIPELINE_GRAM_PACE:ROMTIME_GRAM_PACE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pace_fetch; mkdir -p $O; rm -f $O/*.log
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for PC in ${1:-0:228 1:222}; do
  P=${PC%%:*}; C=${PC##*:}
  ROMTIME_PIPELINE_GRAM_PACE=$P ROMTIME_GRAM_PACE=$C rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_${P}_$C -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-latency > $O/f_${P}_$C.log 2>&1 || exit 1
  python3 - <<PY >> $O/summary.log
import csv,glob,collections
f=glob.glob("$O/f_${P}_$C/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "gram128" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE": acc[row["Kernel_Name"][30:70]].append(float(row["Counter_Value"]))
tot=0
for k,v in acc.items():
    gb=2*1024*sum(v)/len(v)/1e9; tot+=gb
    print("pipeline pace=$P cfg=$C", k, "launches", len(v), "GB %.2f" % gb)
print("pipeline pace=$P cfg=$C total GB per Gram %.2f" % tot)
PY
  rm -rf $O/f_${P}_$C
done
cat $O/summary.log
