# Paced Gram (ROMTIME_GRAM_FLAGS & 32: workgroups of an XCD kept within a few stages of each other for L2 sharing) against
# the unpaced default: time back to back, L2<-fabric reads (FETCH_SIZE), correctness of G against torch.
# bash tools/probes/gram_pace_ab.sh "1:128 33:128 33:248" (flags:pace config) on the GPU box; output under gpurun_out/gram_pace/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gram_pace; mkdir -p $O; rm -f $O/*.log
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for FC in ${1:-1:128 33:128}; do
  F=${FC%%:*}; C=${FC##*:}
  export ROMTIME_GRAM_PACE=$C
  echo "== ROMTIME_GRAM_FLAGS=$F ROMTIME_GRAM_PACE=$C" >> $O/time.log
  ROMTIME_GRAM_FLAGS=$F timeout -k 10 200 python3 $R/tools/probe_gram_sustained.py >> $O/time.log 2>&1 || exit 1
  ROMTIME_GRAM_FLAGS=$F timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$F -- python3 $R/tools/probes/gram_traffic_ab.py > $O/fetch_$F.log 2>&1 || exit 2
  python3 - <<PY >> $O/fetch.log
import csv,glob,collections
f=glob.glob("$O/fetch_$F/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "gram128" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE": acc[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print("flags $F pace $C", k[30:], "GB per launch (2*FETCH_SIZE*1024): %.2f" % (2*1024*sum(v)/len(v)/1e9))
PY
  rm -rf $O/fetch_$F
done
grep -v amdgpu.ids $O/time.log; cat $O/fetch.log
