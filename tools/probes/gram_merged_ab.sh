# One-launch Gram (round 3) (ROMTIME_GRAM_FLAGS=17) against the default two-launch form: time back to back, HBM reads (FETCH_SIZE).
# bash tools/probes/gram_merged_ab.sh on the GPU box; output under gpurun_out/gram_ab/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gram_ab; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for F in 17 1; do
  echo "== ROMTIME_GRAM_FLAGS=$F (16 = one launch)" >> $O/time.log
  ROMTIME_GRAM_FLAGS=$F timeout -k 10 200 python3 $R/tools/probe_gram_sustained.py >> $O/time.log 2>&1 || exit 1
  ROMTIME_GRAM_FLAGS=$F timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$F -- python3 $R/tools/probes/gram_traffic_ab.py > $O/fetch_$F.log 2>&1 || exit 2
  python3 - <<PY >> $O/fetch.log
import csv,glob,collections
f=glob.glob("$O/fetch_$F/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "gram128" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE": acc[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print("flags $F", k, "launches", len(v), "GB per launch (2*FETCH_SIZE*1024): %.2f" % (2*1024*sum(v)/len(v)/1e9))
PY
  rm -rf $O/fetch_$F
done
cat $O/time.log $O/fetch.log
