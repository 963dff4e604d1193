#!/bin/bash
# Produce the tracked profile set of a round on the GPU box:  bash tools/profile_round.sh r01 v7
# Output: gpurun_out/profiles_<round>_<tag>/  (copy into profiles/ afterwards).  Counter passes are separate
# rocprofv3 runs (one --pmc group each, no trace domains besides the kernel trace of the stats run).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=${1:-r01}; TAG=${2:-v0}
O=$R/gpurun_out/prof_${ROUND}_${TAG}
DEST=$R/gpurun_out/profiles_${ROUND}_${TAG}
mkdir -p $O $DEST
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-latency"
python3 $R/bench.py > $O/bench.log 2>&1 || exit 1
tail -1 $O/bench.log > $DEST/${ROUND}_bench_${TAG}.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-latency > $O/stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || exit 4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1 || exit 5
python3 $R/tools/bench_configs.py > $DEST/${ROUND}_configs_${TAG}.jsonl 2> $O/configs.log || exit 6
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_stats -- python3 $R/tools/bench_configs.py c5sweep > $O/sweep.log 2>&1 || exit 7
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hsweep_stats -- python3 $R/tools/bench_configs.py c5h > $O/hsweep.log 2>&1 || exit 8
cd $R
python3 tools/summarize_prof.py ${ROUND}_bench_${TAG} --stats $O/stats --fetch $O/fetch --write $O/write --sq $O/sq || exit 9
python3 tools/summarize_prof.py ${ROUND}_sweep_${TAG} --stats $O/sweep_stats || exit 10
python3 tools/summarize_prof.py ${ROUND}_hsweep_${TAG} --stats $O/hsweep_stats || exit 11
cp profiles/${ROUND}_*_${TAG}* profiles/gram_traffic.json $DEST/
rm -f $DEST/*_sweep_*_pmc.json $DEST/*_hsweep_*_pmc.json; rm -rf $O/stats $O/fetch $O/write $O/sq $O/sweep_stats $O/hsweep_stats   # raw output is large; the summaries are what is kept
ls -la $DEST
