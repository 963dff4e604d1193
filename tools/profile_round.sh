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
# the Gram of a single POD (whole chip: one launch, paced): latency mode of the same command
BL="python3 $R/bench.py --steps 3 --warmup 1 --mode latency --no-cpu-baseline --no-secondary --no-latency"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat_stats -- $BL > $O/lat_stats.log 2>&1 || exit 40
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lat_fetch -- $BL > $O/lat_fetch.log 2>&1 || exit 41
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lat_write -- $BL > $O/lat_write.log 2>&1 || exit 42
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/lat_sq -- $BL > $O/lat_sq.log 2>&1 || exit 44
python3 $R/tools/bench_configs.py > $DEST/${ROUND}_configs_${TAG}.jsonl 2> $O/configs.log || exit 6
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_stats -- python3 $R/tools/bench_configs.py c5sweep > $O/sweep.log 2>&1 || exit 7
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hsweep_stats -- python3 $R/tools/bench_configs.py c5h > $O/hsweep.log 2>&1 || exit 8
# counter passes of the other kernels DESIGN.md quotes: the direct sweep (project_fused, sweep_values, newton_solve), the
# hyper-reduced sweep (expansion GEMM, newton_solve) and config 4 (deim_*, project_fused at 120 vectors, tallskinny, rank_update)
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA"
n=20
for W in c5sweep c5h c4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${W}_fetch -- python3 $R/tools/bench_configs.py $W > $O/${W}_fetch.log 2>&1 || exit $n
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${W}_write -- python3 $R/tools/bench_configs.py $W > $O/${W}_write.log 2>&1 || exit $((n+1))
  rocprofv3 --pmc $SQ --output-format csv -d $O/${W}_sq -- python3 $R/tools/bench_configs.py $W > $O/${W}_sq.log 2>&1 || exit $((n+2))
  n=$((n+3))
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- python3 $R/tools/bench_configs.py c4 > $O/c4.log 2>&1 || exit 30
cd $R
python3 tools/summarize_prof.py ${ROUND}_bench_${TAG} --stats $O/stats --fetch $O/fetch --write $O/write --sq $O/sq || exit 9
python3 tools/summarize_prof.py ${ROUND}_benchlat_${TAG} --stats $O/lat_stats --fetch $O/lat_fetch --write $O/lat_write --sq $O/lat_sq || exit 43
KEEP=project,newton,sweep,gemm,deim,tallskinny,rank_update,skinny,gram,symeig,solve
for W in c5sweep c5h; do
  python3 tools/summarize_prof.py ${ROUND}_${W}_${TAG} --fetch $O/${W}_fetch --write $O/${W}_write --sq $O/${W}_sq --keep $KEEP || exit 31
done
python3 tools/summarize_prof.py ${ROUND}_c4_${TAG} --stats $O/c4_stats --fetch $O/c4_fetch --write $O/c4_write --sq $O/c4_sq --keep $KEEP || exit 32
python3 tools/summarize_prof.py ${ROUND}_sweep_${TAG} --stats $O/sweep_stats || exit 10
python3 tools/summarize_prof.py ${ROUND}_hsweep_${TAG} --stats $O/hsweep_stats || exit 11
cp profiles/${ROUND}_*_${TAG}* profiles/gram_traffic.json profiles/project_traffic.json $DEST/
rm -f $DEST/*_sweep_${TAG}_pmc.json $DEST/*_hsweep_${TAG}_pmc.json; rm -rf $O/lat_stats $O/lat_fetch $O/lat_write $O/lat_sq $O/stats $O/fetch $O/write $O/sq $O/sweep_stats $O/hsweep_stats $O/c4_stats $O/*_fetch $O/*_write $O/*_sq   # raw output is large; the summaries are what is kept
ls -la $DEST
