#!/bin/bash
# Counter passes + kernel trace of config 4 alone (part of tools/profile_round.sh):  bash tools/profile_c4.sh r03 v1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=${1:-r03}; TAG=${2:-v1}
O=$R/gpurun_out/prof_${ROUND}_${TAG}_c4
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_fetch -- python3 $R/tools/bench_configs.py c4 > $O/c4_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_write -- python3 $R/tools/bench_configs.py c4 > $O/c4_write.log 2>&1 || exit 2
rocprofv3 --pmc $SQ --output-format csv -d $O/c4_sq -- python3 $R/tools/bench_configs.py c4 > $O/c4_sq.log 2>&1 || exit 3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- python3 $R/tools/bench_configs.py c4 > $O/c4.log 2>&1 || exit 4
cd $R
python3 tools/summarize_prof.py ${ROUND}_c4_${TAG} --stats $O/c4_stats --fetch $O/c4_fetch --write $O/c4_write --sq $O/c4_sq \
  --keep project,newton,sweep,gemm,deim,tallskinny,rank_update,gram,symeig,solve > $O/summary.log || exit 5
mkdir -p $R/gpurun_out/profiles_${ROUND}_${TAG}
cp profiles/${ROUND}_c4_${TAG}* $R/gpurun_out/profiles_${ROUND}_${TAG}/
rm -rf $O/c4_fetch $O/c4_write $O/c4_sq $O/c4_stats
