#!/bin/bash
# Only the latency-mode (whole-chip Gram) passes of tools/profile_round.sh:  bash tools/profile_latency_only.sh r03 v5
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; ROUND=${1:-r03}; TAG=${2:-v0}
O=$R/gpurun_out/proflat_${ROUND}_${TAG}; DEST=$R/gpurun_out/profiles_${ROUND}_${TAG}; mkdir -p $O $DEST
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
BL="python3 $R/bench.py --steps 3 --warmup 1 --mode latency --no-cpu-baseline --no-secondary --no-latency"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat_stats -- $BL > $O/lat_stats.log 2>&1 || exit 40
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lat_fetch -- $BL > $O/lat_fetch.log 2>&1 || exit 41
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lat_write -- $BL > $O/lat_write.log 2>&1 || exit 42
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/lat_sq -- $BL > $O/lat_sq.log 2>&1 || exit 44
cd $R
python3 tools/summarize_prof.py ${ROUND}_benchlat_${TAG} --stats $O/lat_stats --fetch $O/lat_fetch --write $O/lat_write --sq $O/lat_sq || exit 43
cp profiles/${ROUND}_benchlat_${TAG}* profiles/gram_traffic.json $DEST/
rm -rf $O
ls -la $DEST
