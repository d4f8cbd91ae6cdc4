#!/bin/bash
# MFMA-utilisation counters of a run (one rocprofv3 PMC pass, kernel trace only -- never combined with the API trace domains):
#   gpurun --timeout 1200 -- 'bash profiles/collect_mfma.sh r2 ["<command>"]'
# -> gpurun_out/<tag>_mfma_util.json (copy into profiles/ to commit).  Counters: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
# SQ_INSTS_VALU_MFMA_MOPS_F32 / _F64 / _BF16, SQ_WAVE_CYCLES (SQ block, 8 slots) and GRBM_GUI_ACTIVE (GRBM block).
set -e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${2:-"python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"}
OUT=$ROOT/gpurun_out
export PROFILE_CMD="$CMD"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1
python $ROOT/profiles/mfma_summary.py $OUT/pmc_mfma $OUT/${TAG}_mfma_util.json
