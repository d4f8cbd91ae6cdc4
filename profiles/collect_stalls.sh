#!/bin/bash
# Stall / occupancy counters of the tensor-pass kernels (round-3 verdict, item 7: what bounds mfma_pair_kernel?):
#   gpurun --timeout 1500 -- 'bash profiles/collect_stalls.sh r4 ["<command>"]'
# Four separate rocprofv3 PMC passes (kernel trace only; SQ block: at most 8 counters per pass), summarised per kernel by profiles/pmc_generic.py
# into gpurun_out/<tag>_pair_stalls.json (copy into profiles/ to commit).
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${2:-"python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TNQS_BENCH_NOPROF=1
pass() { name=$1; shift; rm -rf $OUT/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed (see $OUT/pmc_$name.log)"; }
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass b SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass d TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCC_HIT TCC_MISS TCC_EA_RDREQ GRBM_GUI_ACTIVE
python $ROOT/profiles/pmc_generic.py $OUT/${TAG}_pair_stalls.json $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d
