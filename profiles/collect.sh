#!/bin/bash
# Collect the round's profiles on the GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1'
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/<tag>_kernel_stats.csv
# 2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the MI355X guide prescribes)
#    -> gpurun_out/pmc_{fetch,write}/ ... summarised by profiles/pmc_summary.py into profiles/<tag>_pmc_traffic.json
set -e
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
# optional 2nd / 3rd argument: the command to profile instead of the default bench (e.g. "python $ROOT/profiles/shape_bench.py chi64")
# and "stats" to skip the two PMC passes
CMD=${2:-"python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline"}
CMD1=${2:-"python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"}
MODE=${3:-all}
export PROFILE_CMD="$CMD1"
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
cp $(ls $OUT/prof_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
if [ "$MODE" = "stats" ]; then ls -la $OUT/${TAG}_kernel_stats.csv; exit 0; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD1 > $OUT/pmc_write.log 2>&1
python $ROOT/profiles/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json
ls -la $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_pmc_traffic.json
