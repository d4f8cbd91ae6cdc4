#!/bin/bash
# Everything the round's profiles/ holds, in one go (about 15 minutes on the GPU box):
#   gpurun --timeout 2400 -- 'bash profiles/collect_all.sh r2'
# C2 = the default bench (20x20 chi = 32); chi64 / cubic16 = profiles/shape_bench.py (per-site shapes of BASELINE configs[4] / [3]).
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/profiles/collect.sh ${TAG} || true
bash $ROOT/profiles/collect_mfma.sh ${TAG} || true
for w in chi64 cubic16; do
  NREP=1 bash $ROOT/profiles/collect.sh ${TAG}_${w} "python $ROOT/profiles/shape_bench.py $w" || true
  NREP=1 bash $ROOT/profiles/collect_mfma.sh ${TAG}_${w} "python $ROOT/profiles/shape_bench.py $w" || true
done
ls -la $ROOT/gpurun_out/${TAG}*
