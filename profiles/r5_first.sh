#!/bin/bash
# round 5, first call: where the round starts (headline), fork A/B at three lattice sizes, the two 8-GPU configurations at one rank's footprint
set -x
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/c2_default.json 2> $O/c2_default.err
for L in 20 14 7; do
  for F in default 0; do
    if [ $F = default ]; then env -u TNQS_FORK python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/fork_${F}_L$L.json 2>> $O/err.txt
    else TNQS_FORK=0 python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/fork_${F}_L$L.json 2>> $O/err.txt; fi
  done
done
python bench.py --config c4 --L 5 --steps 3 --warmup 1 > $O/c4_L5.json 2> $O/c4_L5.err
python bench.py --config c5 --L 11 --steps 3 --warmup 1 > $O/c5_L11.json 2> $O/c5_L11.err
for f in $O/*.json; do echo $f; python - <<PY
import json
try:
    d=json.load(open("$f")); print(d["ms_per_step"], d["config"].get("memory"), d["config"].get("bp_sweeps_per_step"), d["config"].get("theta_svd_sweeps_per_gate"))
except Exception as e: print("ERR", e)
PY
done
