#!/bin/bash
# A/B of environment switches on the C2 bench: bash profiles/ab_env.sh OUT "VAR=1 VAR2=1" "..." ...   (first argument: output file)
out=$1; shift
: > $out
for envs in "$@"; do
  echo "== $envs" >> $out
  env $envs python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:(v['ms'],v['TFLOPs']) for k,v in d['kernel_classes'].items() if v['ms']>5})" >> $out
done
