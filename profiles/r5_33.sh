for w in 384 1024 128; do echo "min_wgs $w"; for L in 7 10; do TNQS_X3_MIN_WGS=$w python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' L', d['config']['workload'][:12], d['ms_per_step'], d['kernel_classes']['bp_pairgram'])"; done; done
