NREP=10 python profiles/shape_bench.py heavyhex 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('heavyhex', d['ms_per_layer'], {k:(v['ms'],v['launches']) for k,v in d['classes'].items()})
"
python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('L7', d['ms_per_step'], d['kernel_classes']['jacobi'], d['config']['theta_svd_sweeps_per_gate'])"
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3 or heavy" --tb=short 2>&1 | tail -2
