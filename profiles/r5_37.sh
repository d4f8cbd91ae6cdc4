python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fiber_gemm" --tb=short 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3 or heavy" --tb=short 2>&1 | tail -2
NREP=10 python profiles/shape_bench.py heavyhex | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('heavyhex', d['ms_per_layer'], d['classes'].get('phase_bp_update'), d['classes'].get('phase_gate_batch'))"
