python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "double_pair or bf16x3" --tb=short 2>&1 | tail -3
python profiles/plane_bench.py 100 5 2>&1 | grep gram2
