python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3" --tb=short 2>&1 | tail -12
