python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or pair" --tb=short 2>&1 | tail -5
python -m pytest tests/test_gpu_toggles.py -q -m gpu -k "BF16X3" --tb=short 2>&1 | tail -5
