python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3 or heavy or c1" --tb=short 2>&1 | tail -3
python -m pytest tests/test_gpu_toggles.py tests/test_gpu_parity.py -q -m gpu -x --tb=short 2>&1 | tail -3
bash profiles/r5_24.sh
