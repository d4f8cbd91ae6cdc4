set -x
O=gpurun_out/r5k; mkdir -p gpurun_out/r5k
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" --tb=short 2>&1 | tail -8 > $O/kernels.log
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c2_20x20 or c3_layers" --tb=short 2>&1 | tail -40 > $O/fullsize.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_toggles.py tests/test_gpu_sharded.py -q -m gpu --tb=line 2>&1 | tail -25 > $O/parity.log
cat $O/kernels.log $O/fullsize.log $O/parity.log
