set -x
O=gpurun_out/r5l; mkdir -p gpurun_out/r5l
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" --tb=short 2>&1 | tail -8 > $O/kernels.log
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c2_20x20 or c3_layers or drift" --tb=short 2>&1 | tail -30 > $O/fullsize.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "identity_gate" --tb=short 2>&1 | tail -30 > $O/identity.log
cat $O/kernels.log $O/fullsize.log $O/identity.log
