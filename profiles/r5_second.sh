set -x
python -m pytest tests/test_gpu_toggles.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5b_toggles.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5b_parity.log
bash profiles/trace_small.sh > gpurun_out/r5b_trace.log 2>&1
cat gpurun_out/r5b_toggles.log gpurun_out/r5b_parity.log
