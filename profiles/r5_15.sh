O=gpurun_out/r5t; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "double_pair" --tb=short 2>&1 | tail -3
for m in 0 1 3 4; do echo "== mode $m"; TNQS_X3_MODE=$m python profiles/plane_bench.py 100 5 2>&1 | grep -E "gram2  legs \((0,1|1,2)"; done | tee $O/modes.txt
