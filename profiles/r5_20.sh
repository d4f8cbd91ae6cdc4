O=gpurun_out/r5ac; mkdir -p $O
python profiles/plane16_bench.py 12 3 2>&1 | tee $O/plane16.txt
