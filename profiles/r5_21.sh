O=gpurun_out/r5ad; mkdir -p $O
python profiles/level_bench.py 96 3 2>&1 | tee $O/level.txt
