O=gpurun_out/r5q; mkdir -p $O
for m in 0 1 2; do echo "== mode $m"; TNQS_X3_MODE=$m python profiles/plane_bench.py 100 5 2>&1 | grep gram2; done | tee $O/modes.txt
