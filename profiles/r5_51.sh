python profiles/plane16_bench.py 12 3 01,12,23,35,45 2>&1 | tail -12
