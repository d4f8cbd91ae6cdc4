O=gpurun_out/r5final2; mkdir -p $O
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -1 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; head -3 $O/collect_mfma.log
bash profiles/collect_stalls.sh r5 > $O/collect_stalls.log 2>&1
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "periodic_lattices or small_sites_with_16 or default or random_graphs" 2>&1 | tail -3
