O=gpurun_out/r5be; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "drift" -s 2>&1 | grep -i "layer\|drift\|passed\|failed" | head -30 > $O/drift_x3.txt
TNQS_NO_BF16X3=1 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "drift" -s 2>&1 | grep -i "layer\|drift\|passed\|failed" | head -30 > $O/drift_f32.txt
echo X3; cat $O/drift_x3.txt; echo F32; cat $O/drift_f32.txt
