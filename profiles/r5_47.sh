O=gpurun_out/r5bm; mkdir -p $O
TNQS_NO_BF16X3=1 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -4 > $O/suite_f32.log; cat $O/suite_f32.log
