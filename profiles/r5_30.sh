O=gpurun_out/r5ap; mkdir -p $O
python -m pytest tests -q -m gpu -x --tb=short 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/suite.log; cat $O/suite.log
