O=gpurun_out/r5suite2; mkdir -p $O
python -m pytest tests -q -m gpu -x --tb=short > $O/suite_full.log 2>&1
grep -E "passed|failed|error" $O/suite_full.log | tail -3 > $O/suite.log; cat $O/suite.log
grep -E "FAILED|Error|assert" $O/suite_full.log | head -20
