cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ai; mkdir -p $O
NREP=10 rocprofv3 --hip-runtime-trace --stats -d /tmp/hh -o out --output-format csv -- python $R/profiles/shape_bench.py heavyhex > $O/hh.log 2>&1
f=$(find /tmp/hh -name "*hip_api_stats.csv" | head -1); cp $f $O/hh_hip_api_stats.csv; head -25 $f
tail -2 $O/hh.log | cut -c1-300
