set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5trace2; mkdir -p $O
export TNQS_BENCH_NOPROF=1
rocprofv3 --kernel-trace --output-format csv -d $O/l7 -- python $R/bench.py --L 7 --steps 2 --warmup 2 --no-cpu-baseline > $O/l7.log 2>&1
python $R/profiles/timeline.py $(ls $O/l7/*/*kernel_trace.csv | head -1) 4 > $O/l7_timeline.txt
tail -3 $O/l7.log | cut -c1-300
