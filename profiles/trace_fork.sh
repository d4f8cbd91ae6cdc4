cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4trace; mkdir -p $O
export TNQS_BENCH_NOPROF=1
rocprofv3 --kernel-trace --output-format csv -d $O/l20 -- python $R/bench.py --L ${1:-20} --steps 1 --warmup 1 --no-cpu-baseline > $O/l20.log 2>&1
python $R/profiles/timeline.py $(ls $O/l20/*/*kernel_trace.csv | head -1) 2 > $O/l20_timeline.txt
rm -rf $O/l20
tail -n 2 $O/l20_timeline.txt
