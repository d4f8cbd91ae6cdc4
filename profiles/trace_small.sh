set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5trace; mkdir -p $O
export TNQS_BENCH_NOPROF=1
rocprofv3 --kernel-trace --output-format csv -d $O/l7 -- python $R/bench.py --L 7 --steps 2 --warmup 2 --no-cpu-baseline > $O/l7.log 2>&1
python $R/profiles/timeline.py $(ls $O/l7/*/*kernel_trace.csv | head -1) 4 > $O/l7_timeline.txt
NREP=2 rocprofv3 --kernel-trace --output-format csv -d $O/hh -- python $R/profiles/shape_bench.py heavyhex > $O/hh.log 2>&1
python $R/profiles/timeline.py $(ls $O/hh/*/*kernel_trace.csv | head -1) 4 > $O/hh_timeline.txt
NREP=2 rocprofv3 --kernel-trace --output-format csv -d $O/c1 -- python $R/profiles/shape_bench.py c1 > $O/c1.log 2>&1
python $R/profiles/timeline.py $(ls $O/c1/*/*kernel_trace.csv | head -1) 4 > $O/c1_timeline.txt
rm -rf $O/l7 $O/hh $O/c1
tail -n 3 $O/hh.log
