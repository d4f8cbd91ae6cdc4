cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5trace2; mkdir -p $O
NREP=2 rocprofv3 --kernel-trace --output-format csv -d $O/hh -- python $R/profiles/shape_bench.py heavyhex > $O/hh.log 2>&1
python $R/profiles/timeline.py $(ls $O/hh/*/*kernel_trace.csv | head -1) 4 > $O/hh_timeline.txt
TNQS_NO_SMALL_SITE_FINALIZE=1 NREP=2 rocprofv3 --kernel-trace --output-format csv -d $O/hh1 -- python $R/profiles/shape_bench.py heavyhex > $O/hh1.log 2>&1
python $R/profiles/timeline.py $(ls $O/hh1/*/*kernel_trace.csv | head -1) 4 > $O/hh1_timeline.txt
rm -rf $O/hh $O/hh1
tail -n 2 $O/hh.log
