O=gpurun_out/s2m; mkdir -p $O; ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$O/p1 -o p -- python $ROOT/tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/$O/p2 -o p -- python $ROOT/tools/widem_bench.py 8:4 > /dev/null 2>&1
cd $ROOT
python tools/rocprof_summary.py $O/p1 > $O/k10_src.md 2>&1
python tools/rocprof_summary.py $O/p2 > $O/m8_src.md 2>&1
rm -rf $O/p1 $O/p2
