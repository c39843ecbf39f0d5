O=gpurun_out/s2c; mkdir -p $O
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -5 > $O/widem_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_m8 -o p -- python $ROOT/tools/widem_bench.py 8:4 > $ROOT/$O/m8.txt 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_m5 -o p -- python $ROOT/tools/widem_bench.py 5:4 > $ROOT/$O/m5.txt 2>&1
cd $ROOT
python tools/rocprof_summary.py $O/prof_m8 > $O/m8_kernel_stats.md 2>&1
python tools/rocprof_summary.py $O/prof_m5 > $O/m5_kernel_stats.md 2>&1
python tools/widem_bench.py 8:4 5:4 --dtype float32 > $O/f32.txt 2>&1
rm -rf $O/prof_*
