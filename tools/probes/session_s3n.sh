O=gpurun_out/s3u; mkdir -p $O; rm -f $O/*
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_m5 -o p -- python $R/tools/widem_bench.py 5:4 > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_m5 > $R/$O/m5_kernel_stats.md 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_m8 -o p -- python $R/tools/widem_bench.py 8:4 > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_m8 > $R/$O/m8_kernel_stats.md 2>&1
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
