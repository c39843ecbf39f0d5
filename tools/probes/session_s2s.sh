O=gpurun_out/s2s; mkdir -p $O; ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_widem.py tests/test_gpu_models.py -x -q -k "nmf or source or ilrma" 2>&1 | tail -3 > $O/tests.log
python tools/widem_bench.py 5:4 8:4 > $O/widem.txt 2>&1
python tools/nmf_bench.py float64 4 | head -1 > $O/nmf_k4.txt 2>&1
ASSX_NMF_SMALL=0 python tools/nmf_bench.py float64 4 | head -1 >> $O/nmf_k4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$O/p -o p -- python $ROOT/tools/widem_bench.py 8:4 > /dev/null 2>&1
cd $ROOT; python tools/rocprof_summary.py $O/p > $O/m8.md 2>&1; rm -rf $O/p
