O=gpurun_out/s2k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_widem.py -x -q -k "ip or cond or guard" 2>&1 | tail -4 > $O/ip_tests.log
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "frequency_sharded" 2>&1 | tail -6 > $O/fshard_tests.log
python bench.py --dtype float32 --cpu-iters 0 > $O/bench_f32.json 2>/dev/null
python bench.py --cpu-iters 0 --roofline-b8 0 > $O/bench_f64.json 2>/dev/null
python tools/probes/f32_call_probe.py > $O/f32_call_probe.txt 2>&1
python tools/probes/f32_call_probe.py float64 float32 >> $O/f32_call_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_f32 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --dtype float32 --roofline-b8 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $O/prof_f32 > $O/f32_kernel_stats.md 2>&1; rm -rf $O/prof_f32
