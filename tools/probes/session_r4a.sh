#!/bin/bash
# round 4, session a: the folded NMF finalize + the one-call loops -- parity first, then numbers
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4a}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_iterate.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
python tools/nmf_bench.py float64 > $O/nmf_bench_f64.txt 2>&1
python tools/nmf_bench.py float32 > $O/nmf_bench_f32.txt 2>&1
for c in cfg1 cfg3; do for d in float64 float32; do python tools/probes/small_cfg_probe.py $c $d 2000 >> $O/small_cfgs.txt 2>&1; done; done
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_f64_k10.json 2>$O/bench_k10.err
python bench.py --cpu-iters 0 --roofline-b8 0 > $O/bench_f64.json 2>$O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_nmf -o p -- python $GRAFT_REPO_ROOT/tools/nmf_bench.py float64 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof_nmf > $GRAFT_REPO_ROOT/$O/nmf_kernel_stats.md 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/prof_nmf
