#!/bin/bash
# round 4, session h: X-fed NMF halves for the n_basis > 4 ILRMA source model
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4h}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_iterate.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.log
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_f64_k10.json 2>$O/bench_k10.err
ASSX_NMF_XFED=0 python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_f64_k10_map.json 2>/dev/null
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 --dtype float32 > $O/bench_f32_k10.json 2>/dev/null
python bench.py --cpu-iters 0 --basis 16 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_f64_k16.json 2>/dev/null
python bench.py --cpu-iters 0 --basis 32 --steps 100 --warmup 20 --roofline-b8 0 > $O/bench_f64_k32.json 2>/dev/null
python bench.py --cpu-iters 0 --basis 10 --utterances-per-gpu 8 --steps 50 --warmup 5 --roofline-b8 0 > $O/bench_f64_k10_8utt.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_k10 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --basis 10 --roofline-b8 0 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof_k10 > $GRAFT_REPO_ROOT/$O/k10_kernel_stats.md 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/prof_k10
