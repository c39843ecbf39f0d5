#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4j}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_iterate.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.log
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_f64_k10.json 2>$O/bench_k10.err
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 --with-loss > $O/bench_f64_k10_loss.json 2>/dev/null
ASSX_FUSE_LOSS=0 python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 --with-loss > $O/bench_f64_k10_loss_unfused.json 2>/dev/null
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 --with-loss --dtype float32 > $O/bench_f32_k10_loss.json 2>/dev/null
