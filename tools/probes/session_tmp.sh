cd $GRAFT_REPO_ROOT; O=gpurun_out/r4t; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_widem.py tests/test_gpu_manychan.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.log
python bench.py --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 > $O/new.json 2>/dev/null
ASSX_LIB_PATH=audio_source_separation_amd/csrc/ab/libassx_base.so python bench.py --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 > $O/old.json 2>/dev/null
python bench.py --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 > $O/new2.json 2>/dev/null
ASSX_LIB_PATH=audio_source_separation_amd/csrc/ab/libassx_base.so python bench.py --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 > $O/old2.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --roofline-b8 0 --with-f32 0 --with-default-basis 0 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof > $GRAFT_REPO_ROOT/$O/stats.md 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/prof
