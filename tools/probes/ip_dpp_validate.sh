cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/ipdpp4; mkdir -p $OUT
NODPP=$ROOT/audio_source_separation_amd/csrc/ab/libassx_nodpp.so
python tools/probes/ip_dpp_ab.py > $OUT/a.txt 2>&1
ASSX_LIB_PATH=$NODPP python tools/probes/ip_dpp_ab.py > $OUT/b.txt 2>&1
diff <(sed 's/ *#.*//' $OUT/a.txt) <(sed 's/ *#.*//' $OUT/b.txt) > $OUT/diff.txt && echo "all digests equal" >> $OUT/diff.txt
python tools/covw_ab.py all > $OUT/covw_ab.txt 2>&1
for rep in 1 2; do
  python bench.py > $OUT/bench_a$rep.json 2>$OUT/bench_a$rep.err
  ASSX_LIB_PATH=$NODPP python bench.py > $OUT/bench_b$rep.json 2>$OUT/bench_b$rep.err
done
rocprofv3 --kernel-trace --stats -d $OUT/prof_f32 -o p -- python bench.py --steps 20 --warmup 5 --cpu-iters 0 --dtype float32 --roofline-b8 0 --with-default-basis 0 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/prof_f32 2>/dev/null | head -12 > $OUT/f32_kernels.md
ASSX_LIB_PATH=$NODPP rocprofv3 --kernel-trace --stats -d $OUT/prof_f32b -o p -- python bench.py --steps 20 --warmup 5 --cpu-iters 0 --dtype float32 --roofline-b8 0 --with-default-basis 0 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/prof_f32b 2>/dev/null | head -12 > $OUT/f32_kernels_nodpp.md
rm -rf $OUT/prof_f32 $OUT/prof_f32b
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $OUT/tests.log
cat $OUT/diff.txt; grep -c "same bits" $OUT/covw_ab.txt; cat $OUT/tests.log; cat $OUT/bench_*.json | cut -c1-400; cat $OUT/f32_kernels.md $OUT/f32_kernels_nodpp.md
