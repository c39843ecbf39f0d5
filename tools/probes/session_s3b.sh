O=gpurun_out/s3b; mkdir -p $O
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/main.so
for v in main d4 d5 main; do
  if [ $v = main ]; then cp /tmp/main.so $C/libassx.so; else cp $C/libassx_$v.so $C/libassx.so; fi
  echo "== $v" >> $O/bench.txt
  timeout 600 python tools/widem_bench.py 7:4 8:4 2>/dev/null >> $O/bench.txt
  timeout 600 python tools/widem_bench.py --dtype float32 8:4 2>/dev/null >> $O/bench.txt
done
cp /tmp/main.so $C/libassx.so
timeout 1200 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -3 > $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_m8 -o p -- python $GRAFT_REPO_ROOT/tools/widem_bench.py 8:4 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_m8 > $GRAFT_REPO_ROOT/$O/m8_kernel_stats.md 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_kernel.sh "ilrma_spatial_update" pair_cov s3b/sq_pair_cov_m8 --M 8 > $O/sq_pair_cov_m8.txt 2>&1
