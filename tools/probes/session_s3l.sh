O=gpurun_out/s3l; mkdir -p $O
for c in cfg1 cfg3; do for d in float64 float32; do
  python tools/probes/small_cfg_probe.py $c $d 2>/dev/null >> $O/rates.txt
done; done
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg3; do for d in float64 float32; do
  rocprofv3 --kernel-trace --stats -d /tmp/p_${c}_$d -o p -- python $GRAFT_REPO_ROOT/tools/probes/small_cfg_probe.py $c $d > /dev/null 2>&1
  echo "== $c $d" >> $GRAFT_REPO_ROOT/$O/kernels.md
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/p_${c}_$d 2>&1 | head -12 >> $GRAFT_REPO_ROOT/$O/kernels.md
done; done
