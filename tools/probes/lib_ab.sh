#!/bin/bash
# A/B of two whole libraries on the GPU box (a = $NEW_LIB, default the library in the tree; b = $OLD_LIB): digests of the per-bin sweeps and of
# the covariance shapes, the bench lines (float64 with its float32 / n_basis 10 side lines), the wide-channel, NMF and
# small-config benches, alternating; optionally the GPU suite on (a).   OLD_LIB=... [TESTS=1] bash tools/probes/lib_ab.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/libab; mkdir -p $OUT; rm -f $OUT/*.txt
cd $ROOT
A=${NEW_LIB:-$ROOT/audio_source_separation_amd/csrc/libassx.so}
B=${OLD_LIB:?OLD_LIB}
ASSX_LIB_PATH=$A python tools/probes/ip_dpp_ab.py > $OUT/dig_a.txt 2>&1
ASSX_LIB_PATH=$B python tools/probes/ip_dpp_ab.py > $OUT/dig_b.txt 2>&1
diff <(sed 's/ *#.*//' $OUT/dig_a.txt) <(sed 's/ *#.*//' $OUT/dig_b.txt) > $OUT/diff.txt && echo "sweep digests equal" >> $OUT/diff.txt
ASSX_LIB_PATH=$A python tools/covw_ab.py all 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' > $OUT/covw_a.txt
ASSX_LIB_PATH=$B python tools/covw_ab.py all 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' > $OUT/covw_b.txt
diff $OUT/covw_a.txt $OUT/covw_b.txt >> $OUT/diff.txt && echo "covariance digests equal" >> $OUT/diff.txt
for rep in 1 2; do for v in a b; do
  lib=$A; [ $v = b ] && lib=$B
  ASSX_LIB_PATH=$lib python bench.py --cpu-iters 0 --roofline-b8 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench f64', d['value'], d['ms_per_step'], 'f32', d.get('value_f32'), 'k10', d.get('value_k10'), d.get('value_k10_with_loss'))" | tee -a $OUT/summary.txt
  ASSX_LIB_PATH=$lib python bench.py --dtype float32 --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench f32 main', d['value'], d['ms_per_step'])" | tee -a $OUT/summary.txt
done; done
for v in a b a b; do
  lib=$A; [ $v = b ] && lib=$B
  echo "== $v" >> $OUT/summary.txt
  ASSX_LIB_PATH=$lib python tools/widem_bench.py 5:4 8:4 8:10 2>/dev/null >> $OUT/summary.txt
  ASSX_LIB_PATH=$lib python tools/widem_bench.py 5:4 8:4 --dtype float32 2>/dev/null >> $OUT/summary.txt
  ASSX_LIB_PATH=$lib python tools/nmf_bench.py float64 2>/dev/null | head -4 >> $OUT/summary.txt
  ASSX_LIB_PATH=$lib python tools/nmf_bench.py float32 2>/dev/null | head -4 >> $OUT/summary.txt
  for c in cfg1 cfg3; do for d in float64 float32; do ASSX_LIB_PATH=$lib python tools/probes/small_cfg_probe.py $c $d 2000 2>/dev/null | tail -1 >> $OUT/summary.txt; done; done
done
[ "${TESTS:-0}" = 1 ] && ASSX_LIB_PATH=$A timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 >> $OUT/summary.txt
cat $OUT/diff.txt; cat $OUT/summary.txt
