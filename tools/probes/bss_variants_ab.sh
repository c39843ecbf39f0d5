#!/bin/bash
# A/B of single-precision-or-double-only builds of assx_bss.hip (csrc/ab/libassx_<tag>.so: -DASSX_DEV_ONLY_M4_F32 / _F64
# objects linked with the tree's other objects): the bench line and kernel table per variant, alternating, on the GPU box.
#   TAGS="pk0 pk1 ..." [DTYPE=float64] bash tools/probes/bss_variants_ab.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
DTYPE=${DTYPE:-float32}
OUT=$ROOT/gpurun_out/bssab_$DTYPE; mkdir -p $OUT; rm -f $OUT/summary.txt
cd $ROOT
for rep in 1 2; do for t in $TAGS; do
  lib=$ROOT/audio_source_separation_amd/csrc/ab/libassx_$t.so
  ASSX_LIB_PATH=$lib python bench.py --dtype $DTYPE --cpu-iters 0 --roofline-b8 0 --with-default-basis 0 --with-f32 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])" | tee -a $OUT/summary.txt
done; done
for t in $TAGS; do
  lib=$ROOT/audio_source_separation_amd/csrc/ab/libassx_$t.so
  ASSX_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/prof_$t -o p -- python bench.py --steps 20 --warmup 5 --cpu-iters 0 --dtype $DTYPE --roofline-b8 0 --with-default-basis 0 --with-f32 0 > /dev/null 2>&1
  echo "== $t" >> $OUT/summary.txt; python tools/rocprof_summary.py $OUT/prof_$t 2>/dev/null | sed -n 3,6p | cut -c1-120 | tee -a $OUT/summary.txt
  rm -rf $OUT/prof_$t
done
