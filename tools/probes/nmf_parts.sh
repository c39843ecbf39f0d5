#!/bin/bash
# see nmf_parts_build.sh
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/full.so
for v in 0 ${NMF_PARTS:-1 2 4 8 16 24 31} 0; do
  if [ $v = 0 ]; then cp /tmp/full.so $C/libassx.so; else cp $C/ab/libassx_nmfskip$v.so $C/libassx.so; fi
  echo "== skip $v: K=10 source update $(python tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 2>/dev/null | grep source | awk '{print $5, $6}')   cfg2 $(python tools/nmf_bench.py float64 32 2>/dev/null | head -1)"
done
cp /tmp/full.so $C/libassx.so
