#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4i}; mkdir -p $O
for w in 256 384 512 640 768 1024; do
  echo "== ASSX_NMF_XFED_WGS=$w" >> $O/xfed_wgs.txt
  ASSX_NMF_XFED_WGS=$w python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/xfed_wgs.txt
done
