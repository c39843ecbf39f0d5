#!/bin/bash
# workgroup budgets of the two matrix-core NMF halves (flat partition): config 2, first lines of tools/nmf_bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4c}; mkdir -p $O
for d in float64 float32; do
for b in 256 384 448 512 576 640 768 1024; do for a in 256 512 768 1024; do
  echo "== $d basis_wgs=$b act_wgs=$a" >> $O/nmf_wgs_sweep.txt
  ASSX_NMF_BASIS_WGS=$b ASSX_NMF_ACT_WGS=$a python tools/nmf_bench.py $d 2>/dev/null | head -3 >> $O/nmf_wgs_sweep.txt
done; done; done
