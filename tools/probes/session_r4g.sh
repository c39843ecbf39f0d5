#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4g}; mkdir -p $O
for d in float64 float32; do
  echo "== ASSX_AUX_FOLD=1" >> $O/call_cfgs.txt; ASSX_AUX_FOLD=1 python tools/probes/call_cfgs.py $d 2>/dev/null >> $O/call_cfgs.txt
  echo "== ASSX_AUX_FOLD=0" >> $O/call_cfgs.txt; ASSX_AUX_FOLD=0 python tools/probes/call_cfgs.py $d 2>/dev/null | grep cfg3 >> $O/call_cfgs.txt
done
