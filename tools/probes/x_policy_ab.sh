#!/bin/bash
# Cache-policy bits on the streamed X loads of the M <= 4 streaming kernels (ASSX_X_POLICY_ID builds of csrc/build.sh:
#   for v in base:0 nt:1 sc1:2; do ASSX_DEV=1 ASSX_CHECK=0 ASSX_OBJ=ab/o_${v%%:*} ASSX_OUT=ab/libassx_${v%%:*}.so \
#       ASSX_EXTRA_FLAGS="-DASSX_X_POLICY_ID=${v##*:}" bash build.sh; done )
# alternating libraries on one box: config 4 (X = 268.7 MB, just over the Infinity Cache) and AuxIVA config 3 (67 MB, resident)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-xpol}; mkdir -p $O; C=audio_source_separation_amd/csrc
for rep in 1 2; do for v in base nt sc1; do
  echo "== $v (rep $rep)" >> $O/x_policy_ab.txt
  ASSX_LIB_PATH=$C/ab/libassx_$v.so python bench.py --cpu-iters 0 --roofline-b8 0 --with-f32 0 --with-default-basis 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4', d['value'], 'it/s', d['ms_per_step'], 'ms; cov kernel', d['roofline']['kernel_ms'], 'ms')" >> $O/x_policy_ab.txt
  ASSX_LIB_PATH=$C/ab/libassx_$v.so python tools/auxiva_bench.py 2>/dev/null | grep "M=4" >> $O/x_policy_ab.txt
done; done
