#!/bin/bash
# Knock-out timing of widem::pair_cov_kernel (M = 8): probe libraries in which one part of the trip is removed
# (PAIRCOV_SKIP bits, csrc/assx_widem_cov.hpp; results are wrong, only the kernel time is read).
#   bash tools/probes/paircov_knockout.sh build      # here: csrc/ab/libassx_pcskip<N>.so for N in $SKIPS (needs csrc/*.o)
#   bash tools/probes/paircov_knockout.sh run        # on the GPU box: rocprofv3 kernel time of each
SKIPS="${SKIPS:-1 2 4 8 16 31}"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$ROOT/audio_source_separation_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $C/ab
  for n in $SKIPS; do
    ( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -w -DASSX_PROBE_BUILD -DPAIRCOV_SKIP=$n \
        -c assx_widem.hip -o ab/widem_pcskip$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libassx_pcskip$n.so assx_api.o assx_bss.o assx_nmf.o assx_stft.o \
        assx_generic.o ab/widem_pcskip$n.o assx_xfer.o assx_iterate.o -lpthread && echo "built pcskip$n" ) &
  done
  wait
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pcskip; mkdir -p $OUT
cd $ROOT
for n in 0 $SKIPS; do
  lib=$C/ab/libassx_pcskip$n.so; [ $n = 0 ] && lib=${BASE_LIB:-$C/libassx.so}
  ASSX_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/prof_$n -o p -- python tools/widem_bench.py 8:4 > $OUT/bench_$n.txt 2>&1
  echo "skip=$n $(python tools/rocprof_summary.py $OUT/prof_$n 2>/dev/null | grep pair_cov | head -1)" | tee -a $OUT/summary.txt
  rm -rf $OUT/prof_$n
done
