#!/bin/bash
# Which part of the NMF matrix-core kernels costs what: libraries with one part compiled out (NMF_SKIP bits,
# csrc/assx_nmf_mfma.hpp; results are wrong by construction, only the time means something).
#   bash tools/probes/nmf_parts_build.sh   (here, no GPU)   then on the GPU box:  bash tools/probes/nmf_parts.sh
set -e
cd "$(dirname "$0")/../../audio_source_separation_amd/csrc"
mkdir -p ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
pids=()
for v in ${NMF_PARTS:-1 2 4 8 16 24 31}; do
  (/opt/rocm/bin/hipcc $FLAGS -DNMF_SKIP=$v -c assx_nmf.hip -o ab/nmf_$v.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libassx_nmfskip$v.so assx_api.o assx_bss.o ab/nmf_$v.o assx_stft.o assx_generic.o assx_widem.o assx_xfer.o -lpthread) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la ab/libassx_nmfskip*.so
