#!/bin/bash
# A/B build of the M <= 4 streaming unit only:  bash tools/probes/ab_build.sh <name> [extra -D flags...]
# -> audio_source_separation_amd/csrc/ab_<name>/libassx.so (ASSX_LIB_PATH selects it at run time).  The other translation
# units are taken from the last full build (their objects are copied and touched), assx_bss.hip is compiled with
# ASSX_DEV=1 (M = 4, float64 instantiations only) and without the asm check: a tuning build, never shipped.
set -euo pipefail
NAME=$1; shift
CSRC=$(cd "$(dirname "$0")/../../audio_source_separation_amd/csrc" && pwd)
mkdir -p $CSRC/ab_$NAME
for o in assx_api assx_nmf assx_stft assx_generic assx_widem assx_xfer assx_iterate assx_comm; do
  cp $CSRC/$o.o $CSRC/ab_$NAME/$o.o
done
sleep 1; touch $CSRC/ab_$NAME/*.o
rm -f $CSRC/ab_$NAME/assx_bss.o
ASSX_DEV=${ASSX_DEV:-1} ASSX_CHECK=0 ASSX_EXTRA_FLAGS="$*" ASSX_OBJ=ab_$NAME ASSX_OUT=ab_$NAME/libassx.so bash $CSRC/build.sh
