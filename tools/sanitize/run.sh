#!/bin/bash
# Host-side sanitizer runs of libassx's threaded code (round 5's review, weak #11; SURVEY.md section 5 "ASan for the host shim"):
#     bash tools/sanitize/run.sh [out.log]
# Builds csrc/assx_api.hip + csrc/assx_xfer.hip HOST-ONLY (hipcc --cuda-host-only) twice -- AddressSanitizer +
# UndefinedBehaviorSanitizer, then ThreadSanitizer -- links each against tools/sanitize/hip_stub.cpp (a host emulation of
# the HIP calls those files make: asynchronous in-order streams, events, heap "device" memory) and runs
# tools/sanitize/driver.cpp: uploads / downloads of ragged sizes in all four precision pairings with 1, 3 and 8 pool
# threads, on the default and on a user stream, twice round the staging ring; the ticket slots under more streams than
# slots and repeated growth.  No GPU needed.  Exit code 0 = both runs clean.
set -uo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
CSRC=${ASSX_SAN_CSRC:-$ROOT/audio_source_separation_amd/csrc}   # ASSX_SAN_CSRC: a mutated copy (does the harness catch a seeded bug?)
LOG=${1:-/dev/stdout}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
rc=0
{
for mode in asan tsan; do
  if [ $mode = asan ]; then SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; else SAN="-fsanitize=thread"; fi
  FL="-std=c++17 -g -O1 -fno-omit-frame-pointer $SAN"
  echo "== $mode: $FL"
  for u in assx_api assx_xfer; do
    $HIPCC --cuda-host-only $FL -Wall -c $CSRC/$u.hip -o $TMP/$u.$mode.o || rc=1
  done
  $CXX $FL -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $HERE/hip_stub.cpp -o $TMP/stub.$mode.o || rc=1
  $CXX $FL -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $HERE/driver.cpp -o $TMP/driver.$mode.o || rc=1
  $CXX $FL $TMP/assx_api.$mode.o $TMP/assx_xfer.$mode.o $TMP/stub.$mode.o $TMP/driver.$mode.o -lpthread -o $TMP/run_$mode || rc=1
  if [ $mode = asan ]; then
    ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 $TMP/run_asan || rc=1
  else
    TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 $TMP/run_tsan || rc=1
  fi
  echo "== $mode exit status so far: $rc"
done
echo "sanitizer runs: $([ $rc = 0 ] && echo CLEAN || echo FAILED)"
} > "$LOG" 2>&1
exit $rc
