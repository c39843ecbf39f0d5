#!/bin/bash
# Headline bench, alternating libraries on one box:  bash tools/probes/lib_ab.sh <reps> <name=path/to/libassx.so> ...
# prints per run: name, it/s, ms per step, covariance kernel ms (one utterance), covariance kernel ms (8 utterances),
# utterance-it/s at 8 utterances per launch (AB_B8=1), it/s with the loss recorded (AB_LOSS=1)
REPS=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    ASSX_LIB_PATH=$ROOT/$lib python bench.py --cpu-iters 0 --with-f32 0 --with-default-basis 0 --with-other-configs 0 --with-loss ${AB_LOSS:-0} \
      --with-b8 ${AB_B8:-0} --steps ${AB_STEPS:-300} --warmup 30 ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; r8=d.get('roofline_b8') or {}
print('$name', d['value'], d['ms_per_step'], r['kernel_ms'], r8.get('kernel_ms'), d.get('value_b8'), d.get('value_with_loss'))"
  done
done
