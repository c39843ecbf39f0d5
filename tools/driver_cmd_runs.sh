#!/bin/bash
# The driver's exact bench command, N times on one box, next to the 500/50 line of the same box:
#   bash tools/driver_cmd_runs.sh r06 [N] [other_bench.py]
# -> gpurun_out/<tag>/bench_driver_cmd.json (one JSON line per run, then the 500/50 line); with a third argument the
# runs alternate with another bench file (A/B of the measurement protocol itself) -> bench_driver_cmd_other.json
TAG="${1:-r06}"; N="${2:-10}"; OTHER="${3:-}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
: > $OUT/bench_driver_cmd.json
[ -n "$OTHER" ] && : > $OUT/bench_driver_cmd_other.json
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> $OUT/bench_driver_cmd.json
  [ -n "$OTHER" ] && python3 $OTHER --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> $OUT/bench_driver_cmd_other.json
done
python3 bench.py --cpu-iters 0 2>/dev/null | tail -n 1 > $OUT/bench_f64_500_50.json
python3 - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
for name in ("bench_driver_cmd.json", "bench_driver_cmd_other.json", "bench_f64_500_50.json"):
    p = os.path.join(out, name)
    if not os.path.exists(p):
        continue
    v = [json.loads(l) for l in open(p) if l.strip().startswith("{")]
    vals = [d["value"] for d in v]
    if vals:
        print(name, "n=%d" % len(vals), "min %.1f median %.1f max %.1f" % (min(vals), sorted(vals)[len(vals) // 2], max(vals)),
              "ms/step", [d["ms_per_step"] for d in v])
PY
