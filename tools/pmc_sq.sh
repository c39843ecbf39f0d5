#!/bin/bash
# SQ instruction / stall counters of one microbench entry (two PMC passes; kernel-trace only, as gpurun requires).
#   bash tools/pmc_sq.sh "cov TV" float64 tag
set -e
ONLY="${1:-cov TV}"; DT="${2:-float64}"; TAG="${3:-sq}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/${TAG}_a -o p -- python $ROOT/tools/microbench.py --only "$ONLY" --reps 5 --dtype $DT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $ROOT/gpurun_out/${TAG}_b -o p -- python $ROOT/tools/microbench.py --only "$ONLY" --reps 5 --dtype $DT > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("$ROOT/gpurun_out/${TAG}_a","$ROOT/gpurun_out/${TAG}_b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'stream' in k and 'finalize' not in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
