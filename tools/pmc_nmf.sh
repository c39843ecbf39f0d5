#!/bin/bash
# SQ counters of the NMF matrix-core kernels (config 2).  bash tools/pmc_nmf.sh [dtype]
DT="${1:-float64}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/nmf_a -o p -- python $ROOT/tools/nmf_bench.py $DT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $ROOT/gpurun_out/nmf_b -o p -- python $ROOT/tools/nmf_bench.py $DT > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("$ROOT/gpurun_out/nmf_a","$ROOT/gpurun_out/nmf_b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nmf_' in k and 'mfma' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
