#!/bin/bash
# SQ instruction / stall counters of one microbench entry for any kernel-name substring (PMC passes with kernel-trace
# only, as gpurun requires).   bash tools/pmc_kernel.sh "cov TV partial" cov_wide tag [extra microbench args...]
set -e
ONLY="$1"; KSUB="$2"; TAG="$3"; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc "$@" ; }
P="python $ROOT/tools/microbench.py --only"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/${TAG}_a -o p -- $P "$ONLY" --reps 5 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $ROOT/gpurun_out/${TAG}_b -o p -- $P "$ONLY" --reps 5 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $ROOT/gpurun_out/${TAG}_c -o p -- $P "$ONLY" --reps 5 "$@" > /dev/null 2>&1 || true
python3 - <<PY
import csv,glob,collections
for d in ("$ROOT/gpurun_out/${TAG}_a","$ROOT/gpurun_out/${TAG}_b","$ROOT/gpurun_out/${TAG}_c"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if "$KSUB" in k and 'finalize' not in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
