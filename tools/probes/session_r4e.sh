#!/bin/bash
# round 4, session e: AuxIVA with the statistic's finalize, the log-det terms and the loss sum folded into the pass
cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r4e}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_iterate.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_widem.py -m gpu -x -q -k "aux or iva or Aux or config3" 2>&1 | tail -8 > $O/tests.log
for d in float64 float32; do python tools/probes/small_cfg_probe.py cfg3 $d 2000 >> $O/small_cfgs.txt 2>&1; done
for d in float64 float32; do ASSX_AUX_FOLD=0 python tools/probes/small_cfg_probe.py cfg3 $d 2000 >> $O/small_cfgs_nofold.txt 2>&1; done
python tools/auxiva_bench.py > $O/auxiva_bench.txt 2>&1
