O=gpurun_out/s2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "rccl" 2>&1 | tail -15 > $O/multi_new.log
