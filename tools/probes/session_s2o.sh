O=gpurun_out/s2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_manychan.py -x -q 2>&1 | tail -25 > $O/manychan.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -k "m9" 2>&1 | tail -15 > $O/models_m9.log
timeout 900 python -m pytest tests/test_gpu_widem.py tests/test_gpu_ops.py -x -q 2>&1 | tail -4 > $O/regress.log
