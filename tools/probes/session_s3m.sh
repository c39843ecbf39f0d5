O=gpurun_out/s3m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "wide_channel_full_size" 2>&1 | tail -60 > $O/tests.log
ASSX_WIDEM_PAIRS=0 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "wide_channel_full_size" 2>&1 | tail -5 > $O/tests_pairs0.log
