# wide-channel streaming covariance: parity, then timing against the round-2 kernel (ASSX_WIDEM_COV=0)
O=gpurun_out/s2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -15 > $O/widem_tests.log
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -k "m5 or m6 or m8 or widem or wide" 2>&1 | tail -5 > $O/widem_model_tests.log
python tools/widem_bench.py > $O/widem_bench_new.txt 2>&1
ASSX_WIDEM_COV=0 python tools/widem_bench.py > $O/widem_bench_old.txt 2>&1
