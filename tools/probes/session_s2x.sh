O=gpurun_out/s2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_manychan.py -x -q 2>&1 | tail -2 > $O/tests.log
for ns in 1 2 4 8; do echo "ASSX_RT_NS=$ns" >> $O/ns_sweep.txt; ASSX_RT_NS=$ns python tools/widem_bench.py 9:4 12:4 16:4 2>/dev/null >> $O/ns_sweep.txt; done
