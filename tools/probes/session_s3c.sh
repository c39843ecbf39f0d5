O=gpurun_out/s3h; mkdir -p $O
C=audio_source_separation_amd/csrc
timeout 600 python tools/widem_bench.py 5:4 6:4 7:4 8:4 2>/dev/null > $O/bench.txt
ASSX_WIDEM_PAIRS=0 timeout 600 python tools/widem_bench.py 8:4 2>/dev/null >> $O/bench.txt
timeout 1200 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -3 > $O/tests.log
cp $C/libassx.so /tmp/main.so; cp $C/libassx_trace.so $C/libassx.so
timeout 300 python tools/probes/paircov_trace.py 8 > $O/trace_m8.txt 2>&1
timeout 300 python tools/probes/paircov_trace.py 5 > $O/trace_m5.txt 2>&1
cp /tmp/main.so $C/libassx.so
