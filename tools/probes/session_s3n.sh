O=gpurun_out/s3v; mkdir -p $O; rm -f $O/*
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/main.so
for v in main nmf2 main nmf2; do
  if [ $v = main ]; then cp /tmp/main.so $C/libassx.so; else cp $C/libassx_$v.so $C/libassx.so; fi
  echo "== $v" >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float64 32 2>/dev/null | head -3 >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float64 10 2>/dev/null | head -1 >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float32 32 2>/dev/null | head -1 >> $O/bench.txt
  timeout 300 python bench.py --basis 10 --cpu-iters 0 --roofline-b8 0 --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=10', d['value'], d['ms_per_step'])" >> $O/bench.txt
done
cp $C/libassx_nmf2.so $C/libassx.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -x -q -k "nmf or NMF or source or ilrma" 2>&1 | tail -3 > $O/tests.log
cp /tmp/main.so $C/libassx.so
