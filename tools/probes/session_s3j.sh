O=gpurun_out/s3j; mkdir -p $O
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/main.so
for v in main vf main vf; do
  if [ $v = main ]; then cp /tmp/main.so $C/libassx.so; else cp $C/libassx_$v.so $C/libassx.so; fi
  echo "== $v" >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float64 32 2>/dev/null | head -3 >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float64 10 2>/dev/null | head -1 >> $O/bench.txt
  timeout 300 python tools/nmf_bench.py float32 32 2>/dev/null | head -1 >> $O/bench.txt
  timeout 300 python bench.py --basis 10 --cpu-iters 0 --roofline-b8 0 --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=10', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])" >> $O/bench.txt
  timeout 300 python tools/widem_bench.py 8:10 2>/dev/null >> $O/bench.txt
done
cp /tmp/main.so $C/libassx.so
