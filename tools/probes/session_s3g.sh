O=gpurun_out/s3g; mkdir -p $O
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/main.so
for v in main d4 d6 main; do
  if [ $v = main ]; then cp /tmp/main.so $C/libassx.so; else cp $C/libassx_$v.so $C/libassx.so; fi
  echo "== $v" >> $O/bench.txt
  timeout 600 python tools/widem_bench.py 5:4 6:4 2>/dev/null >> $O/bench.txt
  timeout 600 python tools/widem_bench.py --dtype float32 5:4 6:4 2>/dev/null >> $O/bench.txt
  if [ $v = d4 ]; then timeout 1200 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -3 > $O/tests_d4.log; fi
done
cp /tmp/main.so $C/libassx.so
