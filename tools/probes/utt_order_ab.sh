# needs a laboratory build of the library (ASSX_EXTRA_FLAGS=-DASSX_LAB=1 csrc/build.sh; ASSX_LIB_PATH): the shipped one does not read ASSX_UTT_ORDER
# utterance-sequential launch order with alternating direction (ASSX_UTT_ORDER=1, default) against the legacy order
for b in 2 4 8 16; do for o in 0 1; do
ASSX_UTT_ORDER=$o python bench.py --cpu-iters 0 --utterances-per-gpu $b --steps $((400/b+10)) --warmup 10 --roofline-b8 0 --kernel-reps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ASSX_UTT_ORDER=$o utterances/GPU %2d: %8.1f utterance-it/s, %.4f ms/step, cov kernel %.4f ms = %.3f of 8 TB/s' % (d['config']['utterances_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
