for G in 0 2050 3075 4096 4100; do
echo "ASSX_G=$G: $(ASSX_G=$G python tools/microbench.py --dtype float32 --only "cov TV" --reps 50 2>/dev/null | grep cov)"
echo "ASSX_G=$G: $(ASSX_G=$G python tools/microbench.py --dtype float32 --only "ilrma_source" --reps 50 2>/dev/null | grep ilrma)"
done
