# covariance kernel at 8 utterances per launch against the per-utterance workgroup target (ASSX_G)
for G in 0 1025 1640 1988 2050 2187 2624 3280 4100; do
echo "ASSX_G=$G: $(ASSX_G=$G python tools/microbench.py --B 8 --only "cov TV" --reps 20 2>/dev/null | grep cov)  | B=1: $(ASSX_G=$G python tools/microbench.py --B 1 --only "cov TV" --reps 40 2>/dev/null | grep cov | awk '{print $5, $6}')"
done
