# does the power-of-two row stride of X (T = 4096 frames x 16 B = 64 KiB) cost bandwidth?  same kernel, other T
for T in 4096 4032 4160 4100 3968 4224; do
echo "T=$T B=8: $(python tools/microbench.py --B 8 --T $T --only "cov TV" --reps 20 2>/dev/null | grep cov | awk '{print $5, $6, $7, $8}')  | B=1: $(python tools/microbench.py --B 1 --T $T --only "cov TV" --reps 40 2>/dev/null | grep cov | awk '{print $5, $6, $7, $8}')"
done
