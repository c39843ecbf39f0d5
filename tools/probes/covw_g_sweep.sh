# cov_wide kernels against the number of workgroups in the launch (ASSX_G forces the partition's target)
for ring in 0 3; do for G in 128 192 224 240 248 252 256 260 288 384 512; do
echo "RING=$ring PP=0 G=$G: $(ASSX_G=$G ASSX_COVW_RING=$ring ASSX_COVW_PP=0 python tools/covw_ab.py child float64 0 2>/dev/null | grep F1025 | awk '{print $(NF-1)}') us"
done; done
