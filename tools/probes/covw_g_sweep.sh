# cov_wide_kernel (n_basis = 10) against the number of workgroups in the launch (ASSX_G forces the partition's target):
# is there a second-round effect at 256 workgroups, and is the cost per item or per workgroup?
for G in 128 192 224 240 248 252 256 260 288 384 512; do
echo "G=$G: $(ASSX_G=$G python tools/covw_ab.py child float64 0 2>/dev/null | grep F1025 | awk '{print $(NF-1)}') us"
done
