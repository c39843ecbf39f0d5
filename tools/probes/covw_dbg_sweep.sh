for pp in 0 1; do for dbg in 0 1 2 4 6 8 16 9 15 31; do
echo "PP=$pp DBG=$dbg: $(ASSX_COVW_RING=3 ASSX_COVW_PP=$pp ASSX_COVW_DBG=$dbg python tools/covw_ab.py child float64 0 2>/dev/null | grep F1025 | awk '{print $(NF-1)}') us"
done; done
