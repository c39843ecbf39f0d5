#!/usr/bin/env python3
"""Timeline of cov_mfma_kernel's trips from a -DASSX_PROBE_BUILD -DCOVM_TRACE=1 build (see assx_cov_mfma.hpp): shader-clock stamps of waves 0
and 5 of workgroup 100 at 8 points of every trip, printed as the mean cycles between consecutive points."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.ops import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = Engine("float64")
B, M, F, T = 1, 4, 1025, 4096
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
     1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).contiguous()
Tb = torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
V = torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
for _ in range(5):
    eng.ilrma_cov_partials(X, Tb, V)
torch.cuda.synchronize()
ws = eng._scratch(B, M, F, T, K)
n = eng._L.assx_workspace_bytes(B, M, F, T, K, eng.prec.code)
tail = ws[n - 65536:n].view(torch.int64)
tail.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.ilrma_cov_partials(X, Tb, V)
e1.record()
torch.cuda.synchronize()
print("kernel (event) %.1f us" % (e0.elapsed_time(e1) * 1e3))
st = tail.cpu().numpy()
names = ["top->vmcnt", "vmcnt->barrier", "barrier->publish", "publish->products+fan+requests", "->rows/flush/advance", "loop->top"]
NS = len(names)
for wsel, off in (("wave0", 0), ("wave5", 400)):
    a = st[off:off + 400]
    a = a[a != 0]
    trips = len(a) // NS
    a = a[:trips * NS].reshape(trips, NS).astype(np.float64)
    flat = a.reshape(-1)
    dd = np.diff(flat)
    per = np.full((trips, NS), np.nan)
    per.reshape(-1)[:len(dd)] = dd  # per[i, j] = time from stamp j of trip i to the next stamp
    print(wsel, "trips", trips, "total cycles", flat[-1] - flat[0], "mean cycles per trip", (flat[-NS] - flat[0]) / max(trips - 1, 1))
    mid = per[2:-2]
    for j, nm in enumerate(names):
        print("   %-16s mean %7.0f  min %7.0f  max %7.0f" % (nm, np.nanmean(mid[:, j]), np.nanmin(mid[:, j]), np.nanmax(mid[:, j])))
