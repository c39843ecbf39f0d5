#!/usr/bin/env python3
"""Timeline of pair_cov_kernel's trips from a -DASSX_PROBE_BUILD -DPAIRCOV_TRACE=1 build (csrc/assx_widem_cov.hpp): shader-clock stamps of
waves 0 and 5 of workgroup 100 at 6 points of every trip, printed as the mean cycles between consecutive points."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd import _lib  # noqa: E402
from audio_source_separation_amd.ops import Engine  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = 4
eng = Engine("float64")
B, F, T = 1, 1025, 4096
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
     1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).contiguous()
W = (torch.eye(M, dtype=torch.complex128, device=eng.dev)[None, None] + torch.zeros((B, F, M, M), dtype=torch.complex128, device=eng.dev)).contiguous()
Tb = torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
V = torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 1400)()
st = eng.new_status(B)
for _ in range(3):
    eng.ilrma_spatial_update(X, W.clone(), Tb, V, domain=2, status=st)
torch.cuda.synchronize()
lib.assx_debug_paircov_trace(buf, 1)
eng.ilrma_spatial_update(X, W.clone(), Tb, V, domain=2, status=st)
torch.cuda.synchronize()
lib.assx_debug_paircov_trace(buf, 0)
st_ = np.array(buf[:], dtype=np.int64)
names = ["top->barrier", "barrier->products", "->slices (fan-out, chain, reads, request)", "->lds_wait 0", "->vmcnt", "flush/moves->top"]
NS = len(names)
for wsel, off in (("wave0", 0), ("wave5", 700)):
    a = st_[off:off + 700]
    a = a[a != 0]
    if len(a) < 2 * NS:
        continue
    trips = len(a) // NS
    a = a[:trips * NS].reshape(trips, NS).astype(np.float64)
    flat = a.reshape(-1)
    dd = np.diff(flat)
    per = np.full((trips, NS), np.nan)
    per.reshape(-1)[:len(dd)] = dd
    print(wsel, "trips", trips, "total", flat[-1] - flat[0], "mean per trip", (flat[-NS] - flat[0]) / max(trips - 1, 1))
    mid = per[2:-2]
    for j, nm in enumerate(names):
        print("   %-34s mean %7.0f  min %7.0f  max %7.0f" % (nm, np.nanmean(mid[:, j]), np.nanmin(mid[:, j]), np.nanmax(mid[:, j])))
