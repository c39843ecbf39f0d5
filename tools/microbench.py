#!/usr/bin/env python3
"""Time individual C-ABI entry points on the config-4 workload (HIP events on the launch stream)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd import _lib  # noqa: E402
from audio_source_separation_amd.ops import Engine  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--dtype", default="float64")
p.add_argument("--B", type=int, default=1)
p.add_argument("--M", type=int, default=4)
p.add_argument("--F", type=int, default=1025)
p.add_argument("--T", type=int, default=4096)
p.add_argument("--K", type=int, default=4)
p.add_argument("--reps", type=int, default=30)
p.add_argument("--only", default="")
a = p.parse_args()

eng = Engine(a.dtype)
B, M, F, T, K = a.B, a.M, a.F, a.T, a.K
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
     1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
W = (torch.eye(M, dtype=torch.complex128, device=eng.dev).repeat(B, F, 1, 1) +
     0.1 * torch.randn((B, F, M, M), dtype=torch.complex128, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
Tb = torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real) + 0.1
V = torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real) + 0.1
r_nt = torch.rand((B, M, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real) + 0.1
r_nft = torch.rand((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real) + 0.1
st = eng.new_status(B)
xbytes = X.numel() * X.element_size()


def timeit(name, fn, nbytes=xbytes):
    if a.only and a.only not in name:
        return
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("%-34s %8.1f us   %7.1f GB/s (X bytes)" % (name, ms * 1e3, nbytes / ms / 1e6))


timeit("cov unweighted (partial+finalize)", lambda: eng.cov_accumulate(X))
timeit("cov NT weights", lambda: eng.cov_accumulate(X, r_nt))
timeit("cov NFT weights", lambda: eng.cov_accumulate(X, r_nft))
timeit("cov TV partial only", lambda: eng.ilrma_cov_partials(X, Tb, V))
timeit("ilrma_spatial_update (cov+fin+ip)", lambda: eng.ilrma_spatial_update(X, W, Tb, V, status=st))
Tb2, V2 = Tb.clone(), V.clone()
timeit("ilrma_source_update (4 kernels)", lambda: eng.ilrma_source_update(X, W, Tb2, V2))
timeit("demix_power", lambda: eng.demix_power(X, W))
timeit("ilrma_loss", lambda: eng.ilrma_loss(X, W, Tb, V))
timeit("auxiva_weights (laplace, +loss)", lambda: eng.auxiva_weights(X, W, _lib.IVA_LAPLACE, with_loss=True))
timeit("projection_back_scale", lambda: eng.projection_back_scale(X, W, 0, st))
Y = eng.empty((B, M, F, T), complex_=True)
timeit("demix (read X, write Y)", lambda: eng.demix(X, W, out=Y), 2 * xbytes)
U = eng.cov_accumulate(X, r_nt)
W2 = W.clone()
timeit("ip_update", lambda: eng.ip_update(U, W2, 1e12, st), U.numel() * U.element_size())
# BASELINE config 2 (IS-NMF F x T, n_basis 32) through the same harness, for the PMC scripts
if a.only.startswith("nmf"):
    Kn = a.K if a.K > 4 else 32
    Xn = (torch.rand((1, F, T), dtype=torch.float64, device=eng.dev, generator=g) ** 2).to(eng.prec.real)
    Tn = torch.rand((1, F, Kn), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
    Vn = torch.rand((1, Kn, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
    timeit("nmf_update IS", lambda: eng.nmf_update(_lib.NMF_IS_MM, Xn, Tn, Vn), 2 * F * T * Xn.element_size())
