#!/usr/bin/env python3
"""pair_cov_kernel against src_cov_kernel at full size: U of one spatial update per M, saved per mode; compare mode prints
the bins that differ.   paircov_check.py run out.npz | paircov_check.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k] - b[k]).max(axis=(-1, -2)) / np.abs(b[k]).max(axis=(-1, -2))  # (1, N, F)
        bad = np.argwhere(d > (1e-10 if a[k].dtype == np.complex128 else 1e-3))
        print(k, "max rel", d.max(), "bad (n,f) count", len(bad), "of", d.size)
        if len(bad):
            fs = sorted(set(int(x[2]) for x in bad))
            ns = sorted(set(int(x[1]) for x in bad))
            print("   sources", ns, "bins", fs[:40], "..." if len(fs) > 40 else "")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.ops import Engine  # noqa: E402
DT = sys.argv[3] if len(sys.argv) > 3 else "float64"
eng = Engine(dtype=DT)
CD = torch.complex128 if DT == "float64" else torch.complex64
RD = torch.float64 if DT == "float64" else torch.float32
g = torch.Generator(device=eng.dev).manual_seed(5)
out = {}
F, T, K = 1025, 4096, 4
for M in (5, 6, 7, 8):
    X = torch.view_as_complex(torch.randn((1, M, F, T, 2), dtype=torch.float64, device=eng.dev, generator=g)).to(CD).contiguous()
    W = (torch.eye(M, dtype=CD, device=eng.dev).expand(1, F, M, M) + 0).contiguous()
    Tb = (torch.rand((1, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(RD)
    V = (torch.rand((1, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(RD)
    for rep in range(2):
        U = eng.empty((1, M, F, M, M), complex_=True)
        eng.ilrma_spatial_update(X, W.clone(), Tb, V, domain=2, status=eng.new_status(1), U_out=U)
        out["U%d_rep%d" % (M, rep)] = U.cpu().numpy()
    r_nt = (torch.rand((1, M, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(RD)
    r_nft = (torch.rand((1, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(RD)
    out["Unt%d" % M] = eng.cov_accumulate(X, r_nt).cpu().numpy()
    out["Unft%d" % M] = eng.cov_accumulate(X, r_nft).cpu().numpy()
np.savez(sys.argv[2], **out)
