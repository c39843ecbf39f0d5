#!/usr/bin/env python3
"""BASELINE configs 1-3 the way a user runs them: model(X, iteration=n) on a device-resident input, loss recording on
(the reference's default) and off; per-iteration time of the whole call.   call_cfgs.py float64|float32 [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.algorithm.nmf import EUCNMF, ISNMF  # noqa: E402
from audio_source_separation_amd.bss.iva import AuxLaplaceIVA  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "float64"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rd = torch.float64 if dtype == "float64" else torch.float32
cd = torch.complex128 if dtype == "float64" else torch.complex64
X1 = (torch.rand((513, 256), dtype=torch.float64, device=dev, generator=g) ** 2).to(rd)
X2 = (torch.rand((1025, 4096), dtype=torch.float64, device=dev, generator=g) ** 2).to(rd)
S = torch.randn((2, 1025, 2048), dtype=torch.float64, device=dev, generator=g) + 1j * torch.randn((2, 1025, 2048), dtype=torch.float64, device=dev, generator=g)
A = torch.randn((1025, 2, 2), dtype=torch.complex128, device=dev, generator=g)
X3 = torch.einsum("fmn,nft->mft", A, S).contiguous().to(cd)
for name, make, X in (("cfg1 EUC-NMF 513x256 K=8", lambda rl: EUCNMF(n_basis=8, dtype=dtype, recordable_loss=rl), X1),
                      ("cfg2 IS-NMF 1025x4096 K=32", lambda rl: ISNMF(n_basis=32, dtype=dtype, recordable_loss=rl), X2),
                      ("cfg3 AuxLaplaceIVA M=2 1025x2048", lambda rl: AuxLaplaceIVA(dtype=dtype, recordable_loss=rl), X3)):
    for rl in (True, False):
        np.random.seed(0)
        m = make(rl)
        m(X, iteration=20)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            np.random.seed(0)
            m = make(rl)
            t0 = time.perf_counter()
            m(X, iteration=n)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-34s %s loss %-3s: %7.1f us per iteration of model(X, iteration=%d) = %.0f it/s" %
              (name, dtype, "on" if rl else "off", best / n * 1e6, n, n / best))
