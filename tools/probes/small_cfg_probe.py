#!/usr/bin/env python3
"""BASELINE configs 1 and 3 (the small ones) through the classes, one dtype per process: update_once per second, and --
under rocprofv3 --kernel-trace --stats -- where the time goes.   small_cfg_probe.py cfg1|cfg3 float64|float32 [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.algorithm.nmf import EUCNMF  # noqa: E402
from audio_source_separation_amd.bss.iva import AuxLaplaceIVA  # noqa: E402

cfg, dtype = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
if cfg == "cfg1":
    X = torch.rand((513, 256), dtype=torch.float64, device=dev, generator=g) ** 2
    np.random.seed(0)
    m = EUCNMF(n_basis=8, dtype=dtype)
    m.target = X
else:
    M, F, T = 2, 1025, 2048
    S = torch.randn((M, F, T), dtype=torch.float64, device=dev, generator=g) + 1j * torch.randn((M, F, T), dtype=torch.float64, device=dev, generator=g)
    A = torch.randn((F, M, M), dtype=torch.complex128, device=dev, generator=g)
    X = torch.einsum("fmn,nft->mft", A, S).contiguous()
    m = AuxLaplaceIVA(recordable_loss=False, dtype=dtype)
    m.input = X.to(torch.complex128 if dtype == "float64" else torch.complex64)
m._reset()
for _ in range(20):
    m.update_once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.update_once()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s %s: %.1f update_once/s (%.1f us each; host loop alone %.1f us each)" % (cfg, dtype, steps / (t2 - t0), (t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6))
# the same iterations as ONE library call (round 4: assx_nmf_iterate / assx_auxiva_iterate, what __call__ uses when no
# callback has to run between iterations); loss evaluation off, like the loop above
eng = m._engine
torch.cuda.synchronize()
t0 = time.perf_counter()
if cfg == "cfg1":
    eng.nmf_iterate(steps, m._kind_code(), m._X, m._dev("T", False), m._dev("V", False), domain=m.domain, eps=m.eps)
else:
    r = eng.empty((1, M, T))
    eng.auxiva_iterate(steps, m._KIND, m._X, m._Wd, r, eps=m.eps, threshold=m.threshold, status=m._status)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s %s: %.1f iterations/s in one library call (%.1f us each; enqueueing alone %.1f us each)" % (cfg, dtype, steps / (t2 - t0), (t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6))
