#!/usr/bin/env python3
"""Time assx_nmf_update / assx_nmf_loss on BASELINE config 2 (IS-NMF F=1025, T=4096, K=32)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd import _lib
from audio_source_separation_amd.ops import Engine
dtype = sys.argv[1] if len(sys.argv) > 1 else "float64"
F, T, K = 1025, 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = Engine(dtype)
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.rand((1, F, T), dtype=torch.float64, device=eng.dev, generator=g) ** 2).to(eng.prec.real)
Tb = torch.rand((1, F, K), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
V = torch.rand((1, K, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
for name, fn in (("nmf_update IS", lambda: eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V)),
                 ("nmf_loss IS", lambda: eng.nmf_loss(_lib.NMF_IS_MM, X, Tb, V)),
                 ("nmf_update EUC", lambda: eng.nmf_update(_lib.NMF_EUC, X, Tb, V)),
                 ("nmf_update KL", lambda: eng.nmf_update(_lib.NMF_KL, X, Tb, V)),
                 ("nmf_update IS d=1", lambda: eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V, domain=1)),
                 ("nmf_update t", lambda: eng.nmf_update(_lib.NMF_T, X, Tb, V, param=4.0)),
                 ("nmf_update Cauchy mm", lambda: eng.nmf_update(_lib.NMF_CAUCHY_MM, X, Tb, V)),
                 ("nmf_update Cauchy me", lambda: eng.nmf_update(_lib.NMF_CAUCHY_ME, X, Tb, V)),
                 ("nmf_update Cauchy mmf", lambda: eng.nmf_update(_lib.NMF_CAUCHY_MM_FAST, X, Tb, V)),
                 ("nmf_loss t", lambda: eng.nmf_loss(_lib.NMF_T, X, Tb, V, param=4.0))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    # flops: an update is six F x T x K products (two reconstructions, four contractions) = 12 F T K; a loss evaluation is
    # ONE reconstruction = 2 F T K (+ the elementwise criterion)
    mult = 2 if "loss" in name else 12
    print("%-22s %8.1f us  %6.1f TFLOP/s (%dFTK)" % (name, ms * 1e3, mult * F * T * K / ms / 1e9, mult))
