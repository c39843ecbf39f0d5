#!/usr/bin/env python3
"""AuxLaplaceIVA update_once on BASELINE config 3 (M=2, F=1025, T=2048) and on the config-4 shape (M=4, T=4096)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd.bss.iva import AuxLaplaceIVA
dev = torch.device("cuda", 0)
for M, F, T in ((2, 1025, 2048), (4, 1025, 4096)):
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((M, F, T), dtype=torch.complex128, device=dev, generator=g)
    m = AuxLaplaceIVA(recordable_loss=False)
    m.input = X
    m._reset()
    for _ in range(5): m.update_once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): m.update_once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("AuxLaplaceIVA M=%d F=%d T=%d: %.1f us/iteration = %.0f it/s, %.2f TB/s over 2 passes" % (M, F, T, dt * 1e6, 1 / dt, 2 * X.numel() * 16 / dt / 1e12))
