#!/usr/bin/env python3
"""Time assx_ilrma_source_update_partitioned at config-4 size (M=4, F=1025, T=4096) for a given n_basis."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd.ops import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = Engine("float64")
M, F, T = 4, 1025, 4096
X = torch.randn((1, M, F, T), dtype=torch.complex128, device=eng.dev)
W = torch.eye(M, dtype=torch.complex128, device=eng.dev).repeat(1, F, 1, 1).contiguous()
Z = torch.full((1, M, K), 1.0 / M, dtype=torch.float64, device=eng.dev)
Tb = torch.rand((1, F, K), dtype=torch.float64, device=eng.dev) + 0.1
V = torch.rand((1, K, T), dtype=torch.float64, device=eng.dev) + 0.1
Teff = torch.empty((1, M, F, K), dtype=torch.float64, device=eng.dev)
Veff = torch.empty((1, M, K, T), dtype=torch.float64, device=eng.dev)
fn = lambda: eng.ilrma_source_update_partitioned(X, W, Z, Tb, V, Teff, Veff)
for _ in range(3): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fn()
torch.cuda.synchronize()
print("ilrma_source_update_partitioned K=%d: %.1f us" % (K, (time.perf_counter() - t0) / 10 * 1e6))
