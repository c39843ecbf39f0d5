#!/usr/bin/env python3
"""Distribution of the Frobenius condition product c2 = ||W U_n||_F^2 ||(W U_n)^-1||_F^2 over (bin, source) on the headline
bench input, at several points of a run (what a closed-form inverse in the IP sweep would have to cope with)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from audio_source_separation_amd.bss.ilrma import GaussILRMA
dev = torch.device("cuda", 0)
M, F, T, K = 4, 1025, 4096, 4
X = bench.synth_mixture(torch, dev, 1, M, F, T, seed=1000).to(torch.complex128).contiguous()
np.random.seed(111)
m = GaussILRMA(n_basis=K, recordable_loss=False, device=dev)
m.input = X
m._reset()
eng = m._engine
edges = [0, 1e2, 1e4, 1e6, 1e8, 1e10, 1e12, 1e14, 1e16, 1e20, 1e30, float("inf")]
done = 0
for upto in (0, 1, 5, 20, 100, 500):
    while done < upto:
        m.update_once(); done += 1
    W = m._Wd.clone()
    U = eng.empty((1, M, F, M, M), complex_=True)
    W2 = W.clone()
    eng.ilrma_spatial_update(X, W2, m._Td, m._Vd, domain=2, status=eng.new_status(1), U_out=U)
    A = torch.einsum("fij,nfjk->nfik", W[0], U[0])           # (N, F, M, M): W U_n with the filters before the sweep
    Ai = torch.linalg.inv(A)
    c2 = (A.abs() ** 2).sum((-1, -2)) * (Ai.abs() ** 2).sum((-1, -2))
    c2 = c2.flatten().cpu().numpy()
    h = np.histogram(c2, bins=edges)[0]
    sv = torch.linalg.svdvals(A).cpu().numpy().reshape(-1, M)
    two_small = ((sv[:, 0] / sv[:, 2]) > 1e3).sum()
    print("after %3d iterations: c2 histogram over %d (bin, source) pairs, edges %s:\n   %s   max %.3g   pairs with sigma1/sigma3 > 1e3: %d"
          % (upto, c2.size, ["%.0e" % e for e in edges[1:-1]], h.tolist(), c2.max(), two_small))
