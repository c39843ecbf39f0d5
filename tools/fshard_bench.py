#!/usr/bin/env python3
"""F-sharded single-utterance mode (bss/ilrma_fshard.py) at config-4 size on ONE GPU: time per update_once for
n_shards in {1, 2, 4, 8} next to the unsharded GaussILRMA class -- what splitting the utterance into bin shards costs
before any transport is involved (the split pieces, the ordered sums, the per-shard launches).  No scaling claim: every
shard runs on the same GPU, one after the other.

    python tools/fshard_bench.py [--basis 4] [--iters 30]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd.bss.ilrma import GaussILRMA  # noqa: E402
from audio_source_separation_amd.bss.ilrma_fshard import FrequencyShardedGaussILRMA  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--basis", type=int, default=4)
p.add_argument("--iters", type=int, default=30)
a = p.parse_args()
M, F, T, K = 4, 1025, 4096, a.basis
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
S = torch.view_as_complex(torch.randn((M, F, T, 2), dtype=torch.float64, device=dev, generator=g))
env = torch.rand((M, 1, T), dtype=torch.float64, device=dev, generator=g) ** 2
A = torch.view_as_complex(torch.randn((F, M, M, 2), dtype=torch.float64, device=dev, generator=g))
X = torch.einsum("fmn,nft->mft", A, S * env).contiguous()
st = np.random.RandomState(1)
T0, V0 = st.rand(M, F, K), st.rand(M, K, T)
out = {"workload": "M=%d F=%d T=%d K=%d float64, loss off, one MI355X" % (M, F, T, K)}


def timed(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = GaussILRMA(n_basis=K, recordable_loss=False)
ref.basis, ref.activation = T0, V0
ref.input = X
ref._reset()
out["unsharded_class_ms_per_iteration"] = round(timed(ref.update_once, a.iters), 4)
for S_ in (1, 2, 4, 8):
    m = FrequencyShardedGaussILRMA(n_basis=K, recordable_loss=False, n_shards=S_)
    m(X, iteration=0, basis=T0, activation=V0)  # builds the shards (uploads, plain covariance)
    ms = timed(m.update_once, a.iters)
    out["n_shards_%d" % S_] = {"ms_per_iteration": round(ms, 4), "iterations_per_s": round(1e3 / ms, 1),
                               "vs_unsharded": round(ms / out["unsharded_class_ms_per_iteration"], 3)}
print(json.dumps(out, indent=1))
