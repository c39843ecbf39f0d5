"""The N>1 path on CPU: world_size-2 gloo processes exercising the utterance sharding, the grouped send/recv edges,
`separate_sharded` (the config-5 driver) and the max-over-ranks timing reduction used by bench.py.

The HIP kernels need a GPU, so the per-utterance work on CPU is a deterministic NumPy stand-in with the model-class
surface: `OracleILRMA.__call__(X (B,M,F,T), iteration)` runs the oracle's Gauss-ILRMA `update_once` loop on every
utterance of its block.  What is checked is the real partition logic: sharded == single process, bit for bit, in
the original utterance order, for even and ragged splits (the 2-rank run with the real GaussILRMA on two GPUs is
tests/test_gpu_multi.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from audio_source_separation_amd import distributed as D
from oracle import oracle_np as orc  # checker / stand-in only

M, F, T, K = 2, 6, 16, 2


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = D.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            sizes = D.shard_sizes(n, world)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
    assert D.shard_range(64, 8, 3) == (24, 32)  # config 5: utt i -> GPU i // 8
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init_state(u):
    """Initial (basis, activation) of utterance u: drawn per UTTERANCE so the result cannot depend on the shard."""
    rng = np.random.RandomState(111 + u)
    return rng.rand(M, F, K), rng.rand(M, K, T)


class OracleILRMA:
    """NumPy stand-in with the GaussILRMA call surface (batched input, `basis` / `activation` warm start)."""

    def __call__(self, X, iteration=3):
        X = X.numpy()
        out = np.empty_like(X)
        for b in range(X.shape[0]):
            out[b] = orc.gauss_ilrma(X[b], iteration, self.basis[b], self.activation[b], record_loss=False)["Y"]
        return out


def _init_fn(model, lo, hi):
    st = [_init_state(u) for u in range(lo, hi)]
    model.basis = np.stack([s[0] for s in st])
    model.activation = np.stack([s[1] for s in st])


def _mixtures(n_items):
    rng = np.random.default_rng(5)
    return rng.standard_normal((n_items, M, F, T)) + 1j * rng.standard_normal((n_items, M, F, T))


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    x_all = torch.from_numpy(_mixtures(n_items)) if rank == 0 else None
    y, model = D.separate_sharded(OracleILRMA, x_all, n_items, (M, F, T), torch.complex128, "cpu", iteration=3,
                                  init_fn=_init_fn)
    # the raw edges as well: scatter -> identity -> gather must reproduce the root's array exactly
    z = D.run_sharded(lambda x: x * 2, x_all, n_items, (M, F, T), torch.complex128, "cpu")
    s = D.all_gather_ordered_sum(torch.tensor([0.1 * (rank + 1), 1e-17 * (rank + 1)], dtype=torch.float64))
    lo, hi = D.shard_range(n_items, world, rank)
    tmax = D.max_over_ranks(1.0 + rank)
    D.barrier()
    if rank == 0:
        q.put((y.numpy(), z.numpy(), tmax, s.numpy()))
    else:
        assert y is None and z is None
        q.put((hi - lo, tmax, s.numpy()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 5, 1])
def test_two_process_scatter_separate_gather(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in results if len(r) == 4][0]
    other = [r for r in results if len(r) == 3][0]
    y, z, tmax, s = root
    X = _mixtures(n_items)
    single = OracleILRMA()
    _init_fn(single, 0, n_items)
    ref = single(torch.from_numpy(X), iteration=3)
    assert np.array_equal(y, ref)  # sharded == single process, bit for bit, original order
    assert np.array_equal(z, X * 2)
    assert tmax == 2.0 and other[1] == 2.0
    assert other[0] == D.shard_range(n_items, world, 1)[1] - D.shard_range(n_items, world, 1)[0]
    # rank-ordered sum: the same bits on both ranks, equal to adding the partials in rank order
    expect = np.array([0.1, 1e-17]) + np.array([0.2, 2e-17])
    assert np.array_equal(s, expect) and np.array_equal(other[2], expect)
