"""The N>1 path on CPU: world_size-2 gloo processes exercising the utterance sharding, the scatter/gather edges
and the max-over-ranks timing reduction used by bench.py.  The per-utterance work is a stand-in function (the HIP
kernels need a GPU); what is checked is that sharding + collectives reproduce the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from audio_source_separation_amd import distributed as D


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


def _standin(x):
    """Per-utterance 'separation' stand-in: independent per item, like the real path."""
    return torch.flip(x, dims=(1,)) * (1.0 + 0.5j) + x.abs().mean(dim=(1, 2, 3), keepdim=True)


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    shape = (2, 5, 6)
    gen = torch.Generator().manual_seed(0)
    x_all = None
    if rank == 0:
        x_all = torch.randn((n_items,) + shape, dtype=torch.float64, generator=gen) + \
            1j * torch.randn((n_items,) + shape, dtype=torch.float64, generator=gen)
    y = D.run_sharded(_standin, x_all, n_items, shape, torch.complex128, "cpu")
    lo, hi = D.shard_range(n_items, world, rank)
    tmax = D.max_over_ranks(1.0 + rank)
    D.barrier()
    if rank == 0:
        q.put((y.numpy(), x_all.numpy(), tmax))
    else:
        assert y is None
        q.put((hi - lo, tmax))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 5])
def test_two_process_scatter_process_gather(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in results if len(r) == 3][0]
    other = [r for r in results if len(r) == 2][0]
    y, x_all, tmax = root
    ref = _standin(torch.from_numpy(x_all)).numpy()
    assert np.array_equal(y, ref)  # sharded == single process, bit for bit, original order
    assert tmax == 2.0 and other[1] == 2.0
    assert other[0] == D.shard_range(n_items, world, 1)[1] - D.shard_range(n_items, world, 1)[0]
