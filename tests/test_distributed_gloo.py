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


def _run_world(world, n_items):
    """`world` gloo processes: scatter -> OracleILRMA on every rank's block -> gather, the raw edges, the rank-ordered sum
    and the max-over-ranks reduction; everything compared with one process, bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    saved = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"  # inherited by the spawned ranks only: 8 ranks on 8 cores, one thread each
    try:
        for p in procs:
            p.start()
    finally:  # never left in this process: the fixture-reproducibility test runs the reference with the default BLAS threads
        if saved is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = saved
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    root = [r for r in results if len(r) == 4][0]
    others = [r for r in results if len(r) == 3]
    assert len(others) == world - 1
    y, z, tmax, s = root
    X = _mixtures(n_items)
    single = OracleILRMA()
    _init_fn(single, 0, n_items)
    ref = single(torch.from_numpy(X), iteration=3)
    assert np.array_equal(y, ref)  # sharded == single process, bit for bit, original order
    assert np.array_equal(z, X * 2)
    assert tmax == float(world) and all(o[1] == float(world) for o in others)
    sizes = D.shard_sizes(n_items, world)
    assert sorted(o[0] for o in others) == sorted(sizes[1:])  # every peer worked on a block of the static partition
    # rank-ordered sum: the same bits on every rank, equal to adding the partials in rank order
    expect = np.array([0.1, 1e-17])
    for r in range(1, world):
        expect = expect + np.array([0.1 * (r + 1), 1e-17 * (r + 1)])
    assert np.array_equal(s, expect) and all(np.array_equal(o[2], expect) for o in others)


@pytest.mark.parametrize("n_items", [4, 5, 1])
def test_two_process_scatter_separate_gather(n_items):
    _run_world(2, n_items)


@pytest.mark.parametrize("n_items", [64, 13, 5])
def test_eight_process_scatter_separate_gather(n_items):
    """BASELINE config 5's process layout for real (round 5's review, item 4): EIGHT gloo ranks, the root scatters 64 / 13 / 5
    utterances with one grouped batch of sends, every rank runs its block (8 each; 2,2,2,2,2,1,1,1; five ranks with one item
    and three with none), the root gathers with one grouped batch of receives.  Until round 6 the 8-rank edges were only
    checked as operation lists against a mocked P2POp (test_edge_operation_lists_for_eight_ranks, kept)."""
    _run_world(8, n_items)


# ---------------------------------------------------------------------------------------------------------------
# F-sharded single utterance (bss/ilrma_fshard.py): the real driver + reductions, NumPy stand-in for the shard steps
# ---------------------------------------------------------------------------------------------------------------
class OracleShardOps:
    """The per-shard steps of FrequencyShardedGaussILRMA restated with the oracle (CPU tensors, float64)."""
    device = torch.device("cpu")
    real, cplx = torch.float64, torch.complex128

    def cov(self, X):
        x = X[0].numpy()
        A = x.transpose(1, 0, 2)
        return torch.from_numpy((A @ A.conj().transpose(0, 2, 1) / x.shape[2])[None])

    def power_map(self, X, W):
        return torch.from_numpy(np.abs(orc.separate(X[0].numpy(), W[0].numpy()))[None] ** 2)

    def half_sums(self, half, P, Tb, V, domain, eps):
        P, T, Vv = P[0].numpy(), Tb[0].numpy(), V[0].numpy()
        TV = T @ Vv
        TV[TV < eps] = eps
        division, TVinv = P / TV ** ((domain + 2) / domain), 1 / TV
        if half == 0:
            Vt = Vv.transpose(0, 2, 1)
            num, den = division @ Vt, TVinv @ Vt
        else:
            Tt = T.transpose(0, 2, 1)
            num, den = Tt @ division, Tt @ TVinv
        N = P.shape[0]
        return torch.from_numpy(np.stack([num.reshape(N, -1), den.reshape(N, -1)]))

    def apply_sums(self, A, sums, domain, eps):
        num, den = sums[0].numpy().reshape(A.shape), sums[1].numpy().reshape(A.shape).copy()
        den[den < eps] = eps
        A.mul_(torch.from_numpy((num / den) ** (domain / (domain + 2))))

    def spatial(self, X, W, Tb, V, C, domain, eps, threshold, status):
        Wn, _, _ = orc.ilrma_spatial_update_ip(X[0].numpy(), W[0].numpy().copy(), Tb[0].numpy(), V[0].numpy(), domain, eps,
                                               threshold)
        W[0].copy_(torch.from_numpy(Wn))

    def shard_power_mean(self, C, W, n_frames):
        Wn, Cn = W[0].numpy(), C[0].numpy()
        return torch.from_numpy(np.einsum("fnm,fml,fnl->n", Wn, Cn, Wn.conj()).real.copy() / Wn.shape[0])

    def ordered_sum(self, parts, weights=None):
        acc = parts[0].clone() * (1.0 if weights is None else weights[0])
        for s in range(1, parts.shape[0]):
            acc += parts[s] * (1.0 if weights is None else weights[s])
        return acc

    def normalize(self, W, Tb, power, domain, eps):
        a = np.sqrt(power[0].numpy())
        a[a < eps] = eps
        W[0].div_(torch.from_numpy(a)[None, :, None])
        Tb[0].div_(torch.from_numpy(a ** domain)[:, None, None])

    def loss(self, X, W, Tb, V, domain, eps):
        return torch.tensor([orc.ilrma_loss(X[0].numpy(), W[0].numpy(), Tb[0].numpy(), V[0].numpy(), domain, eps)])

    def output(self, X, W, ref, status):
        Y = orc.separate(X[0].numpy(), W[0].numpy())
        return torch.from_numpy((Y * orc.projection_back(Y, X[0].numpy()[ref])[..., None])[None])

    def new_status(self):
        return torch.zeros(1, dtype=torch.int32)

    def check(self, status):
        pass


FS_M, FS_F, FS_T, FS_K = 3, 11, 40, 2


def _fs_problem():
    rng = np.random.default_rng(21)
    S = (rng.standard_normal((FS_M, FS_F, FS_T)) + 1j * rng.standard_normal((FS_M, FS_F, FS_T))) * \
        rng.random((FS_M, 1, FS_T)) ** 2
    A = rng.standard_normal((FS_F, FS_M, FS_M)) + 1j * rng.standard_normal((FS_F, FS_M, FS_M))
    X = np.einsum("fmn,nft->mft", A, S)
    st = np.random.RandomState(3)
    return X, st.rand(FS_M, FS_F, FS_K), st.rand(FS_M, FS_K, FS_T)


def _fs_run(n_shards, ops):
    from audio_source_separation_amd.bss.ilrma_fshard import FrequencyShardedGaussILRMA
    X, T0, V0 = _fs_problem()
    m = FrequencyShardedGaussILRMA(n_basis=FS_K, n_shards=n_shards, ops=ops, comm_device="cpu")
    Y = m(X, iteration=3, basis=T0, activation=V0)
    return Y, np.asarray(m.loss), m.demix_filter, m.basis, m.activation


def _fs_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    out = _fs_run(world, OracleShardOps())
    D.barrier()
    q.put((rank,) + out)
    torch.distributed.destroy_process_group()


def test_frequency_sharded_ilrma_two_ranks_equal_two_shards_bitwise():
    """bins split over 2 gloo ranks == the same 2 shards on one process, bit for bit (Y, loss, W, basis, activation);
    and both agree with the unsharded oracle up to the rounding of the f-reduction's association."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _fs_run(2, OracleShardOps())
    for res in results:  # every rank returns the whole result
        for a, b in zip(res[1:], single):
            assert np.array_equal(a, b)
    X, T0, V0 = _fs_problem()
    ref = orc.gauss_ilrma(X, 3, T0, V0)
    one = _fs_run(1, OracleShardOps())
    for got in (single, one, _fs_run(3, OracleShardOps())):
        assert np.linalg.norm(got[0] - ref["Y"]) / np.linalg.norm(ref["Y"]) < 1e-10
        np.testing.assert_allclose(got[1], ref["loss"], rtol=1e-12)
        assert np.linalg.norm(got[2] - ref["W"]) / np.linalg.norm(ref["W"]) < 1e-10
        assert np.linalg.norm(got[3] - ref["T"]) / np.linalg.norm(ref["T"]) < 1e-11


def test_edge_operation_lists_for_eight_ranks(monkeypatch):
    """What `scatter_utterances` / `gather_utterances` hand to `batch_isend_irecv` on a node of 8 ranks -- peers, shapes,
    dtypes, contiguity and aliasing of the operands -- checked without a process group (the first real 8-GPU run must not
    die on a non-contiguous `view_as_real` view): the root sends / receives ONE op per peer on a view of its own array,
    a peer posts exactly one op of its block's shape, nobody talks to itself, empty blocks post nothing."""
    from audio_source_separation_amd import distributed as D

    class Op:  # stands in for dist.P2POp (whose constructor needs an initialised group)
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    sent = []
    monkeypatch.setattr(D.dist, "P2POp", Op)
    monkeypatch.setattr(D, "_run_p2p", lambda ops: sent.append(list(ops)))
    for n_items, dtype in ((64, torch.complex128), (13, torch.complex64), (5, torch.complex128)):
        item = (4, 9, 20)
        x_all = torch.view_as_complex(torch.arange(n_items * 4 * 9 * 20 * 2, dtype=torch.float64).reshape(
            (n_items,) + item + (2,))).to(dtype)
        real = x_all.real.dtype
        for rank in range(8):
            monkeypatch.setattr(D, "_world", lambda rank=rank: (rank, 8))
            lo, hi = D.shard_range(n_items, 8, rank)
            sent.clear()
            got = D.scatter_utterances(x_all if rank == 0 else None, n_items, item, dtype, "cpu")
            ops = sent[0] if sent else []
            if rank == 0:
                peers = [r for r in range(1, 8) if D.shard_range(n_items, 8, r)[1] > D.shard_range(n_items, 8, r)[0]]
                assert [o.peer for o in ops] == peers and all(o.op is D.dist.isend for o in ops)
                for o in ops:
                    a, b = D.shard_range(n_items, 8, o.peer)
                    assert o.tensor.shape == (b - a,) + item + (2,) and o.tensor.dtype == real and o.tensor.is_contiguous()
                    assert o.tensor.data_ptr() == x_all[a:b].data_ptr()  # a view of the root's array, not a copy
                assert got.data_ptr() == x_all[lo:hi].data_ptr()
            elif hi > lo:
                assert len(ops) == 1 and ops[0].op is D.dist.irecv and ops[0].peer == 0
                assert ops[0].tensor.shape == (hi - lo,) + item + (2,) and ops[0].tensor.dtype == real
                assert ops[0].tensor.is_contiguous() and ops[0].tensor.data_ptr() == got.data_ptr()
            else:
                assert ops == [] and got.shape == (0,) + item
            # gather: the mirror image
            sent.clear()
            y_local = x_all[lo:hi].clone()
            out = D.gather_utterances(y_local, n_items)
            ops = sent[0] if sent else []
            if rank == 0:
                assert [o.peer for o in ops] == peers and all(o.op is D.dist.irecv for o in ops)
                for o in ops:
                    a, b = D.shard_range(n_items, 8, o.peer)
                    assert o.tensor.shape == (b - a,) + item + (2,) and o.tensor.is_contiguous()
                    assert o.tensor.data_ptr() == out[a:b].data_ptr()  # received straight into the result
                assert torch.equal(out[lo:hi], y_local)
            elif hi > lo:
                assert out is None and len(ops) == 1 and ops[0].op is D.dist.isend and ops[0].peer == 0
                assert ops[0].tensor.shape == (hi - lo,) + item + (2,) and ops[0].tensor.is_contiguous()
            else:
                assert out is None and ops == []
