"""Parity at BENCHMARK size and on the multi-slab code paths the benchmark sizes exercise.

 * every BASELINE config 2-4 gets ONE oracle step at its quoted size (the oracle costs ~1-2 s of host time each);
 * the NMF matrix-core kernels are forced into >= 8 split-T / split-F slabs per half-update on oracle-sized inputs
   (ASSX_NMF_BASIS_WGS / ASSX_NMF_ACT_WGS), for K in {8, 10, 32, 64} and the K > 64 VALU fallback, f64 and f32;
 * 8 utterances of config 4 in one batched launch sequence == the same utterances one by one, bit for bit.

Tolerances: float64 1e-10 relative Frobenius on the model arrays after one step (summation order only), 1e-8 on W after
an IP sweep at full size (cond(W U) amplification), loss 1e-10; float32 storage 2e-4 / 5e-3.
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle_np as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev_r(eng, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.dev, eng.prec.real).contiguous()


def dev_c(eng, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.dev, eng.prec.cplx).contiguous()


def host(t):
    a = t.detach().cpu().numpy()
    return a.astype(np.complex128) if np.iscomplexobj(a) else a.astype(np.float64)


def mixture(M, F, T, seed):
    rng = np.random.default_rng(seed)
    S = (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))) * rng.random((M, 1, T)) ** 2
    A = rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M))
    return np.einsum("fmn,nft->mft", A, S)


# ------------------------------------------------------------------------------------------------ NMF, many slabs
@pytest.fixture
def nmf_many_slabs():
    os.environ["ASSX_NMF_BASIS_WGS"] = "100000"  # -> one slab per 64 frames (the cap)
    os.environ["ASSX_NMF_ACT_WGS"] = "100000"    # -> one slab per 64 bins
    yield
    os.environ.pop("ASSX_NMF_BASIS_WGS", None)
    os.environ.pop("ASSX_NMF_ACT_WGS", None)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("kind,domain", [("IS", 2), ("IS", 1.5), ("KL", 2), ("EUC", 2)])
@pytest.mark.parametrize("K", [8, 10, 32, 64, 70])
def test_nmf_many_slabs_vs_oracle(nmf_many_slabs, dtype, kind, domain, K):
    """F=530 (9 bin slabs), T=650 (11 frame slabs), ragged against every tile size: 16, 64, 144."""
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    if kind != "IS" and K not in (10, 64):
        pytest.skip("EUC / KL share the kernels: two basis counts are enough")
    eng = Engine(dtype)
    F, T = 530, 650
    rng = np.random.default_rng(K)
    X = rng.random((F, T)) ** 2 + 1e-3
    T0, V0 = rng.random((F, K)), rng.random((K, T))
    code = {"EUC": _lib.NMF_EUC, "KL": _lib.NMF_KL, "IS": _lib.NMF_IS_MM}[kind]
    Xd, Td, Vd = dev_r(eng, X[None]), dev_r(eng, T0[None]), dev_r(eng, V0[None])
    Tr, Vr = T0, V0
    t64, t32 = 1e-11, 2e-4
    for it in range(2):
        eng.nmf_update(code, Xd, Td, Vd, domain=domain)
        Tr, Vr = orc.nmf_update_once(kind, X, Tr, Vr, domain=domain)
        tol = (t64 if dtype == "float64" else t32) * (it + 1)
        assert rel_err(host(Td)[0], Tr) < tol and rel_err(host(Vd)[0], Vr) < tol, (it, K)
    got = eng.nmf_loss(code, Xd, Td, Vd, domain=domain).item()
    np.testing.assert_allclose(got, orc.nmf_loss(kind, X, Tr, Vr, domain=domain), rtol=1e-10 if dtype == "float64" else 5e-4)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("F,T,K", [(530, 650, 10), (33, 4100, 32), (1025, 300, 16), (100, 100, 40)])
@pytest.mark.parametrize("budget", [8, 40, 150, 512, 3000])
def test_nmf_partition_forms_vs_oracle(dtype, F, T, K, budget, monkeypatch):
    """The matrix-core halves under every form of their work partition (round 5): block-aligned with w = 1 ... 16
    workgroups per block (w = 1: the direct update, no tickets), the flat fallback where a block would get fewer than two,
    more blocks than the budget (several rounds of workgroups); a batch of two equals two single calls bit for bit (the
    partition is a function of one matrix's geometry); the partition query agrees with what fits the workspace."""
    import ctypes
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    monkeypatch.setenv("ASSX_NMF_BASIS_WGS", str(budget))  # read on every call
    monkeypatch.setenv("ASSX_NMF_ACT_WGS", str(budget))
    eng = Engine(dtype)
    out = (ctypes.c_int32 * 6)()
    for half in (0, 1):
        assert _lib.lib.assx_nmf_partition_query(0, half, 1, F, T, K, eng.prec.code, out) == 0
        assert 1 <= out[4] <= out[3] <= out[5], list(out)
    rng = np.random.default_rng(F + K + budget)
    X = rng.random((2, F, T)) ** 2 + 1e-3
    T0, V0 = rng.random((2, F, K)) + 0.05, rng.random((2, K, T)) + 0.05
    Xd, Td, Vd = dev_r(eng, X), dev_r(eng, T0), dev_r(eng, V0)
    for it in range(2):
        eng.nmf_update(_lib.NMF_IS_MM, Xd, Td, Vd)
    for b in range(2):
        Tr, Vr = T0[b], V0[b]
        for it in range(2):
            Tr, Vr = orc.nmf_update_once("IS", X[b], Tr, Vr, domain=2)
        tol = 2e-11 if dtype == "float64" else 4e-4
        assert rel_err(host(Td)[b], Tr) < tol and rel_err(host(Vd)[b], Vr) < tol, (b, budget)
        t1, v1 = dev_r(eng, T0[b:b + 1]), dev_r(eng, V0[b:b + 1])
        for it in range(2):
            eng.nmf_update(_lib.NMF_IS_MM, Xd[b:b + 1], t1, v1)
        assert torch.equal(t1[0], Td[b]) and torch.equal(v1[0], Vd[b])


@pytest.mark.parametrize("seed", range(6))
def test_nmf_random_shapes_vs_oracle(seed):
    """Ten random (F, T, n_basis, kind) per seed, shapes down to a single row / column / basis vector, default budgets: the
    matrix-core halves (n_basis > 4) and the small-rank kernels (n_basis <= 4) against the oracle, float64."""
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64")
    rng = np.random.default_rng(9000 + seed)
    for case in range(10):
        F = int(rng.choice([1, 2, 15, 16, 17, 33, int(rng.integers(1, 400))]))
        T = int(rng.choice([1, 3, 16, 64, 65, int(rng.integers(1, 3000))]))
        K = int(rng.choice([1, 2, 4, 5, 16, 17, 33, 64, int(rng.integers(1, 65))]))
        kind = ["EUC", "KL", "IS"][int(rng.integers(0, 3))]
        X = rng.random((F, T)) ** 2 + 1e-3
        T0, V0 = rng.random((F, K)) + 0.05, rng.random((K, T)) + 0.05
        code = {"EUC": _lib.NMF_EUC, "KL": _lib.NMF_KL, "IS": _lib.NMF_IS_MM}[kind]
        Xd, Td, Vd = dev_r(eng, X[None]), dev_r(eng, T0[None]), dev_r(eng, V0[None])
        eng.nmf_update(code, Xd, Td, Vd)
        Tr, Vr = orc.nmf_update_once(kind, X, T0, V0, domain=2)
        assert rel_err(host(Td)[0], Tr) < 1e-11 and rel_err(host(Vd)[0], Vr) < 1e-11, (F, T, K, kind)
        got = eng.nmf_loss(code, Xd, Td, Vd).item()
        np.testing.assert_allclose(got, orc.nmf_loss(kind, X, Tr, Vr, domain=2), rtol=1e-9, atol=1e-9, err_msg=str((F, T, K, kind)))


# ------------------------------------------------------------------------------------------------ config 2
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_config2_isnmf_full_size_oracle_step(dtype):
    """IS-NMF F=1025 T=4096 K=32 (BASELINE config 2): one update_once + loss against the oracle at full size."""
    from audio_source_separation_amd.algorithm.nmf import ISNMF
    F, T, K = 1025, 4096, 32
    rng = np.random.default_rng(2)
    X = rng.random((F, T)) ** 2 + 1e-3
    np.random.seed(3)
    state = np.random.get_state()
    T0, V0 = np.random.rand(F, K), np.random.rand(K, T)
    np.random.set_state(state)
    m = ISNMF(n_basis=K, dtype=dtype)
    Tb, V = m(X, iteration=1)
    Tr, Vr = orc.nmf_update_once("IS", X, T0, V0)
    tol = 1e-10 if dtype == "float64" else 2e-4
    assert rel_err(Tb, Tr) < tol and rel_err(V, Vr) < tol
    np.testing.assert_allclose(np.asarray(m.loss)[-1], orc.nmf_loss("IS", X, Tr, Vr), rtol=1e-10 if dtype == "float64" else 5e-4)


# ------------------------------------------------------------------------------------------------ config 3
@pytest.mark.parametrize("cls_name,kind", [("AuxLaplaceIVA", "laplace"), ("AuxGaussIVA", "gauss")])
def test_config3_auxiva_full_size_oracle_step(cls_name, kind):
    """AuxIVA-IP M=2 F=1025 T=2048 (BASELINE config 3): two update_once sweeps + the losses against the oracle."""
    from audio_source_separation_amd.bss import iva
    M, F, T = 2, 1025, 2048
    X = mixture(M, F, T, 30)
    m = getattr(iva, cls_name)()
    Y = m(X, iteration=2)
    ref = orc.auxiva(X, 2, kind)
    assert rel_err(m.demix_filter, ref["W"]) < 1e-9
    assert rel_err(Y, ref["Y"]) < 1e-9
    np.testing.assert_allclose(np.asarray(m.loss), ref["loss"], rtol=1e-10)


# ------------------------------------------------------------------------------------------------ config 4
def _cfg4_state(seed, M=4, F=1025, T=4096, K=4):
    X = mixture(M, F, T, seed)
    rng = np.random.RandomState(111 + seed)
    return X, rng.rand(M, F, K), rng.rand(M, K, T)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_config4_ilrma_full_size_oracle_step(dtype):
    """Gauss-ILRMA M=4 F=1025 T=4096 K=4 (BASELINE config 4, the headline workload): update_once twice against
    orc.ilrma_update_once -- W, basis, activation, the cond mask (all kept) and the recorded losses."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X, T0, V0 = _cfg4_state(4)
    m = GaussILRMA(n_basis=4, dtype=dtype)
    m.basis, m.activation = T0, V0
    m.input = X
    m._reset()
    W = np.tile(np.eye(4, dtype=np.complex128), (1025, 1, 1))
    Tr, Vr = T0, V0
    losses = [orc.ilrma_loss(X, W, Tr, Vr)]
    m._record_loss()
    for it in range(2):
        m.update_once()
        m._record_loss()
        W, Tr, Vr, mask = orc.ilrma_update_once(X, W, Tr, Vr)
        losses.append(orc.ilrma_loss(X, W, Tr, Vr))
        assert mask.all()
        s = it + 1
        if dtype == "float64":
            assert rel_err(m.basis, Tr) < 1e-10 * s and rel_err(m.activation, Vr) < 1e-10 * s
            assert rel_err(m.demix_filter, W) < 1e-8 * s
        else:
            assert rel_err(m.basis, Tr) < 5e-4 * s and rel_err(m.activation, Vr) < 5e-4 * s
            assert rel_err(m.demix_filter, W) < 5e-3 * s
    m._check_status()
    np.testing.assert_allclose(np.asarray(m.loss), losses, rtol=1e-10 if dtype == "float64" else 2e-4)


def test_config4_default_basis_full_size_oracle_step():
    """The reference's default n_basis = 10 (ilrma.py:183) at config-4 size: the wide-basis kernels (cov_wide_kernel,
    demixed-power map + matrix-core source model) against one oracle step."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X, T0, V0 = _cfg4_state(5, K=10)
    m = GaussILRMA(n_basis=10)
    m.basis, m.activation = T0, V0
    m.input = X
    m._reset()
    m.update_once()
    W = np.tile(np.eye(4, dtype=np.complex128), (1025, 1, 1))
    W, Tr, Vr, mask = orc.ilrma_update_once(X, W, T0, V0)
    assert mask.all()
    assert rel_err(m.basis, Tr) < 1e-10 and rel_err(m.activation, Vr) < 1e-10 and rel_err(m.demix_filter, W) < 1e-8
    np.testing.assert_allclose(m.compute_negative_loglikelihood(), orc.ilrma_loss(X, W, Tr, Vr), rtol=1e-10)


def test_config5_batch_of_8_full_size_equals_single():
    """8 utterances of config 4 per GPU (config 5's per-rank batch) in ONE batched call == each utterance alone,
    bit for bit (W, basis, activation, output, loss), 3 iterations incl. the final projection back."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    M, F, T, K, B = 4, 1025, 4096, 4, 8
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(55)
    X = torch.view_as_complex(torch.randn((B, M, F, T, 2), dtype=torch.float64, device=dev, generator=gen))
    env = torch.rand((B, M, 1, T), dtype=torch.float64, device=dev, generator=gen) ** 2
    A = torch.view_as_complex(torch.randn((B, F, M, M, 2), dtype=torch.float64, device=dev, generator=gen))
    X = torch.einsum("bfmn,bnft->bmft", A, X * env).contiguous()
    st = [np.random.RandomState(500 + b) for b in range(B)]
    T0 = np.stack([s.rand(M, F, K) for s in st])
    V0 = np.stack([s.rand(M, K, T) for s in st])
    mb = GaussILRMA(n_basis=K)
    mb.basis, mb.activation = T0, V0
    Yb = mb(X, iteration=3)
    lossb = np.asarray(mb.loss)  # (4, B)
    for b in (0, 3, 7):
        m1 = GaussILRMA(n_basis=K)
        m1.basis, m1.activation = T0[b], V0[b]
        Y1 = m1(X[b], iteration=3)
        assert torch.equal(Y1, Yb[b])
        assert np.array_equal(m1.demix_filter, mb.demix_filter[b])
        assert np.array_equal(m1.basis, mb.basis[b]) and np.array_equal(m1.activation, mb.activation[b])
        assert np.array_equal(np.asarray(m1.loss), lossb[:, b])


# ------------------------------------------------------------------------------------------------ a9 surface
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("M", [2, 3, 4, 5, 6, 8])
def test_compute_demix_filter(dtype, M):
    """compute_demix_filter (ilrma.py:167-173, iva.py:119-125) vs the oracle; Y = W X must give W back."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.bss.iva import AuxLaplaceIVA
    F, T = 21, 333
    X = mixture(M, F, T, 60 + M)
    rng = np.random.default_rng(61)
    W = np.eye(M)[None] + 0.3 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))
    Y = orc.separate(X, W) + 0.05 * (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T)))
    ref = orc.compute_demix_filter(Y, X)
    tol = 1e-11 if dtype == "float64" else 5e-5
    for model in (GaussILRMA(dtype=dtype), AuxLaplaceIVA(dtype=dtype)):
        got = model.compute_demix_filter(Y, X)
        assert got.shape == (F, M, M) and got.dtype == np.complex128
        assert rel_err(got, ref) < tol
    m = GaussILRMA(dtype=dtype)
    assert rel_err(m.compute_demix_filter(orc.separate(X, W), X), W) < tol * 10
    # batched + device tensors in -> device tensor out
    Xt = torch.from_numpy(np.stack([X, X[:, ::-1].copy()])).cuda()
    Yt = torch.from_numpy(np.stack([Y, Y[:, ::-1].copy()])).cuda()
    Wt = m.compute_demix_filter(Yt, Xt)
    assert isinstance(Wt, torch.Tensor) and tuple(Wt.shape) == (2, F, M, M)
    assert rel_err(Wt[0].cpu().numpy(), ref) < tol and rel_err(Wt[1].cpu().numpy(), ref[::-1]) < tol
    # an exactly singular X X^H raises like numpy.linalg.inv
    for make in (lambda a: a.__setitem__(1, a[0]), lambda a: a.__setitem__(M - 1, 0)):  # duplicated / silent channel
        Xs = X.copy()
        make(Xs)
        with pytest.raises(np.linalg.LinAlgError):
            orc.compute_demix_filter(Y, Xs)
        with pytest.raises(np.linalg.LinAlgError):
            m.compute_demix_filter(Y, Xs)


def test_iss_callback_can_rebuild_the_filter():
    """A reference-style ISS callback: W = model.compute_demix_filter(model.estimation, model.input) agrees with the
    filter the HIP path carries (the reference keeps demix_filter = None during ISS and rebuilds it exactly so)."""
    import warnings
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X = mixture(3, 17, 400, 70)
    seen = []

    def cb(model):
        seen.append(rel_err(model.compute_demix_filter(model.estimation, model.input), model.demix_filter))

    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = GaussILRMA(n_basis=3, algorithm_spatial='ISS', callbacks=cb)
    m(X, iteration=3)
    assert len(seen) == 4 and max(seen[:-1]) < 1e-9


def test_state_edits_in_place_and_by_assignment_reach_the_device():
    """The reference's model arrays are plain NumPy attributes: `model.basis[...] *= s` or `model.demix_filter[f] = w`
    between updates take effect at the next one (ilrma.py:97-104).  Here the arrays live in HBM; the snapshot handed out
    is tracked, an in-place edit (also through a slice) makes the next kernel upload it.  Round 3 refused such edits."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X = mixture(2, 9, 128, 80)

    def run(edit):
        np.random.seed(1)
        m = GaussILRMA(n_basis=2)
        m(X, iteration=1)
        edit(m)
        m.update_once()
        return m.demix_filter.copy(), m.basis.copy(), m.activation.copy()

    def in_place(m):
        m.basis[...] *= 1.5            # ufunc with out=
        m.demix_filter[3] = 2.0 * m.demix_filter[3]   # __setitem__
        row = m.activation[1]          # a view of the snapshot
        row[0, :7] = 0.25

    def by_assignment(m):
        T, W, V = m.basis.copy(), m.demix_filter.copy(), m.activation.copy()
        T *= 1.5
        W[3] = 2.0 * W[3]
        V[1, 0, :7] = 0.25
        m.basis, m.demix_filter, m.activation = T, W, V

    a, b, c = run(in_place), run(by_assignment), run(lambda m: None)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0], c[0]) and not np.array_equal(a[1], c[1])   # the edits did something
    np.random.seed(1)
    m = GaussILRMA(n_basis=2)
    m(X, iteration=1)
    W = m.demix_filter
    Wc = W.copy()
    Wc[0] = 0                       # a copy is the caller's own array
    old = m.demix_filter
    m.update_once()
    old[0] = 0                      # an OLD snapshot: the model has moved on, nothing happens to it
    assert np.abs(m.demix_filter[0]).max() > 0
    m.demix_filter = W * 2.0        # assignment still works
    assert np.array_equal(m.demix_filter, W * 2.0)
    assert rel_err(m.separate(X, m.demix_filter), 2.0 * orc.separate(X, W)) < 1e-12


def test_engine_guards_the_current_device():
    """The C library refuses a call whose context device is not current; the Engine makes it current per call."""
    import ctypes
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64", device="cuda:0")
    X = dev_c(eng, mixture(2, 5, 70, 90)[None])
    Wd = dev_c(eng, np.tile(np.eye(2, dtype=np.complex128), (1, 5, 1, 1)))
    Y = eng.demix(X, Wd)
    assert rel_err(host(Y)[0], host(X)[0]) == 0.0
    if torch.cuda.device_count() >= 2:
        with torch.cuda.device(1):
            Y2 = eng.demix(X, Wd)  # guarded: runs on cuda:0 although cuda:1 is current
            assert torch.equal(Y2, Y)
            rc = _lib.lib.assx_demix(eng.ctx, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Wd.data_ptr()), None,
                                     ctypes.c_void_p(Y.data_ptr()), 1, 2, 5, 70, _lib.F64, None)
            assert rc == -1 and b"current device" in _lib.lib.assx_last_error(eng.ctx)


def test_frequency_sharded_mode_full_size_matches_the_class():
    """bss/ilrma_fshard.py at config-4 size: 4 bin shards on one process against the unsharded GaussILRMA class from the
    same initial model, two iterations + the final projection back.  They differ by the rounding of the reduction over
    f only (shard partials added in shard order): W, basis, activation, output to 1e-8, the recorded loss to 1e-10."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.bss.ilrma_fshard import FrequencyShardedGaussILRMA
    M, F, T, K = 4, 1025, 4096, 4
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(77)
    S = torch.view_as_complex(torch.randn((M, F, T, 2), dtype=torch.float64, device=dev, generator=g))
    env = 0.1 + torch.rand((M, 1, T), dtype=torch.float64, device=dev, generator=g) ** 2
    A = torch.view_as_complex(torch.randn((F, M, M, 2), dtype=torch.float64, device=dev, generator=g))
    X = torch.einsum("fmn,nft->mft", A, S * env).contiguous()
    st = np.random.RandomState(78)
    T0, V0 = st.rand(M, F, K), st.rand(M, K, T)
    m = FrequencyShardedGaussILRMA(n_basis=K, n_shards=4)
    Y = m(X, iteration=2, basis=T0, activation=V0)
    ref = GaussILRMA(n_basis=K)
    ref.basis, ref.activation = T0, V0
    Yr = ref(X, iteration=2)
    Yr = Yr.cpu().numpy() if isinstance(Yr, torch.Tensor) else Yr
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(m.demix_filter, np.asarray(ref.demix_filter)) < 1e-8
    assert rel(m.basis, np.asarray(ref.basis)) < 1e-8 and rel(m.activation, np.asarray(ref.activation)) < 1e-8
    assert rel(Y, Yr) < 1e-8
    np.testing.assert_allclose(m.loss, np.asarray(ref.loss), rtol=1e-10)


@pytest.mark.parametrize("M,K", [(8, 4), (5, 4), (8, 10)])
def test_wide_channel_full_size_oracle_step(M, K):
    """The wide-channel path (5 <= M <= 8: src_cov_kernel's deep LDS ring, flat partition with the real 512 ranges, the
    demixed-power map + matrix-core source model, 64-lane IP groups) at config-4 bins / frames against ONE oracle step:
    W, basis, activation, the cond mask and the loss after it."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X, T0, V0 = _cfg4_state(40 + M, K=K, M=M)
    m = GaussILRMA(n_basis=K)
    m.basis, m.activation = T0, V0
    m.input = X
    m._reset()
    m.update_once()
    W = np.tile(np.eye(M, dtype=np.complex128), (1025, 1, 1))
    W, Tr, Vr, mask = orc.ilrma_update_once(X, W, T0, V0)
    assert mask.all()
    assert rel_err(m.basis, Tr) < 1e-10 and rel_err(m.activation, Vr) < 1e-10 and rel_err(m.demix_filter, W) < 1e-8
    np.testing.assert_allclose(m.compute_negative_loglikelihood(), orc.ilrma_loss(X, W, Tr, Vr), rtol=1e-10)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_wide_channel_covariance_forms_agree_at_full_size(dtype):
    """pair_cov_kernel against src_cov_kernel (ASSX_WIDEM_PAIRS=0; the switch is read once per process) at config-4 bins /
    frames with the real 512 ranges, M = 5..8, every weight form (rebuilt from (Tb, V), (N,T), (N,F,T)), each twice: the
    occupancies of the benchmark size (several workgroups per CU in float32 and at M = 5) are not reached by the small
    cases of test_gpu_widem.py, and a timing-dependent error shows as garbage in SOME bins of SOME runs (round 3: registers
    of an inline-asm LDS read copied before their wait; tools/probes/paircov_check.py is the body)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "probes", "paircov_check.py")
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        from conftest import lab_build
        lab = lab_build()  # src_cov_kernel (round 3's form) exists in laboratory builds only: the shipped library runs
        for mode in ("1", "0"):  # pair_cov_kernel in two processes, which must then agree bit for bit
            paths[mode] = os.path.join(d, "u%s.npz" % mode)
            subprocess.run([sys.executable, script, "run", paths[mode], dtype], check=True, timeout=900,
                           env=dict(os.environ, ASSX_WIDEM_PAIRS=mode), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        a, b = np.load(paths["1"]), np.load(paths["0"])
        tol = (1e-10 if dtype == "float64" else 1e-3) if lab else 0.0
        for k in a.files:
            dd = np.abs(a[k] - b[k]).max(axis=(-1, -2)) / np.abs(b[k]).max(axis=(-1, -2))
            assert dd.max() <= tol, (k, float(dd.max()), int((dd > tol).sum()))
