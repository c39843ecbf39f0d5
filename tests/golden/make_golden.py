#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *reference* itself.

Runs ONLY in the build container (it imports /root/reference/src, which never
travels to the GPU box).  Everything it writes is data: seeded inputs, the
initial state the reference drew from the global NumPy RNG, and the reference's
outputs after k iterations.  No reference source is copied.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
    python tests/golden/make_golden.py f4 iss     # only the named groups
    python tests/golden/make_golden.py --verify   # regenerate into a temporary directory and compare every array of
                                                  # every committed fixture bit for bit (exit code 1 on any difference)

NumPy >= 2 note (caveat 1): the reference's IP update
calls ``np.linalg.solve(WU, e_n)`` with a stack of vectors as ``b``
(src/bss/ilrma.py:523, src/bss/iva.py:511,744).  NumPy 2 interprets that as a
matrix and raises.  We wrap -- not edit -- the call with NumPy-1.x semantics
before importing the reference.
"""
import os
import sys
import warnings

import numpy as np

REFERENCE_SRC = os.environ.get("ASSX_REFERENCE_SRC", "/root/reference/src")
OUT_DIR = os.path.dirname(os.path.abspath(__file__))

_orig_solve = np.linalg.solve


def _solve_numpy1(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    if b.ndim == a.ndim - 1:
        return _orig_solve(a, b[..., None])[..., 0]
    return _orig_solve(a, b)


np.linalg.solve = _solve_numpy1
sys.path.insert(0, REFERENCE_SRC)

from algorithm.nmf import EUCNMF, KLNMF, ISNMF  # noqa: E402
from algorithm.projection_back import projection_back  # noqa: E402
from bss.iva import AuxLaplaceIVA, AuxGaussIVA  # noqa: E402
from bss.ilrma import GaussILRMA, tILRMA  # noqa: E402

warnings.simplefilter("ignore")

VERSIONS = np.array("numpy=%s" % np.__version__)
SNAP_ITERS = (1, 2, 5, 20)


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def noise_mixture(M, F, T, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))


def convolutive_mixture(M, F, T, seed, n_basis=3):
    """Low-rank-variance complex Gaussian sources mixed by a random per-bin matrix."""
    rng = np.random.default_rng(seed)
    S = np.empty((M, F, T), dtype=np.complex128)
    for n in range(M):
        Tb = rng.random((F, n_basis)) ** 2
        Vb = rng.random((n_basis, T)) ** 4  # sparse-ish activations
        var = Tb @ Vb + 1e-3
        S[n] = np.sqrt(var / 2) * (rng.standard_normal((F, T)) + 1j * rng.standard_normal((F, T)))
    A = rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M))
    X = np.einsum("fmn,nft->mft", A, S)
    return X


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, versions=VERSIONS, **arrays)
    print("wrote %-44s %8.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------
# G1: NMF
# ----------------------------------------------------------------------------
def gen_nmf():
    cases = [
        # name, cls, kwargs, (F, T, K), seed
        ("euc_d2", EUCNMF, dict(domain=2), (65, 48, 4)),
        ("euc_d1", EUCNMF, dict(domain=1), (65, 48, 4)),
        ("euc_d15", EUCNMF, dict(domain=1.5), (33, 40, 3)),
        ("kl_d2", KLNMF, dict(domain=2), (65, 48, 4)),
        ("kl_d1", KLNMF, dict(domain=1), (65, 48, 4)),
        ("kl_d15", KLNMF, dict(domain=1.5), (33, 40, 3)),
        ("is_mm_d2", ISNMF, dict(domain=2, algorithm="mm"), (65, 48, 4)),
        ("is_mm_d1", ISNMF, dict(domain=1, algorithm="mm"), (65, 48, 4)),
        ("is_mm_d15", ISNMF, dict(domain=1.5, algorithm="mm"), (33, 40, 3)),
        ("is_me_d2", ISNMF, dict(domain=2, algorithm="me"), (65, 48, 4)),
        # config 1 of BASELINE.json: EUC-NMF F=513, T=256, K=8
        ("euc_cfg1", EUCNMF, dict(domain=2), (513, 256, 8)),
        ("is_k32", ISNMF, dict(domain=2), (40, 96, 32)),
    ]
    for idx, (name, cls, kw, (F, T, K)) in enumerate(cases):
        rng = np.random.default_rng(100 + idx)
        if name == "euc_cfg1":
            X = rng.random((F, T)) ** 2
        else:
            # power spectrogram of a low-rank model + noise; contains a few exact zeros (eps floors)
            X = (rng.random((F, K)) @ rng.random((K, T))) * rng.exponential(size=(F, T))
            X[rng.random((F, T)) < 0.01] = 0.0
        out = dict(X=X, F=F, T=T, K=K, domain=kw.get("domain", 2), algorithm=kw.get("algorithm", "mm"),
                   kind=name.split("_")[0].upper(), seed=7 + idx)
        iters = SNAP_ITERS if name != "euc_cfg1" else (1, 5)
        for k in iters:
            np.random.seed(7 + idx)
            model = cls(n_basis=K, **kw)
            if k == iters[0]:
                # record the exact init the reference draws (basis first, then activation: nmf.py:42-43)
                state = np.random.get_state()
                out["T0"] = np.random.rand(F, K)
                out["V0"] = np.random.rand(K, T)
                np.random.set_state(state)
            Tk, Vk = model(X, iteration=k)
            out["T_%d" % k] = Tk
            out["V_%d" % k] = Vk
            out["loss_%d" % k] = np.asarray(model.loss, dtype=np.float64)
        out["iters"] = np.asarray(iters)
        save("nmf_" + name, **out)


# ----------------------------------------------------------------------------
# G2: AuxIVA (IP)
# ----------------------------------------------------------------------------
class Snapshot:
    """Callback recording the public state the reference exposes after each iteration."""

    def __init__(self, iters, with_nmf):
        self.iters = set(iters)
        self.with_nmf = with_nmf
        self.count = -1
        self.data = {}

    def __call__(self, model):
        self.count += 1  # 0 = before the loop
        k = self.count
        if k in self.iters:
            self.data["W_%d" % k] = model.demix_filter.copy()
            if self.with_nmf:
                self.data["T_%d" % k] = model.basis.copy()
                self.data["V_%d" % k] = model.activation.copy()


def gen_auxiva():
    for cls, tag in ((AuxLaplaceIVA, "laplace"), (AuxGaussIVA, "gauss")):
        for M in (2, 3, 4):
            F, T = 33, 64
            X = convolutive_mixture(M, F, T, seed=200 + M)
            snap = Snapshot(SNAP_ITERS, with_nmf=False)
            model = cls(algorithm_spatial="IP", callbacks=snap)
            Y = model(X, iteration=max(SNAP_ITERS))
            save("auxiva_%s_m%d" % (tag, M), X=X, M=M, F=F, T=T, kind=tag, iters=np.asarray(SNAP_ITERS),
                 loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter, **snap.data)
        # no projection back, noise input, reference_id != 0
        X = noise_mixture(3, 17, 40, seed=260)
        model = cls(algorithm_spatial="IP", apply_projection_back=False)
        Y = model(X, iteration=3)
        model2 = cls(algorithm_spatial="IP1", reference_id=2)
        Y2 = model2(X, iteration=3)
        save("auxiva_%s_opts" % tag, X=X, kind=tag, Y_nopb=Y, loss_nopb=np.asarray(model.loss),
             Y_ref2=Y2, loss_ref2=np.asarray(model2.loss), W_nopb=model.demix_filter, W_ref2=model2.demix_filter)


# ----------------------------------------------------------------------------
# G3: GaussILRMA (IP)
# ----------------------------------------------------------------------------
def gen_ilrma():
    seed = 300
    for M, K, normalize, domain in [
        (2, 2, "power", 2), (4, 4, "power", 2), (3, 5, "power", 2),
        (2, 4, "projection-back", 2), (4, 2, "projection-back", 2),
        (2, 2, False, 2), (4, 4, False, 2),
        (2, 4, "power", 1), (4, 2, "projection-back", 1), (3, 3, "power", 1.5),
    ]:
        seed += 1
        F, T = 33, 64
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0 = np.random.rand(M, F, K)
        V0 = np.random.rand(M, K, T)
        np.random.set_state(state)
        snap = Snapshot(SNAP_ITERS, with_nmf=True)
        model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, callbacks=snap)
        Y = model(X, iteration=max(SNAP_ITERS))
        tag = "m%d_k%d_%s_d%s" % (M, K, {"power": "pow", "projection-back": "pb", False: "none"}[normalize],
                                   str(domain).replace(".", ""))
        save("ilrma_" + tag, X=X, M=M, F=F, T=T, K=K, domain=domain,
             normalize=np.array(str(normalize)), seed=seed, T0=T0, V0=V0, iters=np.asarray(SNAP_ITERS),
             loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
             T_final=model.basis, V_final=model.activation, **snap.data)

    # stage-level intermediates: one source-model step, then one spatial step, from a non-trivial state
    M, K, F, T = 4, 4, 17, 48
    X = convolutive_mixture(M, F, T, seed=390)
    np.random.seed(390)
    model = GaussILRMA(n_basis=K, recordable_loss=True)
    model(X, iteration=3)  # warm state
    W0, T0, V0 = model.demix_filter.copy(), model.basis.copy(), model.activation.copy()
    stage = GaussILRMA(n_basis=K)
    stage.input = X
    stage._reset(demix_filter=W0.copy(), basis=T0.copy(), activation=V0.copy())
    loss0 = stage.compute_negative_loglikelihood()
    stage.update_source_model()
    T1, V1 = stage.basis.copy(), stage.activation.copy()
    # weighted covariance as the reference forms it (ilrma.py:497-511), small enough to materialise
    R = (T1 @ V1) ** (2 / 2)
    R[R < stage.eps] = stage.eps
    Xt = X.transpose(1, 2, 0)[..., None]
    XX = Xt @ Xt.transpose(0, 1, 3, 2).conj()
    U = (XX / R[..., None, None]).mean(axis=2)
    stage.update_spatial_model()
    W1 = stage.demix_filter.copy()
    Y1 = stage.estimation.copy()
    loss1 = stage.compute_negative_loglikelihood()
    save("ilrma_stages", X=X, W0=W0, T0=T0, V0=V0, T1=T1, V1=V1, U=U, W1=W1, Y1=Y1,
         loss0=loss0, loss1=loss1)

    # warm start across two calls + loss list continuation (ilrma.py:67-72, 44-48)
    X = convolutive_mixture(2, 17, 40, seed=395)
    np.random.seed(395)
    model = GaussILRMA(n_basis=3)
    Ya = model(X, iteration=2)
    Yb = model(X, iteration=3)
    save("ilrma_warm", X=X, seed=395, K=3, Y_a=Ya, Y_b=Yb, loss=np.asarray(model.loss),
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation)


# ----------------------------------------------------------------------------
# G4: projection_back; G5: edge cases
# ----------------------------------------------------------------------------
def gen_projection_back():
    rng = np.random.default_rng(400)
    out = {}
    for N in (2, 3, 4):
        F, T = 9, 50
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        ref = rng.standard_normal((F, T)) + 1j * rng.standard_normal((F, T))
        refs = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        out["Y_n%d" % N] = Y
        out["ref_n%d" % N] = ref
        out["scale_n%d" % N] = projection_back(Y, ref)
        out["refs_n%d" % N] = refs
        out["scale3_n%d" % N] = projection_back(Y, refs)
    save("projection_back", **out)


def gen_edge():
    # (a) bins whose WU is numerically singular: cond(WU) >= 1e12 -> the old row must be kept
    #     (ilrma.py:520-528).  Channel 1 = channel 0 * c + 1e-14 noise in bins 2 and 5.
    M, F, T, K = 3, 8, 40, 2
    X = convolutive_mixture(M, F, T, seed=500)
    rng = np.random.default_rng(501)
    for f in (2, 5):
        X[1, f] = X[0, f] * (0.7 - 0.2j) + 1e-14 * rng.standard_normal(T)
    np.random.seed(500)
    T0 = np.random.rand(M, F, K)
    V0 = np.random.rand(M, K, T)
    np.random.seed(500)
    snap = Snapshot((1, 2), with_nmf=True)
    model = GaussILRMA(n_basis=K, callbacks=snap)
    Y = model(X, iteration=2)
    # recompute the cond mask of the first IP sweep exactly as the reference does
    save("edge_cond_ilrma", X=X, K=K, seed=500, T0=T0, V0=V0, loss=np.asarray(model.loss),
         W_final=model.demix_filter, Y_out=Y, **snap.data)

    snap = Snapshot((1, 2), with_nmf=False)
    model = AuxLaplaceIVA(callbacks=snap)
    Y = model(X, iteration=2)
    save("edge_cond_auxiva", X=X, loss=np.asarray(model.loss), W_final=model.demix_filter, Y_out=Y, **snap.data)

    # (b) exact zeros in X: whole frames silent -> eps floors on R / TV are hit (ilrma.py:415,509; iva.py:497)
    X = convolutive_mixture(2, 9, 48, seed=510)
    X[:, :, 10:14] = 0.0
    X[:, 3, :] *= 1e-9
    np.random.seed(510)
    T0 = np.random.rand(2, 9, 2)
    V0 = np.random.rand(2, 2, 48)
    np.random.seed(510)
    model = GaussILRMA(n_basis=2)
    Y = model(X, iteration=3)
    save("edge_zeros_ilrma", X=X, K=2, seed=510, T0=T0, V0=V0, loss=np.asarray(model.loss),
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation, Y_out=Y)
    model = AuxLaplaceIVA()
    Y = model(X, iteration=3)
    save("edge_zeros_auxlaplace", X=X, loss=np.asarray(model.loss), W_final=model.demix_filter, Y_out=Y)
    model = AuxGaussIVA()
    Y = model(X, iteration=3)
    save("edge_zeros_auxgauss", X=X, loss=np.asarray(model.loss), W_final=model.demix_filter, Y_out=Y)


# ----------------------------------------------------------------------------
# G6: ISS spatial updates
# ----------------------------------------------------------------------------
class SnapshotISS(Snapshot):
    """During ISS the reference exposes demix_filter only inside callbacks (ilrma.py:219-228)."""


def gen_iss():
    for cls, tag in ((AuxLaplaceIVA, "laplace"), (AuxGaussIVA, "gauss")):
        for M in (2, 3, 4):
            F, T = 17, 48
            X = convolutive_mixture(M, F, T, seed=600 + M)
            snap = Snapshot((1, 2, 5), with_nmf=False)
            model = cls(algorithm_spatial="ISS", callbacks=snap)
            Y = model(X, iteration=5)
            save("iss_auxiva_%s_m%d" % (tag, M), X=X, kind=tag, iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss),
                 Y_out=Y, W_final=model.demix_filter, **snap.data)
    seed = 650
    for M, K, normalize, domain in [(2, 2, "power", 2), (4, 4, "power", 2), (3, 3, "projection-back", 2),
                                    (4, 2, False, 2), (3, 4, "power", 1)]:
        seed += 1
        F, T = 17, 48
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0 = np.random.rand(M, F, K)
        V0 = np.random.rand(M, K, T)
        np.random.set_state(state)
        snap = Snapshot((1, 2, 5), with_nmf=True)
        model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, algorithm_spatial="ISS", callbacks=snap)
        Y = model(X, iteration=5)
        tag = "m%d_k%d_%s_d%s" % (M, K, {"power": "pow", "projection-back": "pb", False: "none"}[normalize],
                                   str(domain).replace(".", ""))
        save("iss_ilrma_" + tag, X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)), seed=seed, T0=T0,
             V0=V0, iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
             T_final=model.basis, V_final=model.activation, **snap.data)


# ----------------------------------------------------------------------------
# G6b: IP2 / pairwise spatial updates
# ----------------------------------------------------------------------------
def gen_ip2():
    for M in (2, 3, 4):
        F, T = 17, 48
        X = convolutive_mixture(M, F, T, seed=700 + M)
        snap = Snapshot((1, 2, 6), with_nmf=False)
        model = AuxLaplaceIVA(algorithm_spatial="IP2", callbacks=snap)
        Y = model(X, iteration=6)
        save("ip2_auxlaplace_m%d" % M, X=X, iters=np.asarray((1, 2, 6)), loss=np.asarray(model.loss), Y_out=Y,
             W_final=model.demix_filter, update_pair=np.asarray(model.update_pair), **snap.data)
    seed = 750
    for M, K, normalize, domain, alg in [(2, 2, "power", 2, "IP2"), (4, 4, "power", 2, "pairwise"),
                                         (3, 3, "projection-back", 2, "IP2"), (4, 2, False, 1, "IP2")]:
        seed += 1
        F, T = 17, 48
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0 = np.random.rand(M, F, K)
        V0 = np.random.rand(M, K, T)
        np.random.set_state(state)
        snap = Snapshot((1, 2, 6), with_nmf=True)
        model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, algorithm_spatial=alg, callbacks=snap)
        Y = model(X, iteration=6)
        tag = "m%d_k%d_%s_d%s" % (M, K, {"power": "pow", "projection-back": "pb", False: "none"}[normalize],
                                   str(domain).replace(".", ""))
        save("ip2_ilrma_" + tag, X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)), seed=seed, T0=T0,
             V0=V0, alg=np.array(alg), iters=np.asarray((1, 2, 6)), loss=np.asarray(model.loss), Y_out=Y,
             W_final=model.demix_filter, T_final=model.basis, V_final=model.activation,
             update_pair=np.asarray(model.update_pair), **snap.data)


# ----------------------------------------------------------------------------
# G6c: partitioning function (shared bases + latent Z)   (ilrma.py:79-95, 368-408, 313-320)
# ----------------------------------------------------------------------------
def gen_part(cases=((2, 3, "power", "IP"), (3, 4, "power", "IP"), (4, 4, False, "IP"), (3, 3, "power", "ISS")), seed=800,
             shape=(17, 48)):
    for M, K, normalize, alg in cases:
        seed += 1
        F, T = shape
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        Z0 = np.random.rand(M, K) * 1e-2 + 1 / M
        Z0 = Z0 / Z0.sum(axis=0)
        T0 = np.random.rand(F, K)
        V0 = np.random.rand(K, T)
        np.random.set_state(state)
        snap_iters = (1, 2, 5)

        class SnapZ(Snapshot):
            def __call__(self, model):
                super().__call__(model)
                if self.count in self.iters:
                    self.data["Z_%d" % self.count] = model.latent.copy()

        snap = SnapZ(snap_iters, with_nmf=True)
        model = GaussILRMA(n_basis=K, partitioning=True, normalize=normalize, algorithm_spatial=alg, callbacks=snap)
        Y = model(X, iteration=5)
        tag = "m%d_k%d_%s_%s" % (M, K, {"power": "pow", False: "none"}[normalize], alg.lower())
        save("part_ilrma_" + tag, X=X, M=M, K=K, normalize=np.array(str(normalize)), alg=np.array(alg), seed=seed,
             Z0=Z0, T0=T0, V0=V0, iters=np.asarray(snap_iters), loss=np.asarray(model.loss), Y_out=Y,
             W_final=model.demix_filter, Z_final=model.latent, T_final=model.basis, V_final=model.activation,
             **snap.data)


# ----------------------------------------------------------------------------
# G6d: t-ILRMA (ilrma.py:713-1020)
# ----------------------------------------------------------------------------
def gen_tilrma(cases=((2, 2, 1, "power"), (3, 4, 5, "power"), (4, 4, 100, "power"), (4, 6, 2.5, False)), seed=900,
               shape=(17, 72)):
    for M, K, nu, normalize in cases:
        seed += 1
        F, T = shape
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0 = np.random.rand(M, F, K)
        V0 = np.random.rand(M, K, T)
        np.random.set_state(state)
        snap_iters = (1, 2, 5)
        snap = Snapshot(snap_iters, with_nmf=True)
        model = tILRMA(n_basis=K, nu=nu, normalize=normalize, callbacks=snap)
        Y = model(X, iteration=5)
        tag = "m%d_k%d_nu%s_%s" % (M, K, str(nu).replace(".", "p"), {"power": "pow", False: "none"}[normalize])
        save("tilrma_" + tag, X=X, M=M, K=K, nu=float(nu), normalize=np.array(str(normalize)), seed=seed, T0=T0, V0=V0,
             iters=np.asarray(snap_iters), loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
             T_final=model.basis, V_final=model.activation, **snap.data)


def gen_k10():
    """The reference's DEFAULT n_basis = 10 (ilrma.py:183) through every spatial algorithm: the n_basis > 4 kernels."""
    def init(M, F, T, K, seed):
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0 = np.random.rand(M, F, K)
        V0 = np.random.rand(M, K, T)
        np.random.set_state(state)
        return X, T0, V0

    def tag_of(M, K, normalize, domain):
        return "m%d_k%d_%s_d%s" % (M, K, {"power": "pow", "projection-back": "pb", False: "none"}[normalize],
                                   str(domain).replace(".", ""))
    K = 10
    for seed, (M, normalize, domain) in zip((1201, 1202), [(3, "power", 2), (4, "projection-back", 1)]):
        F, T = 33, 64
        X, T0, V0 = init(M, F, T, K, seed)
        snap = Snapshot(SNAP_ITERS, with_nmf=True)
        model = GaussILRMA(domain=domain, normalize=normalize, callbacks=snap)   # n_basis left at its default
        assert model.n_basis == K
        Y = model(X, iteration=max(SNAP_ITERS))
        save("ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, F=F, T=T, K=K, domain=domain,
             normalize=np.array(str(normalize)), seed=seed, T0=T0, V0=V0, iters=np.asarray(SNAP_ITERS),
             loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
             T_final=model.basis, V_final=model.activation, **snap.data)
    M, normalize, domain, seed = 3, "power", 2, 1203
    X, T0, V0 = init(M, 17, 48, K, seed)
    snap = Snapshot((1, 2, 5), with_nmf=True)
    model = GaussILRMA(domain=domain, normalize=normalize, algorithm_spatial="ISS", callbacks=snap)
    Y = model(X, iteration=5)
    save("iss_ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)),
         seed=seed, T0=T0, V0=V0, iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss), Y_out=Y,
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation, **snap.data)
    M, normalize, domain, seed, alg = 4, "power", 2, 1204, "IP2"
    X, T0, V0 = init(M, 17, 48, K, seed)
    snap = Snapshot((1, 2, 6), with_nmf=True)
    model = GaussILRMA(domain=domain, normalize=normalize, algorithm_spatial=alg, callbacks=snap)
    Y = model(X, iteration=6)
    save("ip2_ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)),
         seed=seed, T0=T0, V0=V0, alg=np.array(alg), iters=np.asarray((1, 2, 6)), loss=np.asarray(model.loss), Y_out=Y,
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation,
         update_pair=np.asarray(model.update_pair), **snap.data)


def gen_tilrma_k10():
    """tILRMA at its default n_basis = 10 (ilrma.py:713)."""
    seed, M, K, nu, normalize, F, T = 1301, 3, 10, 5, "power", 17, 72
    X = convolutive_mixture(M, F, T, seed=seed)
    np.random.seed(seed)
    state = np.random.get_state()
    T0 = np.random.rand(M, F, K)
    V0 = np.random.rand(M, K, T)
    np.random.set_state(state)
    snap_iters = (1, 2, 5)
    snap = Snapshot(snap_iters, with_nmf=True)
    model = tILRMA(nu=nu, normalize=normalize, callbacks=snap)
    assert model.n_basis == K
    Y = model(X, iteration=5)
    save("tilrma_m3_k10_nu5_pow", X=X, M=M, K=K, nu=float(nu), normalize=np.array(str(normalize)), seed=seed, T0=T0, V0=V0,
         iters=np.asarray(snap_iters), loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
         T_final=model.basis, V_final=model.activation, **snap.data)


def gen_consistent():
    """ConsistentGaussILRMA (ilrma.py:1089-1233), IP only."""
    from bss.ilrma import ConsistentGaussILRMA
    seed, M, K, fft_size = 1401, 3, 4, 32
    F, T = fft_size // 2 + 1, 40
    X = convolutive_mixture(M, F, T, seed=seed)
    np.random.seed(seed)
    state = np.random.get_state()
    T0 = np.random.rand(M, F, K)
    V0 = np.random.rand(M, K, T)
    np.random.set_state(state)
    snap = Snapshot((1, 2, 5), with_nmf=True)
    model = ConsistentGaussILRMA(n_basis=K, fft_size=fft_size, hop_size=fft_size // 2, callbacks=snap)
    Y = model(X, iteration=5)
    save("consistent_ilrma_m3_k4", X=X, M=M, K=K, fft_size=fft_size, seed=seed, T0=T0, V0=V0,
         iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter,
         T_final=model.basis, V_final=model.activation, repr=np.array(repr(model)), **snap.data)


def gen_part_k10():
    """partitioning=True at the default n_basis = 10 (IP and ISS)."""
    gen_part(cases=((3, 10, "power", "IP"), (4, 10, "power", "ISS")), seed=1500)


def gen_xnmf():
    """tNMF / CauchyNMF (the other users of the NMF skeleton), nmf.py:358-600."""
    from algorithm.nmf import tNMF, CauchyNMF
    cases = [("t_nu1", tNMF, dict(nu=1.0), (33, 40, 3)), ("t_nu1000", tNMF, dict(), (65, 48, 4)),
             ("t_k20", tNMF, dict(nu=4.0), (40, 96, 20)),
             ("cauchy_naive", CauchyNMF, dict(algorithm="naive-multipricative"), (33, 40, 3)),
             ("cauchy_mm", CauchyNMF, dict(algorithm="mm"), (65, 48, 4)),
             ("cauchy_me", CauchyNMF, dict(algorithm="me"), (33, 40, 3)),
             ("cauchy_mm_fast", CauchyNMF, dict(algorithm="mm_fast"), (65, 48, 4)),
             ("cauchy_mm_k20", CauchyNMF, dict(algorithm="mm"), (40, 96, 20))]
    for idx, (name, cls, kw, (F, T, K)) in enumerate(cases):
        rng = np.random.default_rng(1100 + idx)
        X = (rng.random((F, K)) @ rng.random((K, T))) * rng.exponential(size=(F, T))
        X[rng.random((F, T)) < 0.01] = 0.0
        out = dict(X=X, F=F, T=T, K=K, nu=float(kw.get("nu", 1e3)), algorithm=kw.get("algorithm", "mm"),
                   kind=name.split("_")[0], seed=70 + idx)
        iters = (1, 2, 5, 20)
        for k in iters:
            np.random.seed(70 + idx)
            model = cls(n_basis=K, **kw)
            if k == iters[0]:
                state = np.random.get_state()
                out["T0"] = np.random.rand(F, K)
                out["V0"] = np.random.rand(K, T)
                np.random.set_state(state)
            Tk, Vk = model(X, iteration=k)
            out["T_%d" % k] = Tk
            out["V_%d" % k] = Vk
            out["loss_%d" % k] = np.asarray(model.loss, dtype=np.float64)
        out["iters"] = np.asarray(iters)
        save("xnmf_" + name, **out)


def stft_perturb(X):
    """Deterministic modulation that takes a spectrogram off the set of consistent ones (also sets the imaginary
    parts of the DC / Nyquist bins, which irfft must ignore)."""
    return X * (1.0 + 0.1 * np.cos(np.arange(X.size, dtype=np.float64)).reshape(X.shape)) + 0.01j


def gen_stft():
    """stft / istft of src/transform/stft.py (scipy.signal.stft / istft) on seeded noise: power-of-two and
    odd / non-power-of-two fft sizes, hops that do and do not divide the length, both windows."""
    import warnings
    from transform.stft import stft, istft
    rng = np.random.default_rng(1000)
    out = {}
    cases = [(1000, 64, 16, "hann"), (777, 128, 64, "hann"), (500, 32, 8, "hamming"), (300, 30, 10, "hann"),
             (401, 33, 11, "hann"), (4096, 256, 128, "hann"), (100, 64, 48, "hann"), (5000, 1024, 256, "hann")]
    for i, (L, N, hop, wf) in enumerate(cases):
        x = rng.standard_normal((2 if L < 2000 else 1, L))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")   # NOLA warnings for hops that break the overlap-add constraint
            X = stft(x, fft_size=N, hop_size=hop, window_fn=wf)
            Z = stft_perturb(X)   # not a consistent spectrogram (rule shared with the tests, not stored)
            y = istft(Z, fft_size=N, hop_size=hop, window_fn=wf)
            y_cut = istft(X, fft_size=N, hop_size=hop, window_fn=wf, length=L)
        out.update({"x%d" % i: x, "X%d" % i: X, "y%d" % i: y, "ycut%d" % i: y_cut})
    save("stft", cases=np.array([(L, N, hop, wf == "hamming") for L, N, hop, wf in cases]), **out)


def gen_f4(idlma_cases=((2, 2), (3, 1), (4, 2), (4, 1.5)),
           mnmf_cases=((2, 2, 3, False), (3, 2, 4, False), (4, 3, 3, False), (4, 5, 2, True)), shape=(17, 96)):
    """Other callers of the covariance + IP kernels: the reference's own methods on seeded state."""
    from sss.idlma import GaussIDLMA
    from bss.mnmf import FastMultichannelISNMF
    F, T = shape
    for M, domain in idlma_cases:
        rng = np.random.default_rng(900 + M)
        X = convolutive_mixture(M, F, T, 901 + M)
        W0 = np.eye(M)[None] + 0.3 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))
        dnn = rng.random((M, F, T)) ** 2 + 1e-3
        dnn[0, 3, :max(5, M + 2)] = 0.0  # hits the eps floor (idlma.py:189); >= M frames keep the bin well conditioned
        m = object.__new__(GaussIDLMA)
        m.input, m.demix_filter, m.dnn_output = X, W0.copy(), dnn.copy()
        m.domain, m.eps, m.threshold = domain, 1e-12, 1e12
        m.n_sources = m.n_channels = M
        m.n_bins, m.n_frames = F, T
        m.update_space_model()
        save("f4_idlma_m%d_d%s" % (M, str(domain).replace(".", "")), X=X, W0=W0, dnn_output=dnn, domain=float(domain),
             W1=m.demix_filter)
    for M, N, K, part in mnmf_cases:
        rng = np.random.default_rng(950 + M + N)
        X = convolutive_mixture(M, F, T, 951 + M)
        Q0 = np.eye(M)[None] + 0.3 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))
        g = rng.random((N, F, M)) + 1e-2
        m = object.__new__(FastMultichannelISNMF)
        m.input, m.diagonalizer, m.spatial_covariance = X, Q0.copy(), g.copy()
        m.partitioning = part
        if part:
            Z = rng.random((N, K))
            m.latent = Z / Z.sum(axis=0)
            m.basis, m.activation = rng.random((F, K)), rng.random((K, T))
            Lam = (m.latent[:, None, :] * m.basis[None]) @ m.activation[None]
        else:
            m.basis, m.activation = rng.random((N, F, K)), rng.random((N, K, T))
            Lam = m.basis @ m.activation
        m.eps, m.threshold = 1e-12, 1e12
        m.n_bins, m.n_channels, m.n_sources = F, M, N
        m.update_diagonalizer()
        extra = dict(latent=m.latent) if part else {}
        save("f4_fastmnmf_m%d_n%d%s" % (M, N, "_part" if part else ""), X=X, Q0=Q0, g=g, basis=m.basis,
             activation=m.activation, variance=Lam, partitioning=part, Q1=m.diagonalizer, **extra)


def gen_wide_m():
    """5 <= M <= 8 channels (the reference is generic in M, ilrma.py:61-62): same file formats as the M <= 4 groups, so
    the same oracle / GPU tests pick them up."""
    def tag_of(M, K, normalize, domain):
        return "m%d_k%d_%s_d%s" % (M, K, {"power": "pow", "projection-back": "pb", False: "none"}[normalize],
                                   str(domain).replace(".", ""))
    F, T = 13, 256  # enough frames for well-conditioned 8 x 8 covariances: rounding is not amplified beyond the
    WIDE_ITERS = (1, 2, 5, 10)  # M <= 4 fixtures' tolerances
    seed = 1300
    for M, K, normalize, domain in [(5, 3, "power", 2), (6, 10, "projection-back", 1), (8, 4, "power", 2),
                                    (9, 3, "power", 2)]:  # M = 9: beyond the compile-time channel counts (round 3)
        seed += 1
        X = convolutive_mixture(M, F, T, seed=seed)
        np.random.seed(seed)
        state = np.random.get_state()
        T0, V0 = np.random.rand(M, F, K), np.random.rand(M, K, T)
        np.random.set_state(state)
        snap = Snapshot(WIDE_ITERS, with_nmf=True)
        model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, callbacks=snap)
        Y = model(X, iteration=max(WIDE_ITERS))
        save("ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, F=F, T=T, K=K, domain=domain,
             normalize=np.array(str(normalize)), seed=seed, T0=T0, V0=V0, iters=np.asarray(WIDE_ITERS),
             loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter, T_final=model.basis,
             V_final=model.activation, **snap.data)
    for cls, tag, M in ((AuxLaplaceIVA, "laplace", 5), (AuxGaussIVA, "gauss", 6)):
        X = convolutive_mixture(M, F, T, seed=1320 + M)
        snap = Snapshot(WIDE_ITERS, with_nmf=False)
        model = cls(algorithm_spatial="IP", callbacks=snap)
        Y = model(X, iteration=max(WIDE_ITERS))
        save("auxiva_%s_m%d" % (tag, M), X=X, M=M, F=F, T=T, kind=tag, iters=np.asarray(WIDE_ITERS),
             loss=np.asarray(model.loss), Y_out=Y, W_final=model.demix_filter, **snap.data)
    # ISS
    M = 5
    X = convolutive_mixture(M, F, T, seed=1340)
    snap = Snapshot((1, 2, 5), with_nmf=False)
    model = AuxLaplaceIVA(algorithm_spatial="ISS", callbacks=snap)
    Y = model(X, iteration=5)
    save("iss_auxiva_laplace_m%d" % M, X=X, kind="laplace", iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss),
         Y_out=Y, W_final=model.demix_filter, **snap.data)
    M, K, normalize, domain, seed = 5, 2, "power", 2, 1341
    X = convolutive_mixture(M, F, T, seed=seed)
    np.random.seed(seed)
    state = np.random.get_state()
    T0, V0 = np.random.rand(M, F, K), np.random.rand(M, K, T)
    np.random.set_state(state)
    snap = Snapshot((1, 2, 5), with_nmf=True)
    model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, algorithm_spatial="ISS", callbacks=snap)
    Y = model(X, iteration=5)
    save("iss_ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)),
         seed=seed, T0=T0, V0=V0, iters=np.asarray((1, 2, 5)), loss=np.asarray(model.loss), Y_out=Y,
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation, **snap.data)
    # IP2
    M = 6
    X = convolutive_mixture(M, F, T, seed=1350)
    snap = Snapshot((1, 2, 6), with_nmf=False)
    model = AuxLaplaceIVA(algorithm_spatial="IP2", callbacks=snap)
    Y = model(X, iteration=6)
    save("ip2_auxlaplace_m%d" % M, X=X, iters=np.asarray((1, 2, 6)), loss=np.asarray(model.loss), Y_out=Y,
         W_final=model.demix_filter, update_pair=np.asarray(model.update_pair), **snap.data)
    M, K, normalize, domain, alg, seed = 5, 3, "power", 2, "IP2", 1351
    X = convolutive_mixture(M, F, T, seed=seed)
    np.random.seed(seed)
    state = np.random.get_state()
    T0, V0 = np.random.rand(M, F, K), np.random.rand(M, K, T)
    np.random.set_state(state)
    snap = Snapshot((1, 2, 6), with_nmf=True)
    model = GaussILRMA(n_basis=K, domain=domain, normalize=normalize, algorithm_spatial=alg, callbacks=snap)
    Y = model(X, iteration=6)
    save("ip2_ilrma_" + tag_of(M, K, normalize, domain), X=X, M=M, K=K, domain=domain, normalize=np.array(str(normalize)),
         seed=seed, T0=T0, V0=V0, alg=np.array(alg), iters=np.asarray((1, 2, 6)), loss=np.asarray(model.loss), Y_out=Y,
         W_final=model.demix_filter, T_final=model.basis, V_final=model.activation,
         update_pair=np.asarray(model.update_pair), **snap.data)


def gen_wide_variants():
    """partitioning=True and tILRMA beyond 4 channels (the wide-channel path; same file formats as the M <= 4 groups)."""
    gen_part(cases=((5, 3, "power", "IP"), (6, 10, "power", "ISS")), seed=1600, shape=(13, 256))
    gen_tilrma(cases=((5, 3, 5, "power"), (6, 10, 1, "power")), seed=1700, shape=(13, 256))
    gen_f4(idlma_cases=((5, 1.5), (6, 2)), mnmf_cases=((5, 3, 3, False), (6, 4, 2, True)), shape=(13, 256))


ALL_GROUPS = ("iss", "nmf", "auxiva", "ilrma", "projection_back", "edge", "ip2", "part", "tilrma", "stft", "xnmf", "k10",
              "tilrma_k10", "consistent", "part_k10", "f4", "wide_m", "wide_variants")


def verify(groups):
    """Regenerate `groups` into a temporary directory and compare with the committed files: the recipe, not only the
    data, is what pins the oracle."""
    import tempfile
    global OUT_DIR
    committed = OUT_DIR
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT_DIR = tmp
        for name in groups:
            globals()["gen_" + name]()
        OUT_DIR = committed
        fresh = sorted(f for f in os.listdir(tmp) if f.endswith(".npz"))
        for f in fresh:
            path = os.path.join(committed, f)
            if not os.path.exists(path):
                bad.append("%s: not committed" % f)
                continue
            a, b = np.load(os.path.join(tmp, f), allow_pickle=False), np.load(path, allow_pickle=False)
            if sorted(a.files) != sorted(b.files):
                bad.append("%s: keys differ %s" % (f, sorted(set(a.files) ^ set(b.files))))
                continue
            for k in a.files:
                if k == "versions":
                    continue
                if a[k].dtype != b[k].dtype or a[k].shape != b[k].shape or a[k].tobytes() != b[k].tobytes():
                    bad.append("%s[%s] differs" % (f, k))
        if groups == ALL_GROUPS:
            for f in sorted(set(x for x in os.listdir(committed) if x.endswith(".npz")) - set(fresh)):
                bad.append("%s: committed but no generator writes it" % f)
    print("verified %d files, %d problems" % (len(fresh), len(bad)))
    for line in bad:
        print("  MISMATCH", line)
    return 1 if bad else 0


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--verify":
        sys.exit(verify(tuple(args[1:]) or ALL_GROUPS))
    for name in (args or ALL_GROUPS):  # regenerate only the named groups, e.g. `make_golden.py iss`
        globals()["gen_" + name]()
