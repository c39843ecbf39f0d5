"""CPU-side checks: the C-ABI library loads and exports every symbol include/assx.h declares, the ctypes
table covers the header, and the host-side class logic that needs no GPU behaves like the reference."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "assx.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(assx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from audio_source_separation_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libassx.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and include/assx.h disagree"
    assert _lib.version().startswith("assx ")


def test_workspace_queries_need_no_gpu():
    from audio_source_separation_amd import _lib
    n64 = _lib.lib.assx_workspace_bytes(1, 4, 1025, 4096, 4, _lib.F64)
    n32 = _lib.lib.assx_workspace_bytes(1, 4, 1025, 4096, 4, _lib.F32)
    assert 0 < n32 < n64 < 1 << 30
    assert _lib.lib.assx_workspace_bytes(0, 4, 10, 10, 4, _lib.F64) == 0
    assert _lib.lib.assx_nmf_workspace_bytes(1, 1025, 4096, 32, _lib.F64) > 2 * 1025 * 4096 * 8


def _partition_query(feed, half, group, F, T, K, dtype):
    from audio_source_separation_amd import _lib
    out = (ctypes.c_int32 * 6)()
    rc = _lib.lib.assx_nmf_partition_query(feed, half, group, F, T, K, dtype, out)
    assert rc == 0, (rc, feed, half, group, F, T, K)
    return list(out)


def test_nmf_slab_area_holds_every_partition():
    """Round 4's advisor finding: the X-fed source-model halves cut their ranges with a larger workgroup budget than the
    map-fed ones, so more workgroups meet one block and write more slabs -- F = 1025, T = 660, n_basis = 10: 12 -- than
    the workspace had room for (11).  For every shape: workgroups that meet a block <= the launchers' bound <= slabs the
    workspace reserves; on both feeds, both halves, both precisions."""
    from audio_source_separation_amd import _lib
    g, nblk, nstep, bound, worst, room = _partition_query(1, 0, 1, 1025, 660, 10, _lib.F64)
    # the advisor's example: 12 workgroups met a block under round 4's flat partition (ASSX_NMF_ALIGNED=0), 10 do since
    # the partition is block-aligned; either way the area must hold them
    assert (nblk, nstep) == (65, 42) and worst in (10, 12) and worst <= bound <= room
    rng = np.random.default_rng(5)
    shapes = [(F, T) for F in (129, 257, 513, 1025, 2049, 4097) for T in range(16, 6000, 97)]
    shapes += [(int(F), int(T)) for F, T in zip(rng.integers(1, 5000, 300), rng.integers(1, 7000, 300))]
    shapes += [(1025, T) for T in range(640, 680)] + [(1, 1), (16, 16), (17, 15), (1025, 4096), (513, 256)]
    bad = []
    for F, T in shapes:
        for K in (5, 10, 16, 17, 32, 33, 64):
            for feed in (0, 1):
                if feed == 1 and K > 32:
                    continue
                for half in (0, 1):
                    for dtype in (_lib.F64, _lib.F32):
                        for group in ((1,) if feed else (1, 2, 4, 8)):
                            g, nblk, nstep, bound, worst, room = _partition_query(feed, half, group, F, T, K, dtype)
                            if not (1 <= worst <= bound <= room and g >= 1):
                                bad.append((feed, half, group, F, T, K, dtype, worst, bound, room))
    assert not bad, bad[:10]
    out = (ctypes.c_int32 * 6)()
    assert _lib.lib.assx_nmf_partition_query(1, 0, 1, 1025, 660, 33, _lib.F64, out) == -1  # no X-fed halves there
    assert _lib.lib.assx_nmf_partition_query(2, 0, 1, 1025, 660, 10, _lib.F64, out) == -1
    assert _lib.lib.assx_nmf_partition_query(0, 0, 1, 1025, 660, 10, _lib.F64, None) == -1


@pytest.mark.parametrize("forced_g", [0, 1, 3, 7, 8, 9, 100])
@pytest.mark.parametrize("B,F,T", [(1, 1025, 4096), (2, 1025, 4096), (8, 1025, 4096), (3, 33, 200), (5, 7, 64), (2, 1, 1)])
def test_launch_order_is_a_permutation(monkeypatch, forced_g, B, F, T):
    """Every range of a batched streaming pass is taken by exactly one workgroup, in either direction; utterances
    are walked one after the other and an XCD (workgroup index % 8) owns a contiguous run of an utterance's ranges."""
    from audio_source_separation_amd import _lib
    if forced_g:
        monkeypatch.setenv("ASSX_G", str(forced_g))  # read on every call: shrinks the partition like the GPU tests do
    else:
        monkeypatch.delenv("ASSX_G", raising=False)
    seen = {}
    legacy = os.environ.get("ASSX_UTT_ORDER", "1") == "0"  # read once per process by the library
    for rev in (0, 1):
        grid = _lib.lib.assx_launch_order(B, F, T, rev, None, 0)
        assert grid > 0
        buf = (ctypes.c_int * grid)()
        assert _lib.lib.assx_launch_order(B, F, T, rev, buf, grid) == grid
        assert _lib.lib.assx_launch_order(B, F, T, rev, buf, grid - 1) < 0
        r = np.asarray(buf[:], dtype=np.int64)
        taken = r[r >= 0]
        n = taken.size
        assert n % B == 0 and sorted(taken.tolist()) == list(range(n)), "not a permutation of the ranges"
        seen[rev] = r
        if legacy:  # ASSX_UTT_ORDER=0 in the environment of this run: the older order, a permutation is all it promises
            continue
        if B >= 2:
            gu, gp = n // B, grid // B
            assert gp % 8 == 0 and gu <= gp < gu + 8
            for b in range(B):  # launch slots b*gp .. (b+1)*gp - 1 hold exactly one utterance
                blk = r[b * gp:(b + 1) * gp]
                utt = set((blk[blk >= 0] // gu).tolist())
                assert utt == {B - 1 - b if rev else b}
                for x in range(8):  # an XCD's ranges inside the utterance: consecutive, ascending / descending
                    mine = blk[x::8]
                    mine = mine[mine >= 0]
                    if mine.size > 1:
                        assert np.all(np.diff(mine) == (-1 if rev else 1))
    if B >= 2 and not legacy:
        fwd, bwd = seen[0][seen[0] >= 0], seen[1][seen[1] >= 0]
        assert fwd[0] // (fwd.size // B) == 0 and bwd[0] // (bwd.size // B) == B - 1


def test_invalid_context_is_an_error_not_a_crash():
    from audio_source_separation_amd import _lib
    rc = _lib.lib.assx_demix(None, None, None, None, None, 1, 4, 8, 8, _lib.F64, None)
    assert rc == -3  # ASSX_E_NULL
    assert _lib.lib.assx_last_error(None) == b"ctx is NULL"


def test_no_gpu_fails_loudly():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X = np.ones((2, 4, 8), dtype=np.complex128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussILRMA(n_basis=2)(X, iteration=1)


def test_constructor_contract_matches_reference():
    from audio_source_separation_amd.algorithm.nmf import EUCNMF, ISNMF, KLNMF
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA
    m = GaussILRMA()
    assert (m.n_basis, m.domain, m.partitioning, m.normalize, m.algorithm_spatial, m.reference_id) == \
        (10, 2, False, "power", "IP", 0)
    assert m.eps == 1e-12 and m.threshold == 1e12 and m.loss == [] and m.callbacks is None and m.input is None
    f = lambda model: None  # noqa: E731
    assert GaussILRMA(callbacks=f).callbacks == [f]  # a single callable is wrapped (ilrma.py:27-32)
    assert GaussILRMA(recordable_loss=False).loss is None
    assert GaussILRMA(algorithm_spatial="IP2").update_pair is None
    for bad in ("IVA", "IPA"):
        with pytest.raises(AssertionError):
            GaussILRMA(algorithm_spatial=bad)
    with pytest.raises(AssertionError):
        GaussILRMA(algorithm_spatial="nope")
    a = AuxLaplaceIVA()
    assert (a.algorithm_spatial, a.reference_id, a.apply_projection_back, a.threshold) == ("IP", 0, True, 1e12)
    assert repr(a) == "AuxLaplaceIVA(algorithm_spatial=IP)" and repr(AuxGaussIVA()) == "AuxGaussIVA(algorithm_spatial=IP)"
    with pytest.raises(ValueError):
        AuxGaussIVA(algorithm_spatial="nope")
    for cls in (EUCNMF, KLNMF, ISNMF):
        n = cls()
        assert (n.n_basis, n.domain, n.algorithm, n.eps, n.loss) == (2, 2, "mm", 1e-12, [])
        with pytest.raises(AssertionError):
            cls(domain=0.5)
    with pytest.raises(AssertionError):
        EUCNMF(algorithm="me")
    assert ISNMF(algorithm="me").algorithm == "me"
    assert not hasattr(m, "demix_filter") and not hasattr(m, "basis") and not hasattr(m, "activation")
    m.demix_filter = np.zeros((3, 2, 2), dtype=np.complex128)
    assert hasattr(m, "demix_filter") and m.demix_filter.shape == (3, 2, 2)


def test_lazy_loss_list_defers_and_resolves():
    """`model.loss` semantics without a GPU: parked entries are materialised on first read, after the model had the
    chance to compute a value it deferred (before_flush); copies / pickles carry plain values."""
    import copy
    import pickle
    from audio_source_separation_amd._loss import LazyLossList

    class FakeTensor:
        def __init__(self, v):
            self.v = np.asarray(v, dtype=np.float64)

        def detach(self):
            return self

        def cpu(self):
            return self

        def numpy(self):
            return self.v

    ll = LazyLossList()
    ll.append(3.0)
    box = FakeTensor([0.0])
    calls = []

    def resolve():
        calls.append(1)
        box.v = np.asarray([7.5])

    ll.before_flush = resolve
    ll.append_device(box, batched=False)
    assert len(ll) == 2 and not calls            # len() and append never trigger the computation
    assert ll[-1] == 7.5 and calls == [1]        # first read does, once
    assert list(ll) == [3.0, 7.5] and calls == [1]
    assert copy.deepcopy(ll) == ll and pickle.loads(pickle.dumps(ll))[1] == 7.5
    assert np.asarray(ll).tolist() == [3.0, 7.5]
    ll.append_device(FakeTensor([1.0, 2.0]), batched=True)       # utterance axis: one value per utterance
    assert np.array_equal(np.asarray(ll[2]), [1.0, 2.0])
    # mutators with entries still parked: the reference's `loss` is a plain list, users reset / trim it
    for mutate, expect in ((lambda l: l.clear(), [4.0]), (lambda l: l.insert(0, -1.0), [-1.0, 3.0, 9.0, 4.0]),
                           (lambda l: l.__delitem__(0), [9.0, 4.0]), (lambda l: l.remove(3.0), [9.0, 4.0]),
                           (lambda l: l.extend([5.0]), [3.0, 9.0, 5.0, 4.0]),
                           (lambda l: l.__setitem__(slice(0, 2), [0.0]), [0.0, 4.0])):
        ll = LazyLossList([3.0])
        ll.append_device(FakeTensor([9.0]), batched=False)
        mutate(ll)
        ll.append_device(FakeTensor([4.0]), batched=False)
        assert list(ll) == expect
    ll = LazyLossList([3.0])
    ll.append_device(FakeTensor([9.0]), batched=False)
    ll += [1.0]
    assert isinstance(ll, LazyLossList) and list(ll) == [3.0, 9.0, 1.0]


def test_lazy_loss_list_blocks_of_a_one_call_loop():
    """assx_*_iterate writes n loss values into one device array: they enter `loss` as a block (one download)."""
    from audio_source_separation_amd._loss import LazyLossList

    class FakeBlock:
        def __init__(self, v):
            self.v = np.asarray(v, dtype=np.float64)
            self.shape = self.v.shape

        def detach(self):
            return self

        def cpu(self):
            return self

        def numpy(self):
            return self.v

    ll = LazyLossList([1.0])
    ll.append_device_block(FakeBlock([[2.0], [3.0], [4.0]]), batched=False)
    assert len(ll) == 4
    ll.append(5.0)
    ll.append_device_block(FakeBlock(np.zeros((0, 1))), batched=False)   # zero iterations: nothing
    assert list(ll) == [1.0, 2.0, 3.0, 4.0, 5.0] and all(isinstance(v, float) for v in ll)
    ll = LazyLossList()
    ll.append_device_block(FakeBlock([[1.0, 2.0], [3.0, 4.0]]), batched=True)  # utterance axis kept per entry
    ll.clear()                                                                    # mutators materialise first
    ll.append_device_block(FakeBlock([[1.0, 2.0], [3.0, 4.0]]), batched=True)
    assert len(ll) == 2 and np.array_equal(ll[1], [3.0, 4.0]) and np.asarray(ll).shape == (2, 2)


def test_tracked_snapshot_marks_the_device_copy_stale_on_write():
    """`model.basis[...] *= s` must reach the device (the reference's attributes are plain arrays): the NumPy snapshot of a
    device array is tracked; writes -- __setitem__, ufuncs with out=, through views -- drop the device copy; copies and
    arithmetic results are plain, untracked arrays."""
    from audio_source_separation_amd._state import TrackedArray, _Entry
    ent = _Entry(dev="device tensor")
    a = np.arange(12.0).reshape(3, 4).view(TrackedArray)
    a._entry, a._root = ent, a
    ent.host = a
    assert type(a * 2) is np.ndarray and ent.dev is not None       # reading / arithmetic: nothing happens
    assert float(a.sum()) == 66.0 and ent.dev is not None
    a[1][:] = 0                                                    # through a view
    assert ent.dev is None and np.array_equal(a[1], np.zeros(4))
    ent.dev = "again"
    a *= 3                                                         # in-place ufunc
    assert ent.dev is None and a[0, 1] == 3.0
    ent.dev = "again"
    c = a.copy()
    c[0, 0] = 5                                                    # the caller's own copy
    assert ent.dev == "again"
    ent.host = None                                                # the model moved on: `a` is an old snapshot
    a[0, 0] = 7
    assert ent.dev == "again"


def test_tracked_snapshot_other_in_place_routes():
    """Round 4's advisor: fill / sort / put / np.copyto / np.add.at / np.put ... also write in place and must drop the
    device copy; a write through a PLAIN view of the snapshot's memory passes no hook and is caught by the content
    signature that DeviceState._dev compares before it reuses the device copy."""
    from audio_source_separation_amd._state import DeviceState, TrackedArray, _Entry, _signature

    def fresh():
        ent = _Entry(dev="device tensor")
        a = np.arange(12.0).reshape(3, 4).view(TrackedArray)
        a._entry, a._root = ent, a
        ent.host = a
        ent.sig = _signature(a)
        return ent, a

    routes = {
        "fill": lambda a: a.fill(1.0),
        "sort": lambda a: a[:, ::-1].sort(axis=1),
        "put": lambda a: a.put([0, 1], [9.0, 8.0]),
        "np.put": lambda a: np.put(a, [2], [7.0]),
        "np.copyto": lambda a: np.copyto(a, np.ones((3, 4))),
        "np.copyto view": lambda a: np.copyto(a[1], np.ones(4)),
        "np.add.at": lambda a: np.add.at(a, (0, 0), 5.0),
        "np.putmask": lambda a: np.putmask(a, a > 5, 0.0),
        "np.place": lambda a: np.place(a, a > 5, [0.0]),
        "np.fill_diagonal": lambda a: np.fill_diagonal(a, -1.0),
        "partition": lambda a: a.reshape(-1)[::-1].partition(3),
    }
    for name, edit in routes.items():
        ent, a = fresh()
        before = a.copy()
        edit(a)
        assert ent.dev is None, name
        assert not np.array_equal(np.asarray(a), before) or name == "partition", name
    ent, a = fresh()
    assert float(np.sum(a)) == 66.0 and np.array_equal(np.sort(a, axis=None), np.arange(12.0)) and ent.dev is not None
    # behind the hooks: plain views of the same memory

    class Model(DeviceState):
        class _engine:  # what _dev needs to upload: never reached here (the upload is patched below)
            pass

    import audio_source_separation_amd._state as st
    uploads = []
    orig = st.to_device
    st.to_device = lambda arr, dt, dev: type("T", (), {"clone": lambda self: uploads.append(np.array(arr)) or "uploaded"})()
    try:
        def swap_rows(a):  # a permutation fix behind the hooks: same multiset of values, another order (round 5's advisor:
            p = np.asarray(a)  # the sum + xor signature of round 5 could not see it)
            p[[0, 1]] = p[[1, 0]]

        def flip(a):
            p = np.asarray(a)
            p[...] = p[::-1, ::-1].copy()

        for write in (lambda a: np.asarray(a).__setitem__((0, 0), 100.0), lambda a: a.view(np.ndarray).fill(2.0),
                      swap_rows, flip):
            ent, a = fresh()
            m = Model()
            m._engine = type("E", (), {"prec": type("P", (), {"cplx": None, "real": None})(), "dev": None})()
            m.__dict__["_arrays"] = {"basis": ent}
            assert m._dev("basis", False) == "device tensor" and not uploads  # unchanged host: the device copy is reused
            write(a)
            assert ent.dev == "device tensor"  # no hook saw it ...
            assert m._dev("basis", False) == "uploaded" and uploads  # ... the signature did
            assert np.array_equal(uploads[-1][0], np.asarray(a))
            del uploads[:]
            assert m._dev("basis", False) == "uploaded" and not uploads  # and agrees again afterwards
    finally:
        st.to_device = orig


def test_thread_contexts_are_not_kept_alive_by_the_exit_hook():
    """The interpreter-exit hook holds WEAK references to the per-thread context sets (round 4's advisor: a strong list kept
    every finished thread's contexts -- pinned staging, host thread pool, ticket buffers -- until exit)."""
    import gc
    import weakref
    from audio_source_separation_amd import _device
    assert isinstance(_device._ALL_HELD, weakref.WeakSet)
    held = _device._ThreadContexts()
    _device._ALL_HELD.add(held)
    n = len(_device._ALL_HELD)
    del held
    gc.collect()
    assert len(_device._ALL_HELD) == n - 1


def test_stft_geometry_matches_scipy_semantics():
    """assx_stft_num_frames / assx_istft_num_samples (host arithmetic, no GPU) against the oracle's restatement of
    scipy.signal.stft / istft (boundary + padding rules) over many lengths, frame sizes and hops."""
    from audio_source_separation_amd import _lib
    from oracle import oracle_np as orc
    lib = _lib.lib
    for N in (2, 8, 30, 33, 64):
        for hop in sorted({1, max(1, N // 4), max(1, N // 2), N - 1 if N > 1 else 1, N}):
            for L in (N, N + 1, 2 * N + 3, 5 * N, 257):
                if L < N:
                    continue
                X = orc.stft(np.zeros((1, L)), N, hop)
                assert lib.assx_stft_num_frames(L, N, hop) == X.shape[-1], (L, N, hop)
                y = orc.istft(X, N, hop)
                assert lib.assx_istft_num_samples(N, hop, X.shape[-1]) == y.shape[-1], (L, N, hop)
    assert lib.assx_stft_num_frames(-1, 8, 2) == -1 and lib.assx_stft_num_frames(10, 8, 0) == -1
    n64 = lib.assx_stft_workspace_bytes(4, 2048, 4096, _lib.F64)
    assert n64 > 4 * 4096 * (1025 * 16 + 2048 * 8) and lib.assx_stft_workspace_bytes(0, 8, 1, _lib.F64) == 0


def test_workspace_grows_for_wide_basis():
    """n_basis > 4 adds the (B,N,F,T) map and the NMF scratch to assx_workspace_bytes (include/assx.h)."""
    from audio_source_separation_amd import _lib
    lib = _lib.lib
    small = lib.assx_workspace_bytes(1, 4, 1025, 4096, 4, _lib.F64)
    wide = lib.assx_workspace_bytes(1, 4, 1025, 4096, 10, _lib.F64)
    assert wide - small >= 4 * 1025 * 4096 * 8 + lib.assx_nmf_workspace_bytes(4, 1025, 4096, 10, _lib.F64) - 4096


def test_bench_gpus_flag_is_honoured_or_refused():
    """`bench.py --gpus N` must run N ranks or fail loudly -- never silently run one (VERDICT r1 / ADVICE r1)."""
    import subprocess
    import torch
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # a launcher that set another world size than --gpus: refused before any GPU work
    r = subprocess.run([sys.executable, bench, "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "must agree" in r.stderr
    # more ranks than visible GPUs: refused by the self-launcher (this container has no GPU at all)
    want = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = subprocess.run([sys.executable, bench, "--gpus", str(want)], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr


def test_wav_io_round_trip(tmp_path):
    """read_wav / write_wav (utils_audio.py:4-18): 16-bit PCM, 2**15 scaling, clipping, channel_last handling."""
    from audio_source_separation_amd.utils.utils_audio import read_wav, write_wav
    rng = np.random.default_rng(0)
    x = np.clip(0.3 * rng.standard_normal((2, 800)), -1.5, 1.5)  # (n_channels, n_samples), some samples clip
    path = str(tmp_path / "mix.wav")
    write_wav(path, x, 16000, channel_last=False)
    y, sr = read_wav(path)
    assert sr == 16000 and y.shape == (800, 2) and y.dtype == np.float64
    expect = np.clip(x * 32768, -32768, 32767).astype(np.int16).T / 32768
    assert np.array_equal(y, expect)
    write_wav(path, y[:, 0], sr)
    y1, _ = read_wav(path)
    assert np.array_equal(y1, y[:, 0])
    with pytest.raises(ValueError):
        write_wav(path, np.zeros((2, 2, 2)), sr)


def test_engine_refuses_model_arrays_of_another_shape():
    """The C-ABI takes pointers and sizes; the host side is where a wrongly shaped model array can still be refused
    (a partitioned basis read as (N, F, K) was an out-of-bounds read that only some allocation layouts turned into a fault)."""
    from audio_source_separation_amd.ops import Engine
    B, M, F, T, K = 1, 3, 5, 7, 2
    X = np.zeros((B, M, F, T), np.complex128)
    W = np.zeros((B, F, M, M), np.complex128)
    Tb, V = np.zeros((B, M, F, K)), np.zeros((B, M, K, T))
    Engine._model_shapes(X, W, Tb, V)
    Engine._model_shapes(X, None, Tb, V)
    with pytest.raises(ValueError, match="basis"):
        Engine._model_shapes(X, W, np.zeros((B, F, K)), V)           # partitioned basis
    with pytest.raises(ValueError, match="activation"):
        Engine._model_shapes(X, W, Tb, np.zeros((B, K, T)))
    with pytest.raises(ValueError, match="demix_filter"):
        Engine._model_shapes(X, np.zeros((B, F, M, M - 1), np.complex128), Tb, V)


def test_signature_is_position_dependent():
    """Round 5's advisor (medium): a value-preserving reorder of a snapshot must change its content signature, for every
    size class of the digest (fewer than one row of words, whole rows, a ragged rest, odd byte counts)."""
    from audio_source_separation_amd._state import _signature
    rng = np.random.default_rng(5)
    for shape, dtype in (((3,), np.float32), ((7, 5), np.float64), ((1025, 4, 4), np.complex128), ((4, 4, 4096), np.float64),
                         ((2, 1031), np.float32), ((4099,), np.uint8)):
        a = (rng.random(shape) * 100).astype(dtype)
        s = _signature(a)
        assert _signature(a.copy()) == s
        flat = a.reshape(-1)
        for i, j in ((0, flat.size - 1), (0, 1), (flat.size // 2, flat.size // 2 + 1)):
            if flat[i] == flat[j]:
                continue
            b = a.copy().reshape(-1)
            b[i], b[j] = b[j], b[i]
            assert _signature(b.reshape(shape)) != s, (shape, dtype, i, j)
        if a.ndim >= 2 and a.shape[-2] > 1:
            b = a.copy()
            b[..., [0, 1], :] = b[..., [1, 0], :]
            assert _signature(b) != s, (shape, dtype, "rows")


def test_shard_range_of_the_library_equals_the_python_partition():
    """assx_shard_range (host-side, no GPU) is the partition `distributed.shard_range` makes: config 5's 64 over 8, ragged
    and empty cases, every rank."""
    from audio_source_separation_amd import distributed as D
    import ctypes
    from audio_source_separation_amd import _lib
    for n in (0, 1, 5, 7, 8, 13, 63, 64, 65, 1000):
        for world in (1, 2, 3, 8, 16):
            seen = []
            for rank in range(world):
                lo, hi = ctypes.c_size_t(99), ctypes.c_size_t(99)
                _lib.lib.assx_shard_range(n, world, rank, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == D.shard_range(n, world, rank)
                seen += list(range(lo.value, hi.value))
            assert seen == list(range(n))
    lo, hi = ctypes.c_size_t(99), ctypes.c_size_t(99)
    _lib.lib.assx_shard_range(10, 4, 4, ctypes.byref(lo), ctypes.byref(hi))  # a rank outside the world holds nothing
    assert (lo.value, hi.value) == (0, 0)
