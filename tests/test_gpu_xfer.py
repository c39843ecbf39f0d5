"""Host <-> HBM staging (assx_upload / assx_download, csrc/assx_xfer.hip): the NumPy-in / NumPy-out edge of the
reference's __call__ (ilrma.py:203-273).  Copies are bit-exact; a precision change equals NumPy's own astype()."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _in_fresh_thread(fn, env=None):
    """A new host thread gets a new assx context (contexts are per thread) and with it a new staging ring that reads
    ASSX_XFER_* at creation: lets a test choose a small chunk size / thread count."""
    box = {}
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})

    def run():
        try:
            box["out"] = fn()
        except BaseException as exc:  # noqa: BLE001  re-raised in the test thread
            box["exc"] = exc

    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if "exc" in box:
        raise box["exc"]
    return box["out"]


NP2T = {np.float64: torch.float64, np.float32: torch.float32, np.complex128: torch.complex128, np.complex64: torch.complex64}


def _rand(shape, dt, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shape)
    if np.issubdtype(dt, np.complexfloating):
        a = a + 1j * rng.standard_normal(shape)
    return a.astype(dt)


@pytest.mark.parametrize("threads,chunk_mb", [(1, 1), (3, 1), (8, 2), (None, None)])
@pytest.mark.parametrize("host_dt,dev_dt", [(np.complex128, np.complex128), (np.complex128, np.complex64),
                                            (np.complex64, np.complex128), (np.float64, np.float64),
                                            (np.float32, np.float64), (np.float64, np.float32)])
def test_round_trip_equals_numpy_casts(threads, chunk_mb, host_dt, dev_dt):
    """Sizes around the chunk and slice boundaries (1 MiB chunks, 64-element slice alignment), more chunks than ring
    slots; upload == astype(device type), download == astype(host type), both bit for bit."""
    from audio_source_separation_amd._device import to_device, to_numpy
    dev = torch.device("cuda", 0)
    env = {} if threads is None else {"ASSX_XFER_THREADS": str(threads), "ASSX_XFER_CHUNK_MB": str(chunk_mb)}

    def body():
        for n in (0, 1, 63, 65537, (1 << 20) // 8, (1 << 20) // 8 + 1, 700001, 9 * (1 << 20) // 8 + 77):
            a = _rand((n,), host_dt, n)
            t = to_device(a, NP2T[dev_dt], dev)
            assert t.dtype == NP2T[dev_dt] and tuple(t.shape) == (n,)
            expect = a.astype(dev_dt)
            assert np.array_equal(t.cpu().numpy(), expect), (n, "upload")
            back = to_numpy(t, host_dt)
            assert back.dtype == host_dt and back.flags.writeable and back.flags.c_contiguous
            assert np.array_equal(back, expect.astype(host_dt)), (n, "download")
        return True

    assert _in_fresh_thread(body, env)


def test_shapes_strides_readonly_and_integer_input():
    from audio_source_separation_amd._device import to_device, to_numpy
    dev = torch.device("cuda", 0)
    a = _rand((3, 5, 7), np.complex128, 1)
    nc = a.transpose(2, 0, 1)  # not contiguous
    assert np.array_equal(to_numpy(to_device(nc, torch.complex128, dev)), nc)
    ro = a.copy()
    ro.setflags(write=False)  # a downloaded model attribute assigned back: no warning, no aliasing
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        t = to_device(ro, torch.complex128, dev)
    assert np.array_equal(to_numpy(t), ro)
    i = np.arange(12, dtype=np.int64).reshape(3, 4)
    assert np.array_equal(to_numpy(to_device(i, torch.float64, dev)), i.astype(np.float64))
    r = _rand((4, 4), np.float64, 2)
    assert np.array_equal(to_numpy(to_device(r, torch.complex128, dev)), r.astype(np.complex128))
    # a non-contiguous device tensor, and a tensor type outside the path (status words)
    td = to_device(a, torch.complex128, dev).permute(1, 0, 2)
    assert np.array_equal(to_numpy(td), a.transpose(1, 0, 2))
    st = torch.arange(5, dtype=torch.int32, device=dev)
    assert np.array_equal(to_numpy(st), np.arange(5, dtype=np.int32))


def test_upload_is_ordered_before_later_work_and_after_earlier_work():
    """The tail of the DMA is not waited for by the host: a kernel queued right after must still see all of it; and a
    recycled allocation still being read by queued work must not be overwritten early."""
    from audio_source_separation_amd._device import to_device
    dev = torch.device("cuda", 0)
    a = _rand((6_000_000,), np.float64, 3)
    for _ in range(5):
        t = to_device(a, torch.float64, dev)
        s = t.sum()  # queued immediately on the current stream
        assert abs(float(s) - float(a.sum())) < 1e-6 * abs(a).sum()
        big = torch.ones(6_000_000, dtype=torch.float64, device=dev)
        acc = torch.zeros((), dtype=torch.float64, device=dev)
        for _ in range(20):
            acc += big.sum()
        del big  # the allocator may hand this block to the next upload while the sums above are still queued
        t2 = to_device(a, torch.float64, dev)
        assert float(acc) == 20 * 6_000_000.0
        assert np.array_equal(t2.cpu().numpy(), a)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_numpy_call_equals_device_resident_call(dtype):
    """GaussILRMA()(X_numpy) -- upload through the staging ring, result downloaded through it -- is bit-identical to the
    same call on a device tensor (same kernels on the same bytes)."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    rng = np.random.default_rng(5)
    M, F, T = 3, 129, 700
    X = rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))
    cplx = torch.complex128 if dtype == "float64" else torch.complex64
    np.random.seed(1)
    Yh = GaussILRMA(n_basis=3, dtype=dtype)(X, iteration=4)
    np.random.seed(1)
    Xd = torch.from_numpy(X).to("cuda:0").to(cplx)
    Yd = GaussILRMA(n_basis=3, dtype=dtype)(Xd, iteration=4)
    assert Yh.dtype == np.complex128 and Yh.flags.writeable
    assert np.array_equal(Yh, Yd.cpu().numpy().astype(np.complex128))


def test_context_is_per_thread_and_resolved_per_call():
    """A model built in one thread and driven from another uses the driving thread's context."""
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64")
    here = eng.ctx.value
    there = _in_fresh_thread(lambda: eng.ctx.value)
    assert here != there and eng.ctx.value == here
