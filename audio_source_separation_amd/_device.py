"""Device plumbing: PyTorch-ROCm is used ONLY as a container for HBM buffers and streams.

All arithmetic of the hot path happens in libassx.so (hand-written HIP); nothing here computes.
"""
import ctypes
import threading
import weakref

import numpy as np

from . import _lib

try:  # torch is the allocator / stream provider, not the compute engine
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("PyTorch-ROCm is required as the device-memory container: %r" % (exc,))

REAL = {"float64": (torch.float64, torch.complex128, _lib.F64, np.float64, np.complex128),
        "float32": (torch.float32, torch.complex64, _lib.F32, np.float32, np.complex64)}


class Precision:
    def __init__(self, dtype):
        name = np.dtype(dtype).name if not isinstance(dtype, str) else dtype
        name = {"complex128": "float64", "complex64": "float32", "double": "float64", "float": "float32"}.get(name, name)
        if name not in REAL:
            raise ValueError("dtype must be 'float64' or 'float32', got %r" % (dtype,))
        self.name = name
        self.real, self.cplx, self.code, self.np_real, self.np_cplx = REAL[name]


def require_gpu(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError(
            "audio_source_separation_amd needs an AMD GPU (MI355X / gfx950): torch.cuda.is_available() is False. "
            "The HIP path has no CPU fallback.")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError("device must be a cuda (ROCm) device, got %s" % dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class _ThreadContexts:
    """The assx contexts of ONE host thread (device index -> handle); destroyed with the thread."""

    def __init__(self):
        self.by_device = {}

    def __del__(self):
        for h in self.by_device.values():
            try:
                _lib.lib.assx_ctx_destroy(h)
            except Exception:  # interpreter shutdown
                pass
        self.by_device = {}


_TLS = threading.local()
# every LIVE thread's _ThreadContexts, for the interpreter-exit hook below.  Weak references: the thread-local slot is the
# only owner, so a thread that ends takes its contexts (pinned staging ring, host thread pool, device ticket buffers)
# with it -- a server that spawns worker threads does not accumulate one context per finished thread.
_ALL_HELD = weakref.WeakSet()


def _destroy_all_contexts():
    """Destroy the contexts while the HIP runtime is still alive: a context owns pinned staging buffers, events and a
    pool of host threads (csrc/assx_xfer.hip), and `__del__` during interpreter finalisation is too late to call HIP."""
    for held in list(_ALL_HELD):
        held.__del__()


import atexit  # noqa: E402

atexit.register(_destroy_all_contexts)


def context(dev):
    """The calling thread's assx context for `dev`, as include/assx.h specifies: a context is not thread-safe (last
    error message, launch state, the staging ring), so two threads driving the same GPU get two contexts.  Resolved
    per CALL (Engine.ctx is a property): a model built in one thread and driven from another uses the driving
    thread's context.  A thread's contexts are destroyed when the thread ends."""
    held = getattr(_TLS, "held", None)
    if held is None:
        held = _TLS.held = _ThreadContexts()
        _ALL_HELD.add(held)
    h = held.by_device.get(dev.index)
    if h is None:
        h = ctypes.c_void_p()
        rc = _lib.lib.assx_ctx_create(int(dev.index), ctypes.byref(h))
        if rc != 0:
            raise _lib.AssxError("assx_ctx_create(%d) failed with code %d" % (dev.index, rc))
        held.by_device[dev.index] = h
    return h


def stream_ptr(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous(), "device buffers handed to the C-ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


_NP_CODE = {np.dtype(np.float64): (_lib.F64, 1), np.dtype(np.float32): (_lib.F32, 1),
            np.dtype(np.complex128): (_lib.F64, 2), np.dtype(np.complex64): (_lib.F32, 2)}
_T_CODE = {torch.float64: (_lib.F64, 1), torch.float32: (_lib.F32, 1),
           torch.complex128: (_lib.F64, 2), torch.complex64: (_lib.F32, 2)}
_T_NP = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128,
         torch.complex64: np.complex64}


def _native_call(dev, fn, *args):
    """libassx refuses a call whose context device is not current (include/assx.h)."""
    if torch.cuda.current_device() == dev.index:
        return fn(*args)
    with torch.cuda.device(dev):
        return fn(*args)


def to_device(a, dtype, dev):
    """numpy array or torch tensor -> contiguous device tensor of `dtype` (copy only when needed).

    A NumPy array goes through `assx_upload` (csrc/assx_xfer.hip): chunks ride a ring of pinned staging buffers, host
    threads copy -- and convert to the device precision, so float32 mode moves half the bytes -- while the previous
    chunk is on the bus.  The call returns once the host array has been consumed; the tail of the DMA is ordered before
    later work on torch's current stream.

    Stream-ordering contract: the returned tensor is valid for work enqueued on torch's CURRENT stream of `dev` (the
    stream every Engine call uses); a consumer on another stream must first wait on the current one
    (`other.wait_stream(torch.cuda.current_stream(dev))`)."""
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    arr = np.ascontiguousarray(a)
    code = _T_CODE.get(dtype)
    if code is None:  # not a floating array of the path (status words, ...): torch's own copy
        return torch.from_numpy(np.array(arr, copy=True)).to(device=dev, dtype=dtype).contiguous()
    if arr.dtype not in _NP_CODE or _NP_CODE[arr.dtype][1] > code[1]:
        # integer / bool / float16 input, or complex -> real (rejected by NumPy's own casting rules loudly)
        arr = arr.astype(_T_NP[dtype])
    hcode, hwidth = _NP_CODE[arr.dtype]
    if hwidth < code[1]:  # real array into a complex tensor
        arr = arr.astype(np.complex128 if hcode == _lib.F64 else np.complex64)
        hwidth = 2
    t = torch.empty(arr.shape, dtype=dtype, device=dev)
    if arr.size:
        ctx = context(dev)
        rc = _native_call(dev, _lib.lib.assx_upload, ctx, ctypes.c_void_p(arr.ctypes.data), hcode,
                          ctypes.c_void_p(t.data_ptr()), code[0], arr.size * hwidth, stream_ptr(dev))
        _lib.check(ctx, rc, "assx_upload")
    return t


def to_numpy(t, np_dtype=None):
    """device tensor -> fresh NumPy array of `np_dtype` (default: the tensor's own type) through `assx_download`: pinned
    staging ring, host threads convert / first-touch the destination while the next chunk is on the bus."""
    t = t.detach()
    code = _T_CODE.get(t.dtype)
    out_dt = np.dtype(np_dtype) if np_dtype is not None else (np.dtype(_T_NP[t.dtype]) if code else None)
    if (not t.is_cuda) or code is None or out_dt not in _NP_CODE or _NP_CODE[out_dt][1] != code[1]:
        a = t.cpu().numpy()
        return a.astype(np_dtype, copy=False) if np_dtype is not None else a
    # a lazily conjugated / negated view shares its storage with the original: materialise it before the raw copy
    t = t.resolve_conj().resolve_neg().contiguous()
    out = np.empty(tuple(t.shape), dtype=out_dt)
    if out.size:
        dev = t.device
        ctx = context(dev)
        rc = _native_call(dev, _lib.lib.assx_download, ctx, ctypes.c_void_p(t.data_ptr()), code[0],
                          ctypes.c_void_p(out.ctypes.data), _NP_CODE[out_dt][0], out.size * code[1], stream_ptr(dev))
        _lib.check(ctx, rc, "assx_download")
    return out


class Workspace:
    """Caller-owned scratch for the C-ABI (assx_workspace_bytes); grows monotonically."""

    def __init__(self, dev):
        self.dev = dev
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
        return self.buf
