"""Device plumbing: PyTorch-ROCm is used ONLY as a container for HBM buffers and streams.

All arithmetic of the hot path happens in libassx.so (hand-written HIP); nothing here computes.
"""
import ctypes
import threading

import numpy as np

from . import _lib

try:  # torch is the allocator / stream provider, not the compute engine
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("PyTorch-ROCm is required as the device-memory container: %r" % (exc,))

_CTX = {}

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


def context(dev):
    """One assx context per (device, host thread), as include/assx.h specifies: a context is not thread-safe (it holds
    the last error message), so two threads driving the same GPU get two contexts."""
    key = (dev.index, threading.get_ident())
    if key not in _CTX:
        h = ctypes.c_void_p()
        rc = _lib.lib.assx_ctx_create(int(dev.index), ctypes.byref(h))
        if rc != 0:
            raise _lib.AssxError("assx_ctx_create(%d) failed with code %d" % (dev.index, rc))
        _CTX[key] = h
    return _CTX[key]


def stream_ptr(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous(), "device buffers handed to the C-ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def to_device(a, dtype, dev):
    """numpy array or torch tensor -> contiguous device tensor of `dtype` (copy only when needed)."""
    if isinstance(a, torch.Tensor):
        t = a.to(device=dev, dtype=dtype)
    else:
        arr = np.ascontiguousarray(a)
        t = torch.from_numpy(arr).to(device=dev, dtype=dtype)
    return t.contiguous()


def to_numpy(t, np_dtype=None):
    a = t.detach().cpu().numpy()
    return a.astype(np_dtype, copy=False) if np_dtype is not None else a


class Workspace:
    """Caller-owned scratch for the C-ABI (assx_workspace_bytes); grows monotonically."""

    def __init__(self, dev):
        self.dev = dev
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
        return self.buf
