"""The multi-GPU edges through the C-ABI (include/assx.h: assx_comm_init / assx_scatter / assx_gather).

`distributed.py` issues the same edges through torch.distributed and is what the classes and bench.py use; this module is
the ctypes face of the C route -- what a host WITHOUT torch binds -- kept in Python so that the
route is exercised by the test suite on the real RCCL: a 1-rank communicator on the 1-GPU box sends its block to itself inside
the same grouped ncclSend / ncclRecv batch that root <-> 7 peers use.

No reference counterpart (the reference is a single-process NumPy program, src/bss/ilrma.py:203-273)."""
import ctypes

import torch

from . import _lib
from ._device import context, ptr, require_gpu, stream_ptr


def shard_range(n_items, world, rank):
    """(lo, hi) of the static block partition, computed by the library (== distributed.shard_range)."""
    lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.lib.assx_shard_range(int(n_items), int(world), int(rank), ctypes.byref(lo), ctypes.byref(hi))
    return int(lo.value), int(hi.value)


def unique_id():
    """ASSX_COMM_ID_BYTES bytes made on ONE process (ncclGetUniqueId); the host hands them to every rank by its own means."""
    buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = _lib.lib.assx_comm_unique_id(buf)
    if rc != 0:
        raise _lib.AssxError("assx_comm_unique_id failed (%d): RCCL not available?" % rc)
    return buf.raw


class Comm:
    """One rank of an RCCL communicator made through the C-ABI, bound to this thread's context of `device`."""

    def __init__(self, world, rank, id_bytes, device=None):
        self.dev = require_gpu(device)
        self.ctx = context(self.dev)
        self.world, self.rank = int(world), int(rank)
        h = ctypes.c_void_p()
        idb = ctypes.create_string_buffer(bytes(id_bytes), _lib.COMM_ID_BYTES)
        with torch.cuda.device(self.dev):
            _lib.check(self.ctx, _lib.lib.assx_comm_init(self.ctx, self.world, self.rank, idb, ctypes.byref(h)), "assx_comm_init")
        self._h = h

    def close(self):
        if self._h:
            _lib.lib.assx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return stream_ptr(self.dev)

    def scatter(self, x_all, n_items, item_shape, dtype, root=0):
        """Root passes x_all (n_items, *item_shape) on its device, the others None; every rank returns its own block."""
        lo, hi = shard_range(n_items, self.world, self.rank)
        out = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=self.dev)
        n_elem = 1
        for d in item_shape:
            n_elem *= int(d)
        item_bytes = n_elem * out.element_size()
        src = None
        if self.rank == root:
            src = x_all.to(device=self.dev, dtype=dtype).contiguous()
        with torch.cuda.device(self.dev):
            _lib.check(self.ctx, _lib.lib.assx_scatter(self._h, int(root), ptr(src), ptr(out), int(n_items), int(item_bytes),
                                                      self._stream()), "assx_scatter")
        self._keep = src  # the sends read it asynchronously on the stream
        return out

    def gather(self, y_local, n_items, root=0):
        """Every rank passes its block; the root returns (n_items, ...) in the original order, the others None."""
        y_local = y_local.contiguous()
        item_shape = tuple(y_local.shape[1:])
        n_elem = 1
        for d in item_shape:
            n_elem *= int(d)
        item_bytes = n_elem * y_local.element_size()
        out = torch.empty((n_items,) + item_shape, dtype=y_local.dtype, device=self.dev) if self.rank == root else None
        with torch.cuda.device(self.dev):
            _lib.check(self.ctx, _lib.lib.assx_gather(self._h, int(root), ptr(y_local), ptr(out), int(n_items), int(item_bytes),
                                                     self._stream()), "assx_gather")
        self._keep = y_local
        return out
