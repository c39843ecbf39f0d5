"""`model.loss`: a list whose newest entries may still live in HBM.

The reference appends a float after every iteration (src/bss/ilrma.py:239-241).  Reading an 8-byte scalar back
each iteration forces a host sync that stalls the launch queue (measured: 2780 -> 3300 it/s at config 4), so the
device scalars are parked here and converted to Python floats only when somebody looks at the list (indexing,
iteration, len-independent reads, repr, NumPy conversion, comparison ...).  `len()` and `append()` never sync.

A parked scalar may not even have been computed yet: Gauss-ILRMA folds the loss of iteration i into the basis pass of
iteration i+1 (same y = W x, same T V); `before_flush` lets the model run the stand-alone loss kernel for the one
value still outstanding when the list is read first.
"""
import numpy as np


class LazyLossList(list):
    def __init__(self, iterable=()):
        super().__init__(iterable)
        self._pending = {}  # index -> (device tensor, batched)
        self._blocks = []   # (first index, (n, B) device tensor, batched): n entries written by one iterate call
        self.before_flush = None  # set by the model: fills device scalars whose computation was deferred

    # ---- producers
    def append_device(self, tensor, batched):
        self._pending[len(self)] = (tensor, batched)
        super().append(None)

    def append_device_block(self, block, batched):
        """`block` (n, B) float64 on the device: n consecutive entries written by ONE call that ran n iterations
        (assx_*_iterate); downloaded in one piece when the list is read."""
        n = int(block.shape[0])
        if n:
            self._blocks.append((len(self), block, batched))
            super().extend([None] * n)

    # ---- materialisation
    def _flush(self):
        if self._pending or self._blocks:
            if self.before_flush is not None:
                self.before_flush()
            for idx, (t, batched) in self._pending.items():
                a = t.detach().cpu().numpy().astype(np.float64)
                super().__setitem__(idx, a if batched else np.float64(a.reshape(-1)[0]))
            self._pending.clear()
            for start, block, batched in self._blocks:
                a = block.detach().cpu().numpy().astype(np.float64)
                for i in range(a.shape[0]):
                    super().__setitem__(start + i, a[i].copy() if batched else np.float64(a[i].reshape(-1)[0]))
            self._blocks = []

    def _wrap(name):  # noqa: N805
        def method(self, *args, **kwargs):
            self._flush()
            return getattr(list, name)(self, *args, **kwargs)
        method.__name__ = name
        return method

    # readers AND every mutator that moves, drops or overwrites entries: pending device scalars are keyed by absolute
    # index, so they are materialised first and the list then behaves exactly like the reference's plain list
    # (`loss.clear()`, `del loss[:k]`, `insert`, `extend`, slicing assignment ... between calls are all legal there)
    for _n in ("__getitem__", "__iter__", "__repr__", "__str__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__",
               "__ge__", "__contains__", "__reversed__", "__add__", "__mul__", "__rmul__", "copy", "count", "index",
               "pop", "sort", "reverse", "clear", "insert", "remove", "extend", "__iadd__", "__imul__",
               "__setitem__", "__delitem__"):
        locals()[_n] = _wrap(_n)
    del _n, _wrap

    def __reduce_ex__(self, protocol):  # pickles / deep-copies as the materialised values only
        self._flush()
        return (LazyLossList, (list.copy(self),))

    def __array__(self, dtype=None, copy=None):
        self._flush()
        return np.array(list.copy(self), dtype=dtype)
