"""Public attributes that live in HBM and materialise as NumPy arrays on demand.

The reference keeps `demix_filter`, `basis`, `activation`, `estimation` as NumPy attributes that
callbacks read after every iteration (egs/bss-example/ilrma/test_gauss-ilrma.ipynb cells 69-78)
and that `_reset` warm-starts from via `hasattr` (src/bss/ilrma.py:67-72, 88-104).  Here the
authoritative copy is a device tensor; the NumPy view is downloaded lazily and cached until a
kernel touches the tensor again, so a loop without callbacks never leaves the GPU.  Downloaded views are READ-ONLY
(in-place edits would not reach the device): assign a new array to modify state.
"""
import numpy as np

from ._device import to_device, to_numpy, torch


class _Entry:
    __slots__ = ("host", "dev")

    def __init__(self, host=None, dev=None):
        self.host = host
        self.dev = dev


class DeviceArray:
    """Descriptor.  get -> NumPy (float64/complex128, batch axis squeezed unless the model is batched);
    raises AttributeError while unset so `hasattr(model, name)` behaves as in the reference."""

    def __init__(self, name, complex_):
        self.name = name
        self.complex_ = complex_

    def __set_name__(self, owner, attr):
        self.attr = attr

    def _store(self, obj):
        return obj.__dict__.setdefault("_arrays", {})

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        ent = self._store(obj).get(self.name)
        if ent is None:
            raise AttributeError("'%s' object has no attribute '%s'" % (type(obj).__name__, self.attr))
        if ent.host is None:
            a = to_numpy(ent.dev, np.complex128 if self.complex_ else np.float64)
            a = a if obj._batched else a[0]
            # the device tensor stays authoritative: an in-place edit of this snapshot (`model.demix_filter[...] *= s`)
            # would be silently dropped at the next kernel, whereas it takes effect in the reference.  Make it fail
            # loudly instead; ASSIGNING an array (`model.demix_filter = new`) is the supported way to modify state.
            a.setflags(write=False)
            ent.host = a
        return ent.host

    def __set__(self, obj, value):
        resolve = getattr(obj, "_resolve_deferred_loss", None)
        if resolve is not None:
            resolve()  # a loss still to be folded into the next pass refers to the state being replaced
        if isinstance(value, torch.Tensor):
            self._store(obj)[self.name] = _Entry(host=None, dev=None)
            self._store(obj)[self.name].host = to_numpy(value, np.complex128 if self.complex_ else np.float64)
        else:
            self._store(obj)[self.name] = _Entry(host=value, dev=None)

    def __delete__(self, obj):
        self._store(obj).pop(self.name, None)


class DeviceState:
    """Mixin with the device side of DeviceArray attributes."""

    _batched = False

    def _has(self, name):
        return name in self.__dict__.get("_arrays", {})

    def _dev(self, name, complex_):
        """Device tensor for `name` (uploads the host value on first use, adding the batch axis)."""
        ent = self.__dict__["_arrays"][name]
        if ent.dev is None:
            prec = self._engine.prec
            a = np.asarray(ent.host)
            if not self._batched:
                a = a[None]
            ent.dev = to_device(a, prec.cplx if complex_ else prec.real, self._engine.dev).clone()
        return ent.dev

    def _set_dev(self, name, tensor):
        self.__dict__.setdefault("_arrays", {})[name] = _Entry(host=None, dev=tensor)

    def _touch(self, *names):
        """A kernel modified these device tensors: drop the cached NumPy views."""
        arrays = self.__dict__.get("_arrays", {})
        for n in names:
            if n in arrays and arrays[n].dev is not None:
                arrays[n].host = None
