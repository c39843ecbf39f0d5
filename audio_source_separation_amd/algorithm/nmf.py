"""NMF by multiplicative updates on MI355X -- drop-in for `algorithm.nmf.EUCNMF / KLNMF / ISNMF / tNMF / CauchyNMF`
of the reference (/root/reference/src/algorithm/nmf.py:10-56, 150-600).

Same constructors, `nmf(target, iteration=100, **kwargs) -> (basis.copy(), activation.copy())`,
`basis` / `activation` / `loss` attributes.  `update_once()` and the per-iteration loss run as HIP
kernels (include/assx.h: assx_nmf_update / assx_nmf_loss); there is no CPU fallback.
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from .._state import DeviceArray, DeviceState
from .._loss import LazyLossList
from .. import _lib
from ..ops import Engine

EPS = 1e-12

__metrics__ = ['EUC', 'KL', 'IS']


class NMFbase(DeviceState):
    basis = DeviceArray("T", complex_=False)
    activation = DeviceArray("V", complex_=False)

    _KIND = None

    def __init__(self, n_basis=2, eps=EPS, *, dtype='float64', device=None, recordable_loss=True):
        """
        Args:
            n_basis: number of basis
            recordable_loss: extension (the reference always evaluates the criterion after every update,
                nmf.py:48-53): False skips that pass, `loss` then stays empty.
        """

        self.n_basis = n_basis
        self.loss = LazyLossList()  # a list; entries are materialised from HBM on first read
        self.recordable_loss = recordable_loss

        self.eps = eps
        self.domain = 2

        self.dtype = dtype
        self.device = device
        self._engine = None

    def __call__(self, target, iteration=100, **kwargs):
        self.target = target

        self._reset(**kwargs)

        self.update(iteration=iteration)

        T, V = self.basis, self.activation

        return T.copy(), V.copy()

    def _reset(self, **kwargs):
        assert self.target is not None, "Specify data!"

        for key in kwargs.keys():
            setattr(self, key, kwargs[key])

        if self._engine is None:
            self._engine = Engine(dtype=self.dtype, device=self.device)
        eng = self._engine

        n_basis = self.n_basis
        target = self.target
        ndim = target.dim() if isinstance(target, torch.Tensor) else np.ndim(target)
        if ndim not in (2, 3):
            raise ValueError("target must be (n_bins, n_frames), got {} dims".format(ndim))
        self._batched = ndim == 3
        Xd = to_device(target, eng.prec.real, eng.dev)
        if not self._batched:
            Xd = Xd.unsqueeze(0)
        self._X = Xd.contiguous()
        B, n_bins, n_frames = (int(s) for s in self._X.shape)

        # NMF never warm-starts: fresh draws from the global RNG, basis first (nmf.py:36-43)
        lead = (B,) if self._batched else ()
        self.basis = np.random.rand(*(lead + (n_bins, n_basis)))
        self.activation = np.random.rand(*(lead + (n_basis, n_frames)))

    def _kind_code(self):
        if self._KIND is None:
            raise NotImplementedError("Implement 'update_once' function")
        return self._KIND

    def _kind_param(self):
        return 0.0

    def _fast_loop_ok(self):
        """The loop of update() as ONE library call (assx_nmf_iterate): the steps are this module's own (a subclass
        overriding update_once keeps the Python loop) and `loss` is still the list the constructor made."""
        fn = getattr(type(self).update_once, "__func__", type(self).update_once)
        return getattr(fn, "__module__", None) == __name__ and type(self).update is NMFbase.update \
            and isinstance(self.loss, LazyLossList)

    def _record_loss(self):
        loss = self._engine.nmf_loss(self._kind_code(), self._X, self._dev("T", False), self._dev("V", False),
                                     domain=self.domain, eps=self.eps, param=self._kind_param())
        if isinstance(self.loss, LazyLossList):
            self.loss.append_device(loss, self._batched)  # no host sync inside the loop
        else:
            self.loss.append(to_numpy(loss, np.float64) if self._batched else np.float64(loss.item()))

    def update(self, iteration=100):
        if iteration > 1 and self._fast_loop_ok():
            # the first update through update_once(): it validates `algorithm` / `domain` and raises exactly what the
            # reference raises; the remaining ones are enqueued by the library (same entry points, same order)
            self.update_once()
            if self.recordable_loss:
                self._record_loss()
            eng, n = self._engine, iteration - 1
            loss = eng.empty((n, int(self._X.shape[0])), dtype=torch.float64) if self.recordable_loss else None
            eng.nmf_iterate(n, self._kind_code(), self._X, self._dev("T", False), self._dev("V", False),
                            domain=self.domain, eps=self.eps, param=self._kind_param(), loss=loss)
            self._touch("T", "V")
            if loss is not None:
                self.loss.append_device_block(loss, self._batched)
            return

        for idx in range(iteration):
            self.update_once()

            if self.recordable_loss:
                self._record_loss()

    def update_once(self):
        self._engine.nmf_update(self._kind_code(), self._X, self._dev("T", False), self._dev("V", False),
                                domain=self.domain, eps=self.eps, param=self._kind_param())
        self._touch("T", "V")


class EUCNMF(NMFbase):
    """reference: nmf.py:150-207"""
    _KIND = _lib.NMF_EUC

    def __init__(self, n_basis=2, domain=2, algorithm='mm', eps=EPS, *, dtype='float64', device=None, recordable_loss=True):
        """
        Args:
            n_basis: number of basis
        """
        super().__init__(n_basis=n_basis, eps=eps, dtype=dtype, device=device, recordable_loss=recordable_loss)

        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."
        assert algorithm == 'mm', "algorithm must be 'mm'."

        self.domain = domain
        self.algorithm = algorithm

    def update_once(self):
        if self.algorithm == 'mm':
            self.update_once_mm()
        else:
            raise ValueError("Not support {} based update.".format(self.algorithm))

    def update_once_mm(self):
        NMFbase.update_once(self)


class KLNMF(NMFbase):
    """reference: nmf.py:209-266"""
    _KIND = _lib.NMF_KL

    def __init__(self, n_basis=2, domain=2, algorithm='mm', eps=EPS, *, dtype='float64', device=None, recordable_loss=True):
        """
        Args:
            K: number of basis
        """
        super().__init__(n_basis=n_basis, eps=eps, dtype=dtype, device=device, recordable_loss=recordable_loss)

        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."
        assert algorithm == 'mm', "algorithm must be 'mm'."

        self.domain = domain
        self.algorithm = algorithm

    def update_once(self):
        if self.algorithm == 'mm':
            self.update_once_mm()
        else:
            raise ValueError("Not support {} based update.".format(self.algorithm))

    def update_once_mm(self):
        NMFbase.update_once(self)


class ISNMF(NMFbase):
    """reference: nmf.py:268-356"""
    _KIND = _lib.NMF_IS_MM

    def __init__(self, n_basis=2, domain=2, algorithm='mm', eps=EPS, *, dtype='float64', device=None, recordable_loss=True):
        """
        Args:
            K: number of basis
            algorithm: 'mm': MM algorithm based update
        """
        super().__init__(n_basis=n_basis, eps=eps, dtype=dtype, device=device, recordable_loss=recordable_loss)

        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."

        self.domain = domain
        self.algorithm = algorithm

    def _kind_code(self):
        return _lib.NMF_IS_ME if self.algorithm == 'me' else _lib.NMF_IS_MM

    def update_once(self):
        if self.algorithm == 'mm':
            self.update_once_mm()
        elif self.algorithm == 'me':
            self.update_once_me()
        else:
            raise ValueError("Not support {} based update.".format(self.algorithm))

    def update_once_mm(self):
        NMFbase.update_once(self)

    def update_once_me(self):
        assert self.domain == 2, "Only domain = 2 is supported."
        NMFbase.update_once(self)


class tNMF(NMFbase):
    """reference: nmf.py:358-429 (Student's t NMF, MM update; domain 2 only)"""
    _KIND = _lib.NMF_T

    def __init__(self, n_basis=2, nu=1e+3, domain=2, algorithm='mm', eps=EPS, *, dtype='float64', device=None,
                 recordable_loss=True):
        """
        Args:
            K: number of basis
            algorithm: 'mm': MM algorithm based update
        """
        super().__init__(n_basis=n_basis, eps=eps, dtype=dtype, device=device, recordable_loss=recordable_loss)

        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."

        self.nu = nu
        self.domain = domain
        self.algorithm = algorithm

    def _kind_param(self):
        return float(self.nu)

    def update_once(self):
        if self.algorithm == 'mm':
            self.update_once_mm()
        else:
            raise ValueError("Not support {} based update.".format(self.algorithm))

    def update_once_mm(self):
        assert self.domain == 2, "`domain` is expected 2."
        NMFbase.update_once(self)


class CauchyNMF(NMFbase):
    """reference: nmf.py:431-600 ('naive-multipricative', 'mm', 'me', 'mm_fast'; domain 2 only)"""
    _ALGORITHMS = {'naive-multipricative': _lib.NMF_CAUCHY_NAIVE, 'mm': _lib.NMF_CAUCHY_MM,
                   'me': _lib.NMF_CAUCHY_ME, 'mm_fast': _lib.NMF_CAUCHY_MM_FAST}

    def __init__(self, n_basis, domain=2, algorithm='naive-multipricative', eps=EPS, *, dtype='float64', device=None,
                 recordable_loss=True):
        super().__init__(n_basis=n_basis, eps=eps, dtype=dtype, device=device, recordable_loss=recordable_loss)

        assert domain == 2, "Only `domain` = 2 is supported."

        self.domain = domain
        self.algorithm = algorithm

    def _kind_code(self):
        if self.algorithm not in self._ALGORITHMS:
            raise ValueError("Not support {} based update.".format(self.algorithm))
        return self._ALGORITHMS[self.algorithm]

    def update_once(self):
        if self.algorithm == 'naive-multipricative':
            self.update_once_naive()
        elif self.algorithm == 'mm':
            self.update_once_mm()
        elif self.algorithm == 'me':
            self.update_once_me()
        elif self.algorithm == 'mm_fast':
            self.update_once_mm_fast()
        else:
            raise ValueError("Not support {} based update.".format(self.algorithm))

    def _update_once_checked(self):
        assert self.domain == 2, "Only 'domain' = 2 is supported."
        NMFbase.update_once(self)

    update_once_naive = update_once_mm = update_once_me = update_once_mm_fast = _update_once_checked
