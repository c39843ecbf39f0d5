"""AuxIVA (Laplace / Gauss, iterative projection) on MI355X -- drop-in for `bss.iva.AuxLaplaceIVA`
and `bss.iva.AuxGaussIVA` of the reference (/root/reference/src/bss/iva.py:22-128, 289-802).

One iteration = two passes over X on the device:
  pass A  r_n(t) from y = W x (iva.py:489-491 / 722-724) -- the same pass yields the data term of the
          negative log-likelihood for the current filters (iva.py:604-619 / 783-802), so recording the
          loss costs no extra pass;
  pass B  weighted covariance (iva.py:493-499) + IP sweep (iva.py:500-518).
`algorithm_spatial='ISS'` (iva.py:525-542, 758-775) shares both passes: the rank-1 updates are applied to W through
quadratic forms of the same covariances (Y = W X is linear in W), so `demix_filter` stays available in the loop.
`algorithm_spatial in {'IP2', 'pairwise'}` (iva.py:544-599) is on the HIP path for AuxLaplaceIVA; AuxGaussIVA raises
NotImplementedError for it exactly like the reference (iva.py:777-778).  There is no CPU fallback.
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from .._state import DeviceArray, DeviceState
from .._loss import LazyLossList
from .. import _lib
from ..ops import Engine

EPS = 1e-12
THRESHOLD = 1e+12

__algorithms_spatial__ = ['IP', 'IVA', 'ISS', 'IPA', 'pairwise', 'IP1', 'IP2']


class IVAbase(DeviceState):
    """reference: iva.py:22-128"""

    demix_filter = DeviceArray("W", complex_=True)

    def __init__(self, callbacks=None, recordable_loss=True, eps=EPS, *, dtype='float64', device=None):
        if callbacks is not None:
            if callable(callbacks):
                callbacks = [callbacks]
            self.callbacks = callbacks
        else:
            self.callbacks = None
        self.eps = eps

        self.input = None
        self.recordable_loss = recordable_loss
        if self.recordable_loss:
            self.loss = LazyLossList()  # a list; entries are materialised from HBM on first read
        else:
            self.loss = None

        self.dtype = dtype
        self.device = device
        self._engine = None
        self._estimation = None

    def _ensure_engine(self):
        if self._engine is None:
            self._engine = Engine(dtype=self.dtype, device=self.device)
        return self._engine

    def _reset(self, **kwargs):
        assert self.input is not None, "Specify data!"

        for key in kwargs.keys():
            setattr(self, key, kwargs[key])

        eng = self._ensure_engine()
        X = self.input
        ndim = X.dim() if isinstance(X, torch.Tensor) else np.ndim(X)
        if ndim not in (3, 4):
            raise ValueError("input must be (n_channels, n_bins, n_frames), got {} dims".format(ndim))
        self._batched = ndim == 4
        Xd = to_device(X, eng.prec.cplx, eng.dev)
        if not self._batched:
            Xd = Xd.unsqueeze(0)
        self._X = Xd.contiguous()
        B, n_channels, n_bins, n_frames = (int(s) for s in self._X.shape)
        self._status = eng.new_status(B)
        n_sources = n_channels  # n_channels == n_sources (iva.py:47)

        self.n_sources, self.n_channels = n_sources, n_channels
        self.n_bins, self.n_frames = n_bins, n_frames

        if not hasattr(self, 'demix_filter'):
            W = torch.eye(n_sources, n_channels, dtype=eng.prec.cplx, device=eng.dev)
            self._set_dev("W", W.repeat(B, n_bins, 1, 1).contiguous())
        self._estimation = None

    @property
    def _Wd(self):
        return self._dev("W", True)

    @property
    def estimation(self):
        if self._estimation is None:
            if getattr(self, "_X", None) is None:
                raise AttributeError("'{}' object has no attribute 'estimation'".format(type(self).__name__))
            Y = to_numpy(self._engine.demix(self._X, self._Wd), np.complex128)
            self._estimation = Y if self._batched else Y[0]
        return self._estimation

    @estimation.setter
    def estimation(self, value):
        self._estimation = value

    def __repr__(self):
        s = "IVA("
        s += ")"

        return s.format(**self.__dict__)

    def update_once(self):
        raise NotImplementedError("Implement 'update_once' function")

    def separate(self, input, demix_filter):
        """y = W x on the device (iva.py:105-117)."""
        eng = self._ensure_engine()
        X = to_device(input, eng.prec.cplx, eng.dev)
        W = to_device(demix_filter, eng.prec.cplx, eng.dev)
        batched = X.dim() == 4
        if not batched:
            X = X.unsqueeze(0)
        if W.dim() == 2:
            W = W.expand(X.shape[2], -1, -1)
        if W.dim() == 3:
            W = W.unsqueeze(0).expand(X.shape[0], -1, -1, -1)
        Y = eng.demix(X.contiguous(), W.contiguous())
        if isinstance(input, torch.Tensor):
            return Y if batched else Y[0]
        Y = to_numpy(Y, np.complex128)
        return Y if batched else Y[0]

    def compute_demix_filter(self, estimation, input):
        """W = Y X^H (X X^H)^{-1} per bin (iva.py:119-125): the least-squares demixing filter that maps `input`
        onto `estimation`.  (n_sources, n_bins, n_frames) x (n_channels, n_bins, n_frames) -> (n_bins, n_sources,
        n_channels); NumPy in -> NumPy out, device tensors in -> device tensor out; a leading utterance axis is kept."""
        eng = self._ensure_engine()
        Y = to_device(estimation, eng.prec.cplx, eng.dev)
        X = to_device(input, eng.prec.cplx, eng.dev)
        if Y.shape != X.shape:
            raise ValueError("estimation {} and input {} must have the same shape (n_sources == n_channels)".format(tuple(Y.shape), tuple(X.shape)))
        batched = X.dim() == 4
        if not batched:
            X, Y = X.unsqueeze(0), Y.unsqueeze(0)
        status = eng.new_status(X.shape[0])
        W = eng.compute_demix_filter(Y.contiguous(), X.contiguous(), status=status)
        if int(status.max().item()) & _lib.STATUS_SINGULAR:
            raise np.linalg.LinAlgError("Singular matrix")
        if isinstance(input, torch.Tensor) and isinstance(estimation, torch.Tensor):
            return W if batched else W[0]
        W = to_numpy(W, np.complex128)
        return W if batched else W[0]

    def compute_negative_loglikelihood(self):
        raise NotImplementedError("Implement 'compute_negative_loglikelihood' function.")

    def _check_status(self):
        flags = int(self._status.max().item())
        if flags & _lib.STATUS_SINGULAR:
            self._status.zero_()
            raise np.linalg.LinAlgError("Singular matrix")


class AuxIVAbase(IVAbase):
    """reference: iva.py:289-386"""

    _KIND = None
    _NAME = "AuxIVA"

    def __init__(self, algorithm_spatial='IP', reference_id=0, callbacks=None, apply_projection_back=True,
                 recordable_loss=True, eps=EPS, threshold=THRESHOLD, *, dtype='float64', device=None):
        super().__init__(callbacks=callbacks, recordable_loss=recordable_loss, eps=eps, dtype=dtype, device=device)

        self.algorithm_spatial = algorithm_spatial
        self.reference_id = reference_id
        self.apply_projection_back = apply_projection_back
        self.threshold = threshold

        if not self.algorithm_spatial in __algorithms_spatial__:
            raise ValueError("Not support {} based spatial updates.".format(self.algorithm_spatial))

        if self.algorithm_spatial in ['pairwise', 'IP2']:
            self.update_pair = None

    def _require_supported(self):
        if self._KIND is None:
            raise NotImplementedError("Implement 'update_once' function.")
        if self.algorithm_spatial in ('pairwise', 'IP2') and self._KIND == _lib.IVA_GAUSS:
            raise NotImplementedError("In progress...")  # iva.py:777-778
        if self.algorithm_spatial not in ('IP', 'IP1', 'ISS', 'pairwise', 'IP2'):
            raise NotImplementedError("algorithm_spatial='{}' is not on the HIP path (no CPU fallback is provided).".format(self.algorithm_spatial))

    def _reset(self, **kwargs):
        super()._reset(**kwargs)
        self._require_supported()
        self._r = None       # r_n(t) of the CURRENT filters; None = stale
        self._r_src = None
        self._loss_dev = None

    def _refresh_weights(self, with_loss):
        """Pass A: r (and the loss of the current filters when asked) in one sweep over X."""
        self._r, self._loss_dev = self._engine.auxiva_weights(self._X, self._Wd, self._KIND, eps=self.eps,
                                                             with_loss=with_loss)
        self._r_src = self._Wd  # the filters these weights belong to (a host-side assignment replaces the tensor)

    def __call__(self, input, iteration=100, **kwargs):
        """
        Args:
            input (n_channels, n_bins, n_frames)
        Returns:
            output (n_channels, n_bins, n_frames)
        """
        self.input = input

        self._reset(**kwargs)

        if iteration > 0 and self._fast_loop_ok():
            # nothing observes the model between iterations: the whole loop is ONE call into the library
            # (assx_auxiva_iterate enqueues the same entry points in the same order: bit-identical to the loop below)
            self._run_fast_loop(iteration)
        else:
            if self.recordable_loss:
                self._record_loss()

            if self.callbacks is not None:
                for callback in self.callbacks:
                    callback(self)

            for idx in range(iteration):
                if self.algorithm_spatial in ['pairwise', 'IP2']:
                    self._select_update_pair()

                self.update_once()

                if self.recordable_loss:
                    self._record_loss()

                if self.callbacks is not None:
                    for callback in self.callbacks:
                        callback(self)

        eng = self._engine
        scale = None
        if self.apply_projection_back:
            scale = eng.projection_back_scale(self._X, self._Wd, self.reference_id, self._status)
        Y = eng.demix(self._X, self._Wd, scale=scale)
        self._check_status()

        if isinstance(input, torch.Tensor):
            output = Y if self._batched else Y[0]
        else:
            output = to_numpy(Y, np.complex128)
            output = output if self._batched else output[0]
        self.estimation = output

        return output

    # ---- the loop of __call__ as one library call (iva.py:420-441) ---------------------------------------------
    _OWN_STEPS = ("update_once", "update_once_ip", "update_once_iss", "update_once_pairwise", "_spatial_update",
                  "_record_loss", "_refresh_weights", "_select_update_pair")

    def _fast_loop_ok(self):
        if self.callbacks is not None or self._KIND is None:
            return False
        if any(getattr(type(self), name) is not getattr(AuxIVAbase, name) for name in self._OWN_STEPS):
            return False
        if self.recordable_loss and not isinstance(self.loss, LazyLossList):
            return False
        return self.algorithm_spatial in ('IP', 'IP1', 'ISS', 'pairwise', 'IP2')

    def _run_fast_loop(self, iteration):
        eng = self._engine
        B, N, T = int(self._X.shape[0]), self.n_sources, self.n_frames
        spatial, pair = _lib.SPATIAL_IP, (0, 1)
        if self.algorithm_spatial == 'ISS':
            spatial = _lib.SPATIAL_ISS
        elif self.algorithm_spatial in ['pairwise', 'IP2']:
            self._require_supported()
            self._select_update_pair()  # the pair of the first iteration; the library advances it like iva.py:370-382
            spatial, pair = _lib.SPATIAL_IP2, self.update_pair
        r = eng.empty((B, N, T))
        loss = eng.empty((iteration + 1, B), dtype=torch.float64) if self.recordable_loss else None
        eng.auxiva_iterate(iteration, self._KIND, self._X, self._Wd, r, eps=self.eps, threshold=self.threshold,
                           status=self._status, loss=loss, spatial=spatial, pair=pair)
        if spatial == _lib.SPATIAL_IP2:
            self.update_pair = ((pair[0] + iteration - 1) % N, (pair[1] + iteration - 1) % N)
        self._touch("W")
        self._estimation = None
        if loss is not None:  # r and the last loss entry belong to the final filters
            self._r, self._r_src, self._loss_dev = r, self._Wd, loss[iteration]
            self.loss.append_device_block(loss, self._batched)
        else:
            self._r = self._loss_dev = None

    def __repr__(self):
        s = self._NAME + "("
        s += "algorithm_spatial={algorithm_spatial}"
        s += ")"

        return s.format(**self.__dict__)

    def update_once(self):
        if self.algorithm_spatial in ['IP', 'IP1']:
            self.update_once_ip()
        elif self.algorithm_spatial == 'ISS':
            self.update_once_iss()
        elif self.algorithm_spatial in ['pairwise', 'IP2']:
            self.update_once_pairwise()
        else:
            self._require_supported()

    def update_once_ip(self):
        """iva.py:481-523 / 714-757: weights from the current estimate, covariance, IP sweep."""
        self._spatial_update(_lib.SPATIAL_IP)

    def update_once_iss(self):
        """iva.py:525-542 / 758-775: weights from the current estimate, covariance, ISS sweep."""
        self._spatial_update(_lib.SPATIAL_ISS)

    def update_once_pairwise(self):
        """iva.py:544-599: pairwise update of rows `update_pair`."""
        self._require_supported()
        self._spatial_update(_lib.SPATIAL_IP2, pair=self.update_pair)

    def _select_update_pair(self):
        """(0,1), (1,2), ..., (N-1,0)   (iva.py:370-382)."""
        n_sources = self.n_sources

        if self.update_pair is None:
            m, n = 0, 1
        else:
            m, n = self.update_pair
            m, n = m + 1, n + 1
            m, n = m % n_sources, n % n_sources

        self.update_pair = m, n

    def _spatial_update(self, spatial, pair=(0, 1)):
        if self._r is None or self._r_src is not self._Wd:
            self._refresh_weights(with_loss=False)
        self._engine.auxiva_spatial_update(self._X, self._Wd, self._r, eps=self.eps, threshold=self.threshold,
                                           status=self._status, spatial=spatial, pair=pair)
        self._touch("W")
        self._estimation = None
        self._r = None
        self._loss_dev = None

    def _record_loss(self):
        """Append the current loss without a host sync (the value stays in HBM until `loss` is read)."""
        if self._r is None or self._loss_dev is None or self._r_src is not self._Wd:
            self._refresh_weights(with_loss=True)
        if isinstance(self.loss, LazyLossList):
            self.loss.append_device(self._loss_dev, self._batched)
        else:
            self.loss.append(to_numpy(self._loss_dev, np.float64) if self._batched else np.float64(self._loss_dev.item()))

    def compute_negative_loglikelihood(self):
        """iva.py:604-619 / 783-802.  Rides on the r_n(t) pass; syncs to return a Python float."""
        if self._r is None or self._loss_dev is None or self._r_src is not self._Wd:
            self._refresh_weights(with_loss=True)
        self._check_status()
        if self._batched:
            return to_numpy(self._loss_dev, np.float64)
        return np.float64(self._loss_dev.item())


class AuxLaplaceIVA(AuxIVAbase):
    """reference: iva.py:388-619"""
    _KIND = _lib.IVA_LAPLACE
    _NAME = "AuxLaplaceIVA"


class AuxGaussIVA(AuxIVAbase):
    """reference: iva.py:621-802"""
    _KIND = _lib.IVA_GAUSS
    _NAME = "AuxGaussIVA"
