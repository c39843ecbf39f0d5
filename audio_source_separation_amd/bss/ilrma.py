"""Gauss-ILRMA on MI355X -- drop-in for the reference's `bss.ilrma.GaussILRMA` (IP spatial update).

Same constructor, `__call__(input, iteration, **kwargs)`, attributes and callback protocol as
/root/reference/src/bss/ilrma.py:22-677; the per-iteration `update_once()` runs as hand-written
HIP kernels behind the C-ABI in include/assx.h (no CPU fallback).  Extra keyword-only arguments
(`dtype`, `device`, `power_statistic`) default to the reference's behaviour.

`algorithm_spatial='ISS'` is on the HIP path too: the rank-1 updates are applied to the demixing filters (Y = W X is
linear in W, so the reference's statistics on Y are quadratic forms of the same weighted covariances as IP); unlike
the reference, `demix_filter` therefore stays available during the loop instead of being None.

`algorithm_spatial in {'IP2', 'pairwise'}` (ilrma.py:432-481, 566-646) is on the HIP path as well.

`partitioning=True` (shared bases + latent variables, ilrma.py:79-95, 368-408, 313-320; domain 2 only as in the
reference) is on the HIP path: the streaming kernels run on the per-source expansion Z[n,k] T[f,k] of the shared model
and small kernels fold their partial sums into the Z / T / V updates (csrc/assx_partition.hpp).
"""
import warnings

import numpy as np

from .._device import to_device, to_numpy, torch
from .._state import DeviceArray, DeviceState
from .._loss import LazyLossList
from .. import _lib
from ..ops import Engine

EPS = 1e-12
THRESHOLD = 1e+12

__algorithms_spatial__ = ['IP', 'IVA', 'ISS', 'IPA', 'pairwise', 'IP1', 'IP2']


class ILRMAbase(DeviceState):
    """Independent Low-rank Matrix Analysis (reference: ilrma.py:22-176)."""

    demix_filter = DeviceArray("W", complex_=True)
    basis = DeviceArray("T", complex_=False)
    activation = DeviceArray("V", complex_=False)
    latent = DeviceArray("Z", complex_=False)

    def __init__(self, n_basis=10, partitioning=False, normalize=True, algorithm_spatial='IP', callbacks=None,
                 recordable_loss=True, eps=EPS, *, dtype='float64', device=None):
        if callbacks is not None:
            if callable(callbacks):
                callbacks = [callbacks]
            self.callbacks = callbacks
        else:
            self.callbacks = None
        self.eps = eps

        self.n_basis = n_basis
        self.partitioning = partitioning
        self.normalize = normalize

        assert algorithm_spatial in __algorithms_spatial__, "Choose from {} as `algorithm_spatial`.".format(__algorithms_spatial__)
        assert algorithm_spatial in ['IP', 'ISS', 'pairwise', 'IP1', 'IP2'], "Not support {}-based demixing filter updates.".format(algorithm_spatial)
        self.algorithm_spatial = algorithm_spatial

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
        self._deferred_loss = None  # device scalar(s) appended to `loss` whose value the next basis pass will write

    # ------------------------------------------------------------------ device plumbing
    def _require_supported(self):
        if self.algorithm_spatial not in ('IP', 'IP1', 'ISS', 'IP2', 'pairwise'):
            raise NotImplementedError("algorithm_spatial='{}' is not on the HIP path (no CPU fallback is provided).".format(self.algorithm_spatial))

    def _ensure_engine(self):
        if self._engine is None:
            self._engine = Engine(dtype=self.dtype, device=self.device)
        return self._engine

    def _upload_input(self):
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
        self._status = eng.new_status(self._X.shape[0])

    def _reset(self, **kwargs):
        assert self.input is not None, "Specify data!"

        for key in kwargs.keys():
            setattr(self, key, kwargs[key])

        self._require_supported()
        self._upload_input()
        eng = self._engine
        n_basis = self.n_basis

        B, n_channels, n_bins, n_frames = (int(s) for s in self._X.shape)
        n_sources = n_channels  # n_channels == n_sources (ilrma.py:61-62)

        self.n_sources, self.n_channels = n_sources, n_channels
        self.n_bins, self.n_frames = n_bins, n_frames

        if not hasattr(self, 'demix_filter'):
            W = torch.eye(n_sources, n_channels, dtype=eng.prec.cplx, device=eng.dev)
            self._set_dev("W", W.repeat(B, n_bins, 1, 1).contiguous())
        # else: the existing filter (previous call or user-supplied) is the warm start (ilrma.py:70-72)

        lead = (B,) if self._batched else ()
        if self.partitioning:
            # latent, then basis, then activation from the global NumPy RNG (ilrma.py:78-95)
            if not hasattr(self, 'latent'):
                variance_latent = 1e-2
                Z = np.random.rand(*(lead + (n_sources, n_basis))) * variance_latent + 1 / n_sources
                Zsum = Z.sum(axis=-2, keepdims=True)
                Zsum[Zsum < self.eps] = self.eps
                self.latent = Z / Zsum
            shape_T, shape_V = lead + (n_bins, n_basis), lead + (n_basis, n_frames)
        else:
            shape_T, shape_V = lead + (n_sources, n_bins, n_basis), lead + (n_sources, n_basis, n_frames)
        # global NumPy RNG, basis first then activation, exactly as ilrma.py:97-104
        if not hasattr(self, 'basis'):
            self.basis = np.random.rand(*shape_T)
        if not hasattr(self, 'activation'):
            self.activation = np.random.rand(*shape_V)

        self._estimation = None  # = separate(X, W), formed on demand

    # device tensors of the public arrays (uploaded on first use after a host-side assignment)
    @property
    def _Wd(self):
        return self._dev("W", True)

    @property
    def _Td(self):
        return self._dev("T", False)

    @property
    def _Vd(self):
        return self._dev("V", False)

    @property
    def _Zd(self):
        return self._dev("Z", False)

    def _model(self):
        """(basis, activation) device tensors in the per-source layout (B,N,F,K) / (B,N,K,T) every kernel takes.
        With a partitioning function this is the expansion Z[n,k] T[f,k] / V[k,t] of the shared model, rebuilt on
        every use (a few hundred KB) so host-side assignments to latent / basis / activation are always honoured."""
        if not self.partitioning:
            return self._Td, self._Vd
        eng = self._engine
        B, N, F, T, K = self._X.shape[0], self.n_sources, self.n_bins, self.n_frames, int(self._Td.shape[-1])
        if getattr(self, "_Teff", None) is None or tuple(self._Teff.shape) != (B, N, F, K) \
                or tuple(self._Veff.shape) != (B, N, K, T):
            self._Teff, self._Veff = eng.empty((B, N, F, K)), eng.empty((B, N, K, T))
        eng.ilrma_expand_partitioned(self._Zd, self._Td, self._Vd, self._Teff, self._Veff)
        return self._Teff, self._Veff

    # ------------------------------------------------------------------ reference API
    @property
    def estimation(self):
        """(n_sources, n_bins, n_frames): current y = W x (ilrma.py:76), or the final scaled output."""
        if self._estimation is None:
            if getattr(self, "_X", None) is None:
                raise AttributeError("'{}' object has no attribute 'estimation'".format(type(self).__name__))
            Y = self._engine.demix(self._X, self._dev("W", True))
            Y = to_numpy(Y, np.complex128)
            self._estimation = Y if self._batched else Y[0]
        return self._estimation

    @estimation.setter
    def estimation(self, value):
        self._estimation = value

    def __call__(self, input, iteration=100, **kwargs):
        raise NotImplementedError("Implement '__call__' in the subclass")

    def __repr__(self):
        s = "ILRMA("
        s += "n_basis={n_basis}"
        s += ", partitioning={partitioning}"
        s += ", normalize={normalize}"
        s += ")"

        return s.format(**self.__dict__)

    def update_once(self):
        raise NotImplementedError("Implement 'update_once' function")

    def separate(self, input, demix_filter):
        """y = W x on the device (ilrma.py:153-165).  Accepts / returns NumPy arrays like the reference."""
        eng = self._ensure_engine()
        X = to_device(input, eng.prec.cplx, eng.dev)
        W = to_device(demix_filter, eng.prec.cplx, eng.dev)
        batched = X.dim() == 4
        if not batched:
            X = X.unsqueeze(0)
        if W.dim() == 2:  # (N, M) broadcast over bins, as `demix_filter @ input` does in the reference
            W = W.expand(X.shape[2], -1, -1)
        if W.dim() == 3:
            W = W.unsqueeze(0).expand(X.shape[0], -1, -1, -1)
        Y = eng.demix(X.contiguous(), W.contiguous())
        if isinstance(input, torch.Tensor):
            return Y if batched else Y[0]
        Y = to_numpy(Y, np.complex128)
        return Y if batched else Y[0]

    def compute_demix_filter(self, estimation, input):
        """W = Y X^H (X X^H)^{-1} per bin (ilrma.py:167-173): the least-squares demixing filter that maps `input`
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
        """Turn device-side flags into the exceptions NumPy would have raised (one sync)."""
        flags = int(self._status.max().item())
        if flags & _lib.STATUS_SINGULAR:
            self._status.zero_()
            raise np.linalg.LinAlgError("Singular matrix")

    def _run_callbacks(self):
        if self.callbacks is not None:
            for callback in self.callbacks:
                callback(self)


class GaussILRMA(ILRMAbase):
    """
    Reference: "Determined Blind Source Separation Unifying Independent Vector Analysis and Nonnegative Matrix Factorization"
    See https://ieeexplore.ieee.org/document/7486081
    (reference implementation: ilrma.py:178-677)
    """

    def __init__(self, n_basis=10, domain=2, partitioning=False, normalize='power', algorithm_spatial='IP',
                 reference_id=0, callbacks=None, recordable_loss=True, eps=EPS, threshold=THRESHOLD, *,
                 dtype='float64', device=None, power_statistic='covariance'):
        """
        Args:
            normalize <str>: 'power': power based normalization, or 'projection-back': projection back based normalization.
            threshold <float>: threshold for condition number when computing (WU)^{-1}.
            dtype: 'float64' (complex128 kernels = the reference's precision) or 'float32' (throughput mode).
            power_statistic: 'covariance' evaluates mean|y_n|^2 = mean_f w_n^H C_f w_n from the plain covariance
                C_f computed once per call (no pass over X); 'direct' re-reads X every iteration like ilrma.py:298-306.
        """
        super().__init__(n_basis=n_basis, partitioning=partitioning, normalize=normalize,
                         algorithm_spatial=algorithm_spatial, callbacks=callbacks, recordable_loss=recordable_loss,
                         eps=eps, dtype=dtype, device=device)

        assert 1 <= domain <= 2, "1 <= `domain` <= 2 is not satisfied."
        assert power_statistic in ('covariance', 'direct')

        self.domain = domain
        self.reference_id = reference_id
        self.threshold = threshold
        self.power_statistic = power_statistic

        if self.partitioning:
            assert domain == 2, "Not support domain = {}".format(domain)  # ilrma.py:369, 490

        if self.algorithm_spatial == 'ISS':
            warnings.warn("in progress", UserWarning)  # as the reference does (ilrma.py:197-198)

        if self.algorithm_spatial in ['pairwise', 'IP2']:
            self.update_pair = None

    def __call__(self, input, iteration=100, **kwargs):
        """
        Args:
            input (n_channels, n_bins, n_frames)
        Returns:
            output (n_channels, n_bins, n_frames)
        """
        self.input = input

        self._reset(**kwargs)

        plan = self._fast_loop_plan() if iteration > 0 else None
        if plan is not None:
            # no callback has to see the model between iterations: the whole loop is ONE call into the library
            # (assx_ilrma_iterate enqueues the same entry points in the same order: bit-identical to the loop below)
            self._run_fast_loop(iteration, plan)
        else:
            if self.recordable_loss:
                self._record_loss()

            self._run_callbacks()

            for idx in range(iteration):
                if self.algorithm_spatial in ['pairwise', 'IP2']:
                    self._select_update_pair()

                self.update_once()

                if self.recordable_loss:
                    self._record_loss()

                self._run_callbacks()

            self._resolve_deferred_loss()  # the loss of the last iteration has no next pass to ride on

        # final projection back (ilrma.py:258-273); scale and y = W x in two small passes over X
        eng = self._engine
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

    # ---- the loop of __call__ as one library call (ilrma.py:233-256) ------------------------------------------
    _OWN_STEPS = ("update_once", "update_source_model", "update_spatial_model", "_record_loss", "_select_update_pair")

    def _fast_loop_plan(self):
        """dict(normalize=, pb_exponent=) when assx_ilrma_iterate can stand in for the Python loop, else None: nothing
        observes the model between iterations (no callbacks), the steps are this class's own (a subclass that
        overrides one keeps the loop), no partitioning function, and a normalisation the entry point knows."""
        if self.callbacks is not None or self.partitioning:
            return None
        if any(getattr(type(self), name) is not getattr(GaussILRMA, name) for name in self._OWN_STEPS):
            return None
        if self.recordable_loss and not isinstance(self.loss, LazyLossList):
            return None
        if not self.normalize:
            return dict(normalize=0, pb_exponent=2.0)
        if self.normalize == 'power' and self.power_statistic == 'covariance':
            return dict(normalize=1, pb_exponent=2.0)
        if self.normalize == 'projection-back':
            return dict(normalize=2, pb_exponent=float(getattr(self, "_pb_basis_exponent", None) or self.domain))
        return None  # 'direct' power statistic, unknown names (the loop raises what the reference raises)

    def _ensure_plain_covariance(self):
        """C_f = mean_t x x^H: constant over the iterations, one pass per call; with it the IP kernel emits the per-bin
        power statistic of the updated filters."""
        if self._C is None:
            eng = self._engine
            B, M, F, _ = self._X.shape
            self._C = eng.cov_accumulate(self._X).reshape(B, F, M, M)
            self._pbins = eng.empty((B, M, F), dtype=torch.float64)
        return self._C, self._pbins

    def _run_fast_loop(self, iteration, plan):
        eng = self._engine
        B, N = int(self._X.shape[0]), self.n_sources
        spatial, pair = _lib.SPATIAL_IP, (0, 1)
        if self.algorithm_spatial == 'ISS':
            spatial = _lib.SPATIAL_ISS
        elif self.algorithm_spatial in ['pairwise', 'IP2']:
            self._select_update_pair()  # the pair of the first iteration; the library advances it like ilrma.py:635-646
            spatial, pair = _lib.SPATIAL_IP2, self.update_pair
        C = pbins = scale = None
        if plan["normalize"] == 1:
            C, pbins = self._ensure_plain_covariance()
        elif plan["normalize"] == 2:
            scale = eng.empty((B, N, self.n_bins), complex_=True)
        loss = eng.empty((iteration + 1, B), dtype=torch.float64) if self.recordable_loss else None
        eng.ilrma_iterate(iteration, self._X, self._Wd, self._Td, self._Vd, domain=self.domain, eps=self.eps,
                          threshold=self.threshold, status=self._status, loss=loss, spatial=spatial, pair=pair,
                          normalize=plan["normalize"], C=C, power_bins=pbins, scale=scale, ref=self.reference_id,
                          pb_exponent=plan["pb_exponent"])
        if spatial == _lib.SPATIAL_IP2:
            self.update_pair = ((pair[0] + iteration - 1) % N, (pair[1] + iteration - 1) % N)
        self._touch("W", "T", "V")
        self._estimation = None
        if loss is not None:
            self.loss.append_device_block(loss, self._batched)

    def _reset(self, **kwargs):
        self._resolve_deferred_loss()
        super()._reset(**kwargs)
        if self.partitioning:
            assert self.domain == 2, "Not support domain = {}".format(self.domain)
        self._Teff = self._Veff = None
        self._C = None
        self._pbins = None
        self._power = self._engine.empty((self._X.shape[0], self.n_sources))

    def __repr__(self):
        s = "Gauss-ILRMA("
        s += "n_basis={n_basis}"
        s += ", domain={domain}"
        s += ", partitioning={partitioning}"
        s += ", normalize={normalize}"
        s += ", algorithm_spatial={algorithm_spatial}"
        s += ")"

        return s.format(**self.__dict__)

    def update_once(self):
        domain = self.domain
        eps = self.eps
        eng = self._engine

        self.update_source_model()
        self.update_spatial_model()

        if self.normalize:
            if self.normalize == 'power':
                if self.partitioning:
                    # Z / a^2 renormalised over sources, T takes the column sums (ilrma.py:313-320)
                    eng.ilrma_normalize_power_bins_partitioned(self._Wd, self._Zd, self._Td, self._power_bins(),
                                                               self.n_frames, eps=eps)
                    self._touch("Z")
                elif self.power_statistic == 'covariance':
                    # the IP kernel already emitted w_n^H C_f w_n per bin (update_spatial_model)
                    eng.ilrma_normalize_power_bins(self._Wd, self._Td, self._pbins, domain=domain, eps=eps)
                else:
                    eng.demix_power(self._X, self._Wd, out=self._power)
                    eng.ilrma_normalize_power(self._Wd, self._Td, self._power, domain=domain, eps=eps)
            elif self.normalize == 'projection-back':
                if self.partitioning:
                    raise NotImplementedError("Not support 'projection-back' based normalization for partitioninig function. Choose 'power' based normalization.")
                scale = eng.projection_back_scale(self._X, self._Wd, self.reference_id, self._status)
                eng.ilrma_normalize_pb(self._Wd, self._Td, scale,
                                       domain=getattr(self, "_pb_basis_exponent", None) or domain)
            else:
                raise ValueError("Not support normalization based on {}. Choose 'power' or 'projection-back'".format(self.normalize))
            self._touch("W", "T")
            self._estimation = None

    def update_source_model(self):
        """IS-NMF (mm) update of basis then activation on P = |W x|^2 (ilrma.py:356-366, 409-430); with IP2 only the
        selected pair's source models move (ilrma.py:432-481)."""
        sources = None
        if self.algorithm_spatial in ['pairwise', 'IP2']:
            if self.partitioning:
                raise NotImplementedError("Not support partitioning function.")
            sources = self.update_pair
        if self.partitioning:
            Teff, Veff = self._model()
            self._engine.ilrma_source_update_partitioned(self._X, self._Wd, self._Zd, self._Td, self._Vd, Teff, Veff,
                                                         eps=self.eps)
            self._touch("Z", "T", "V")
            return
        loss_prev, self._deferred_loss = self._deferred_loss, None
        self._engine.ilrma_source_update(self._X, self._Wd, self._Td, self._Vd, domain=self.domain, eps=self.eps,
                                         sources=sources, loss_prev=loss_prev)
        self._touch("T", "V")

    def _power_bins(self):
        """Per-bin power statistic w_n^H C_f w_n of the current filters (emitted by the spatial update)."""
        return self._pbins

    def _select_update_pair(self):
        """(0,1), (1,2), ..., (N-1,0)   (ilrma.py:635-646)."""
        n_sources = self.n_sources

        if self.update_pair is None:
            m, n = 0, 1
        else:
            m, n = self.update_pair
            m, n = m + 1, n + 1
            m, n = m % n_sources, n % n_sources

        self.update_pair = m, n

    def update_spatial_model(self):
        """Weighted covariance + iterative projection (ilrma.py:483-535) or ISS sweep (ilrma.py:537-564)."""
        eng = self._engine
        C = pbins = None
        if self.normalize == 'power' and (self.power_statistic == 'covariance' or self.partitioning):
            C, pbins = self._ensure_plain_covariance()
        spatial, pair = _lib.SPATIAL_IP, (0, 1)
        if self.algorithm_spatial == 'ISS':
            spatial = _lib.SPATIAL_ISS
        elif self.algorithm_spatial in ['pairwise', 'IP2']:
            spatial, pair = _lib.SPATIAL_IP2, self.update_pair
        Tb, V = self._model()
        eng.ilrma_spatial_update(self._X, self._Wd, Tb, V, domain=self.domain, eps=self.eps,
                                 threshold=self.threshold, status=self._status, C=C, power_bins=pbins, spatial=spatial,
                                 pair=pair)
        self._touch("W")
        self._estimation = None

    def _resolve_deferred_loss(self):
        """Compute a loss value that was left for the next basis pass, now, with the stand-alone kernel."""
        buf, self._deferred_loss = self._deferred_loss, None
        if buf is not None:
            self._engine.ilrma_loss(self._X, self._Wd, self._Td, self._Vd, domain=self.domain, eps=self.eps, out=buf)

    def _record_loss(self):
        """Append the current loss without a host sync (the value stays in HBM until `loss` is read).

        Without a partitioning function and callbacks the value is not even computed here: the basis pass of the next
        iteration forms the same y = W x and T V and accumulates the loss on the way (`loss_prev` of
        assx_ilrma_source_update); whoever reads `loss`, replaces a model array or ends the call first triggers the
        stand-alone kernel instead."""
        if isinstance(self.loss, LazyLossList) and not self.partitioning and self.callbacks is None:
            self._resolve_deferred_loss()
            self._deferred_loss = self._engine.empty((self._X.shape[0],), dtype=torch.float64)
            self.loss.before_flush = self._resolve_deferred_loss
            self.loss.append_device(self._deferred_loss, self._batched)
            return
        Tb, V = self._model()
        loss = self._engine.ilrma_loss(self._X, self._Wd, Tb, V, domain=self.domain, eps=self.eps)
        if isinstance(self.loss, LazyLossList):
            self.loss.append_device(loss, self._batched)
        else:  # a user replaced `loss` by a plain list
            self.loss.append(to_numpy(loss, np.float64) if self._batched else np.float64(loss.item()))

    def compute_negative_loglikelihood(self):
        """sum(P/R + log R) - 2 T sum_f log|det W_f| (ilrma.py:648-677).  Syncs to return a Python float."""
        Tb, V = self._model()
        loss = self._engine.ilrma_loss(self._X, self._Wd, Tb, V, domain=self.domain, eps=self.eps)
        self._check_status()
        if self._batched:
            return to_numpy(loss, np.float64)
        return np.float64(loss.item())


class ConsistentGaussILRMA(GaussILRMA):
    """
    Reference: "Consistent independent low-rank matrix analysis for determined blind source separation"
    See https://asp-eurasipjournals.springeropen.com/articles/10.1186/s13634-020-00704-4
    (reference implementation: ilrma.py:1089-1233)

    The reference supports IP only, and with IP its `update_once` recomputes the estimate from the demixing filters
    (ilrma.py:356-366), so the istft -> stft projection of `estimation` it performs first (ilrma.py:1206-1207) never
    reaches the model: what remains is Gauss-ILRMA with the projection-back rescaling of W and T after every
    iteration (ilrma.py:1219-1229).  That is what runs here; the projection itself is not recomputed.
    """

    def __init__(self, n_basis=10, partitioning=False, algorithm_spatial='IP', reference_id=0, fft_size=None,
                 hop_size=None, callbacks=None, recordable_loss=True, eps=EPS, threshold=THRESHOLD, *,
                 dtype='float64', device=None):
        super().__init__(n_basis=n_basis, partitioning=partitioning, normalize=False,
                         algorithm_spatial=algorithm_spatial, reference_id=reference_id, callbacks=callbacks,
                         recordable_loss=recordable_loss, eps=eps, threshold=threshold, dtype=dtype, device=device)

        if fft_size is None:
            raise ValueError("Specify `fft_size`.")

        if hop_size is None:
            hop_size = fft_size // 2

        self.fft_size, self.hop_size = fft_size, hop_size

        assert self.algorithm_spatial == 'IP', "Supports only IP-based spatial update."

    def __repr__(self):
        s = "Consistent-GaussILRMA("
        s += "n_basis={n_basis}"
        s += ", domain={domain}"
        s += ", partitioning={partitioning}"
        s += ", normalize={normalize}"
        s += ", algorithm_spatial={algorithm_spatial}"
        s += ")"

        return s.format(**self.__dict__)

    def _fast_loop_plan(self):
        if type(self) is not ConsistentGaussILRMA or self.callbacks is not None or self.partitioning:
            return None
        if self.n_bins != self.fft_size // 2 + 1 or self.normalize:
            return None  # the loop raises / does whatever update_once does
        if self.recordable_loss and not isinstance(self.loss, LazyLossList):
            return None
        return dict(normalize=2, pb_exponent=2.0)  # update_once below, every iteration

    def update_once(self):
        if self.n_bins != self.fft_size // 2 + 1:
            raise ValueError("n_bins = {} does not match fft_size = {}.".format(self.n_bins, self.fft_size))
        if self.partitioning:
            raise NotImplementedError("Not support 'projection-back' based normalization for partitioninig function. Choose 'power' based normalization.")
        # ilrma.py:1219-1229 == the 'projection-back' block of GaussILRMA.update_once, except that the reference
        # rescales the basis by |scale|**2 whatever `domain` a caller set through kwargs (ilrma.py:1229)
        self.normalize, self._pb_basis_exponent = 'projection-back', 2
        try:
            GaussILRMA.update_once(self)
        finally:
            self.normalize, self._pb_basis_exponent = False, None


class tILRMA(ILRMAbase):
    """
    Reference: "Independent low-rank matrix analysis based on complex student's t-distribution for blind audio source separation"
    See: https://ieeexplore.ieee.org/document/8168129
    (reference implementation: ilrma.py:713-1020; IP spatial update, domain 2, no partitioning function)
    """

    def __init__(self, n_basis=10, nu=1, domain=2, partitioning=False, normalize='power', algorithm_spatial='IP',
                 reference_id=0, callbacks=None, recordable_loss=True, eps=EPS, *, dtype='float64', device=None):
        """
        Args:
            nu: degree of freedom. nu = 1: Cauchy distribution, nu -> infty: Gaussian distribution.
            normalize <str>: 'power': power based normalization.
        """
        super().__init__(n_basis=n_basis, partitioning=partitioning, normalize=normalize,
                         algorithm_spatial=algorithm_spatial, callbacks=callbacks, recordable_loss=recordable_loss,
                         eps=eps, dtype=dtype, device=device)

        self.nu = nu
        self.domain = domain
        self.reference_id = reference_id

        assert self.algorithm_spatial == 'IP', "Supports only IP-based spatial update."

    def __call__(self, input, iteration=100, **kwargs):
        """
        Args:
            input (n_channels, n_bins, n_frames)
        Returns:
            output (n_channels, n_bins, n_frames)
        """
        self.input = input

        self._reset(**kwargs)

        if self.recordable_loss:
            self._record_loss()

        self._run_callbacks()

        for idx in range(iteration):
            self.update_once()

            if self.recordable_loss:
                self._record_loss()

            self._run_callbacks()

        eng = self._engine
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

    def _reset(self, **kwargs):
        super()._reset(**kwargs)
        self._C = None
        self._pbins = None
        self._Xi = None

    def __repr__(self):
        s = "t-ILRMA("
        s += "n_basis={n_basis}"
        s += ", nu={nu}"
        s += ", domain={domain}"
        s += ", partitioning={partitioning}"
        s += ", normalize={normalize}"
        s += ", algorithm_spatial={algorithm_spatial}"
        s += ")"

        return s.format(**self.__dict__)

    def update_once(self):
        eps = self.eps
        eng = self._engine

        self.update_source_model()
        self.update_spatial_model()

        if self.normalize:
            if self.normalize == 'power':
                if self.partitioning:  # unreachable in the reference too: the source update raises first
                    raise NotImplementedError("Only support when `partitioning=False` ")
                # T / aux**2 whatever `domain` says (ilrma.py:869-870)
                eng.ilrma_normalize_power_bins(self._Wd, self._Td, self._pbins, domain=2, eps=eps)
            else:
                raise ValueError("Not support normalization based on {}. Choose 'power' or 'projection-back'".format(self.normalize))
            self._touch("W", "T")
            self._estimation = None

    def update_source_model(self):
        """IS-NMF updates on the harmonic statistic (ilrma.py:880-922)."""
        assert self.domain == 2, "Only domain = 2 is supported."
        if self.partitioning:
            raise NotImplementedError("Only support when `partitioning=False` ")
        self._engine.tilrma_source_update(self._X, self._Wd, self._Td, self._Vd, self.nu, eps=self.eps)
        self._touch("T", "V")

    def update_spatial_model(self):
        """Xi-weighted covariance + IP without a condition guard (ilrma.py:926-983)."""
        eng = self._engine
        B, M, F, T = (int(s) for s in self._X.shape)
        C = pbins = None
        if self.normalize == 'power':
            if self._C is None:
                self._C = eng.cov_accumulate(self._X).reshape(B, F, M, M)
                self._pbins = eng.empty((B, M, F), dtype=torch.float64)
            C, pbins = self._C, self._pbins
        if self._Xi is None:
            self._Xi = eng.empty((B, M, F, T))
        eng.tilrma_spatial_update(self._X, self._Wd, self._Td, self._Vd, self.nu, self._Xi, eps=self.eps,
                                  status=self._status, C=C, power_bins=pbins)
        self._touch("W")
        self._estimation = None

    def _require_plain_basis(self):
        # The reference's t-ILRMA cannot update a partitioned model either (its update_source_model raises, ilrma.py:880-
        # 883); here the refusal comes BEFORE the criterion at entry: tilrma_loss reads basis as (n_sources, n_bins,
        # n_basis), and a partitioned basis (n_bins, n_basis) handed to it is read past its end (found as a memory fault
        # that only some allocation layouts produce).  Observable difference: iteration=0 raises too.
        if self.partitioning:
            raise NotImplementedError("Only support when `partitioning=False` ")

    def _record_loss(self):
        self._require_plain_basis()
        loss = self._engine.tilrma_loss(self._X, self._Wd, self._Td, self._Vd, self.nu, eps=self.eps)
        if isinstance(self.loss, LazyLossList):
            self.loss.append_device(loss, self._batched)
        else:
            self.loss.append(to_numpy(loss, np.float64) if self._batched else np.float64(loss.item()))

    def compute_negative_loglikelihood(self):
        """sum (1 + nu/2) log(1 + (2/nu) P/R) + log R - 2 T sum_f log|det W_f|   (ilrma.py:991-1018)."""
        self._require_plain_basis()
        loss = self._engine.tilrma_loss(self._X, self._Wd, self._Td, self._Vd, self.nu, eps=self.eps)
        self._check_status()
        if self._batched:
            return to_numpy(loss, np.float64)
        return np.float64(loss.item())
