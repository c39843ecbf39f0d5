"""Thin, typed wrappers over the C-ABI working on batched device tensors.

Shapes follow include/assx.h (leading utterance axis B):  X (B,M,F,T) complex, W (B,F,N,M) complex,
Tb (B,N,F,K), V (B,N,K,T), U (B,N,F,M,M), Y (B,N,F,T).  Every method launches asynchronously on
torch's current stream for the engine's device; none of them synchronises.
"""
import ctypes

import torch

from . import _lib
from ._device import Precision, Workspace, context, ptr, require_gpu, stream_ptr

_RAW = _lib.lib


class _DeviceGuardedLib:
    """libassx entry points called with the engine's device made current for the duration of the call.

    The C library never switches devices (it refuses a call whose context device is not current); torch owns the
    calling thread's current device, so a model built with device='cuda:1' while cuda:0 is current -- or two models
    on different GPUs in one process -- are handled here, not by the caller."""

    def __init__(self, dev):
        self._dev = dev

    def __getattr__(self, name):
        fn = getattr(_RAW, name)
        dev, index = self._dev, self._dev.index

        def call(*args):
            if torch.cuda.current_device() == index:
                return fn(*args)
            with torch.cuda.device(dev):
                return fn(*args)

        call.__name__ = name
        setattr(self, name, call)
        return call


class Engine:
    def __init__(self, dtype="float64", device=None):
        self.dev = require_gpu(device)
        self.prec = Precision(dtype)
        context(self.dev)  # fail here, not at the first kernel, if the device cannot be opened
        self._ws = Workspace(self.dev)
        self._L = _DeviceGuardedLib(self.dev)

    @property
    def ctx(self):
        """The CALLING thread's context for this engine's device (contexts are per thread, include/assx.h)."""
        return context(self.dev)

    # ------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        _lib.check(self.ctx, rc, what)

    def _scratch(self, B, M, F, T, K):
        n = self._L.assx_workspace_bytes(B, M, F, T, max(int(K), 1), self.prec.code)
        return self._ws.get(n)

    def _st(self):
        return stream_ptr(self.dev)

    def empty(self, shape, complex_=False, dtype=None):
        dt = dtype if dtype is not None else (self.prec.cplx if complex_ else self.prec.real)
        return torch.empty(shape, dtype=dt, device=self.dev)

    def new_status(self, B):
        return torch.zeros(B, dtype=torch.int32, device=self.dev)

    @staticmethod
    def _dims(X):
        B, M, F, T = X.shape
        return int(B), int(M), int(F), int(T)

    # ------------------------------------------------------------------ (a3)
    @staticmethod
    def _model_shapes(X, W=None, Tb=None, V=None):
        """The C-ABI takes pointers and sizes: a model array of another shape would be read past its end (a partitioned
        basis (F, K) handed to an entry point that reads (N, F, K) was found as an intermittent memory fault).  Refuse it
        here, where the shapes are still known."""
        B, M, F, T = (int(d) for d in X.shape)
        if W is not None and tuple(W.shape) != (B, F, M, M):
            raise ValueError("demix_filter: expected shape %s, got %s" % ((B, F, M, M), tuple(W.shape)))
        if Tb is not None:
            K = int(Tb.shape[-1])
            if tuple(Tb.shape) != (B, M, F, K):
                raise ValueError("basis: expected shape %s, got %s" % ((B, M, F, K), tuple(Tb.shape)))
            if V is not None and tuple(V.shape) != (B, M, K, T):
                raise ValueError("activation: expected shape %s, got %s" % ((B, M, K, T), tuple(V.shape)))

    def demix(self, X, W, scale=None, out=None):
        B, M, F, T = self._dims(X)
        Y = out if out is not None else self.empty((B, M, F, T), complex_=True)
        self._check(self._L.assx_demix(self.ctx, ptr(X), ptr(W), ptr(scale), ptr(Y), B, M, F, T, self.prec.code, self._st()),
                    "assx_demix")
        return Y

    # ------------------------------------------------------------------ (a4)
    def cov_accumulate(self, X, r=None, eps=1e-12):
        """r: None (plain covariance, returns (B,1,F,M,M)), (B,N,T) or (B,N,F,T)."""
        B, M, F, T = self._dims(X)
        if r is None:
            kind, N = _lib.W_NONE, 1
        elif r.dim() == 3:
            kind, N = _lib.W_NT, int(r.shape[1])
        else:
            kind, N = _lib.W_NFT, int(r.shape[1])
        U = self.empty((B, N, F, M, M), complex_=True)
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_cov_accumulate(self.ctx, ptr(X), ptr(r), kind, float(eps), ptr(U), ptr(ws), B, M, N, F, T,
                                          self.prec.code, self._st()), "assx_cov_accumulate")
        return U

    # ------------------------------------------------------------------ (a5)
    def ip_update(self, U, W, threshold=1e12, status=None):
        B, F, N, M = (int(s) for s in W.shape)
        self._check(self._L.assx_ip_update(self.ctx, ptr(U), ptr(W), float(threshold), ptr(status), B, M, F, self.prec.code,
                                     self._st()), "assx_ip_update")
        return W

    # ------------------------------------------------------------------ ILRMA
    def ilrma_source_update(self, X, W, Tb, V, domain=2, eps=1e-12, sources=None, loss_prev=None):
        """sources: None = all, or an iterable of source indices (pairwise update).
        loss_prev: optional (B,) float64 tensor receiving the loss of the model at entry (fused into the basis pass)."""
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        mask = (1 << M) - 1 if sources is None else sum(1 << int(n) for n in set(sources))
        self._check(self._L.assx_ilrma_source_update(self.ctx, ptr(X), ptr(W), ptr(Tb), ptr(V), float(domain), float(eps),
                                               mask, ptr(loss_prev), ptr(ws), B, M, F, T, K, self.prec.code,
                                               self._st()),
                    "assx_ilrma_source_update")

    # ---- partitioning function (shared bases + latent variables), domain 2
    def ilrma_expand_partitioned(self, Z, Tb, V, Teff, Veff):
        """Teff (B,N,F,K) = Z[n,k] Tb[f,k], Veff (B,N,K,T) = V[k,t]; either output may be None."""
        B, N, K = (int(s) for s in Z.shape)
        F, T = int(Tb.shape[1]), int(V.shape[2])
        self._check(self._L.assx_ilrma_expand_partitioned(self.ctx, ptr(Z), ptr(Tb), ptr(V), ptr(Teff), ptr(Veff), B, N, F,
                                                    T, K, self.prec.code, self._st()),
                    "assx_ilrma_expand_partitioned")

    def ilrma_source_update_partitioned(self, X, W, Z, Tb, V, Teff, Veff, eps=1e-12):
        B, M, F, T = self._dims(X)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_ilrma_source_update_partitioned(self.ctx, ptr(X), ptr(W), ptr(Z), ptr(Tb), ptr(V),
                                                           ptr(Teff), ptr(Veff), float(eps), ptr(ws), B, M, F, T, K,
                                                           self.prec.code, self._st()),
                    "assx_ilrma_source_update_partitioned")

    def ilrma_normalize_power_bins_partitioned(self, W, Z, Tb, power_bins, n_frames, eps=1e-12):
        B, F, N, M = (int(s) for s in W.shape)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, int(n_frames), K)
        self._check(self._L.assx_ilrma_normalize_power_bins_partitioned(self.ctx, ptr(W), ptr(Z), ptr(Tb), ptr(power_bins),
                                                                  float(eps), ptr(ws), B, M, F, K, self.prec.code,
                                                                  self._st()),
                    "assx_ilrma_normalize_power_bins_partitioned")

    def ip2_update(self, U, W, pair, threshold=1e12, status=None):
        B, F, N, M = (int(s) for s in W.shape)
        self._check(self._L.assx_ip2_update(self.ctx, ptr(U), ptr(W), float(threshold), ptr(status), int(pair[0]),
                                      int(pair[1]), B, M, F, self.prec.code, self._st()), "assx_ip2_update")
        return W

    def iss_update(self, U, W, n_frames):
        B, F, N, M = (int(s) for s in W.shape)
        self._check(self._L.assx_iss_update(self.ctx, ptr(U), ptr(W), int(n_frames), B, M, F, self.prec.code, self._st()),
                    "assx_iss_update")
        return W

    def ilrma_spatial_update(self, X, W, Tb, V, domain=2, eps=1e-12, threshold=1e12, status=None, U_out=None,
                             C=None, power_bins=None, spatial=_lib.SPATIAL_IP, pair=(0, 1)):
        """C (B,F,M,M) + power_bins (B,N,F) float64: also emit the per-bin power statistic of the updated filters."""
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_ilrma_spatial_update(self.ctx, int(spatial), int(pair[0]), int(pair[1]), ptr(X), ptr(W), ptr(Tb), ptr(V), float(domain), float(eps),
                                                float(threshold), ptr(U_out), ptr(C), ptr(power_bins), ptr(status),
                                                ptr(ws), B, M, F, T, K, self.prec.code, self._st()),
                    "assx_ilrma_spatial_update")

    def ilrma_cov_partials(self, X, Tb, V, domain=2, eps=1e-12):
        """ONE launch of the covariance-accumulate kernel (stage 1 of ilrma_spatial_update); for kernel timing."""
        B, M, F, T = self._dims(X)
        self._model_shapes(X, None, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_ilrma_cov_partials(self.ctx, ptr(X), ptr(Tb), ptr(V), float(domain), float(eps), ptr(ws),
                                              B, M, F, T, K, self.prec.code, self._st()), "assx_ilrma_cov_partials")

    def demix_power(self, X, W, out=None):
        B, M, F, T = self._dims(X)
        p = out if out is not None else self.empty((B, M))
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_demix_power(self.ctx, ptr(X), ptr(W), ptr(p), ptr(ws), B, M, F, T, self.prec.code,
                                       self._st()), "assx_demix_power")
        return p

    def power_from_cov(self, C, W, T_frames, out=None):
        B, F, N, M = (int(s) for s in W.shape)
        p = out if out is not None else self.empty((B, M))
        ws = self._scratch(B, M, F, int(T_frames), 1)
        self._check(self._L.assx_power_from_cov(self.ctx, ptr(C), ptr(W), ptr(p), ptr(ws), B, M, F, self.prec.code,
                                          self._st()), "assx_power_from_cov")
        return p

    def ilrma_normalize_power(self, W, Tb, power, domain=2, eps=1e-12):
        B, F, N, M = (int(s) for s in W.shape)
        K = int(Tb.shape[-1])
        self._check(self._L.assx_ilrma_normalize_power(self.ctx, ptr(W), ptr(Tb), ptr(power), float(domain), float(eps), B, M,
                                                 F, K, self.prec.code, self._st()), "assx_ilrma_normalize_power")

    def ilrma_normalize_power_bins(self, W, Tb, power_bins, domain=2, eps=1e-12):
        B, F, N, M = (int(s) for s in W.shape)
        K = int(Tb.shape[-1])
        self._check(self._L.assx_ilrma_normalize_power_bins(self.ctx, ptr(W), ptr(Tb), ptr(power_bins), float(domain),
                                                      float(eps), B, M, F, K, self.prec.code, self._st()),
                    "assx_ilrma_normalize_power_bins")

    def ilrma_normalize_pb(self, W, Tb, scale, domain=2):
        B, F, N, M = (int(s) for s in W.shape)
        K = int(Tb.shape[-1])
        self._check(self._L.assx_ilrma_normalize_pb(self.ctx, ptr(W), ptr(Tb), ptr(scale), float(domain), B, M, F, K,
                                              self.prec.code, self._st()), "assx_ilrma_normalize_pb")

    def ilrma_loss(self, X, W, Tb, V, domain=2, eps=1e-12, out=None):
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        loss = out if out is not None else self.empty((B,), dtype=torch.float64)
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_ilrma_loss(self.ctx, ptr(X), ptr(W), ptr(Tb), ptr(V), float(domain), float(eps), ptr(loss),
                                      ptr(ws), B, M, F, T, K, self.prec.code, self._st()), "assx_ilrma_loss")
        return loss

    # ------------------------------------------------------------------ whole loops in one call
    def nmf_iterate(self, n_iter, kind, X, Tb, V, domain=2, eps=1e-12, param=0.0, loss=None):
        """n_iter x (update, loss[i]); loss: (n_iter, B) float64 device tensor or None (criterion not evaluated)."""
        B, F, T = (int(s) for s in X.shape)
        K = int(Tb.shape[-1])
        ws = self._nmf_scratch(B, F, T, K)
        self._check(self._L.assx_nmf_iterate(self.ctx, int(n_iter), int(kind), float(domain), float(param), float(eps),
                                             ptr(X), ptr(Tb), ptr(V), ptr(loss), ptr(ws), B, F, T, K, self.prec.code,
                                             self._st()), "assx_nmf_iterate")

    def auxiva_iterate(self, n_iter, kind, X, W, r, eps=1e-12, threshold=1e12, status=None, loss=None,
                       spatial=_lib.SPATIAL_IP, pair=(0, 1)):
        """n_iter x (weights [+ loss[i]], covariance + sweep), then loss[n_iter]; loss: (n_iter + 1, B) float64 or None."""
        B, M, F, T = self._dims(X)
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_auxiva_iterate(self.ctx, int(n_iter), int(kind), int(spatial), int(pair[0]), int(pair[1]),
                                                ptr(X), ptr(W), float(eps), float(threshold), ptr(r), ptr(loss),
                                                ptr(status), ptr(ws), B, M, F, T, self.prec.code, self._st()),
                    "assx_auxiva_iterate")

    def ilrma_iterate(self, n_iter, X, W, Tb, V, domain=2, eps=1e-12, threshold=1e12, status=None, loss=None,
                      spatial=_lib.SPATIAL_IP, pair=(0, 1), normalize=0, C=None, power_bins=None, scale=None, ref=0,
                      pb_exponent=2.0):
        """n_iter x (source model, spatial model, normalisation); loss: (n_iter + 1, B) float64 or None.
        normalize: 0 none, 1 'power' (C + power_bins), 2 'projection-back' (scale scratch, ref, pb_exponent)."""
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_ilrma_iterate(self.ctx, int(n_iter), int(spatial), int(pair[0]), int(pair[1]),
                                               int(normalize), int(ref), float(pb_exponent), ptr(X), ptr(W), ptr(Tb),
                                               ptr(V), float(domain), float(eps), float(threshold), ptr(C),
                                               ptr(power_bins), ptr(scale), ptr(loss), ptr(status), ptr(ws), B, M, F,
                                               T, K, self.prec.code, self._st()), "assx_ilrma_iterate")

    # ------------------------------------------------------------------ STFT / iSTFT (row f3)
    def stft(self, x, window, fft_size, hop):
        """x (C, L) real -> X (C, fft_size//2+1, n_frames) complex; scipy.signal.stft semantics (include/assx.h)."""
        C, n = int(x.shape[0]), int(x.shape[1])
        T = int(self._L.assx_stft_num_frames(n, fft_size, hop))
        X = self.empty((C, fft_size // 2 + 1, max(T, 0)), complex_=True)
        ws = self._ws.get(self._L.assx_stft_workspace_bytes(C, fft_size, max(T, 1), self.prec.code))
        self._check(self._L.assx_stft(self.ctx, ptr(x), ptr(window), float(window.sum().item()), ptr(X), ptr(ws), C, n,
                                fft_size, hop, T, self.prec.code, self._st()), "assx_stft")
        return X

    def istft(self, X, window, fft_size, hop):
        """X (C, fft_size//2+1, n_frames) complex -> y (C, n_samples) real; scipy.signal.istft semantics."""
        C, F, T = (int(v) for v in X.shape)
        if F != fft_size // 2 + 1:
            raise ValueError("istft: {} bins do not match fft_size={}".format(F, fft_size))
        n = int(self._L.assx_istft_num_samples(fft_size, hop, T))
        y = self.empty((C, max(n, 0)))
        ws = self._ws.get(self._L.assx_stft_workspace_bytes(C, fft_size, T, self.prec.code))
        self._check(self._L.assx_istft(self.ctx, ptr(X), ptr(window), float(window.sum().item()), ptr(y), ptr(ws), C,
                                 fft_size, hop, T, self.prec.code, self._st()), "assx_istft")
        return y

    # ------------------------------------------------------------------ t-ILRMA
    def tilrma_source_update(self, X, W, Tb, V, nu, eps=1e-12):
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_tilrma_source_update(self.ctx, ptr(X), ptr(W), ptr(Tb), ptr(V), float(nu), float(eps),
                                                ptr(ws), B, M, F, T, K, self.prec.code, self._st()),
                    "assx_tilrma_source_update")

    def tilrma_spatial_update(self, X, W, Tb, V, nu, Xi, eps=1e-12, status=None, C=None, power_bins=None):
        """Xi: scratch (B,N,F,T) reals receiving the auxiliary weights."""
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_tilrma_spatial_update(self.ctx, ptr(X), ptr(W), ptr(Tb), ptr(V), float(nu), float(eps),
                                                 ptr(Xi), ptr(C), ptr(power_bins), ptr(status), ptr(ws), B, M, F, T,
                                                 K, self.prec.code, self._st()), "assx_tilrma_spatial_update")
        return W

    def tilrma_loss(self, X, W, Tb, V, nu, eps=1e-12, out=None):
        B, M, F, T = self._dims(X)
        self._model_shapes(X, W, Tb, V)
        K = int(Tb.shape[-1])
        loss = out if out is not None else self.empty((B,), dtype=torch.float64)
        ws = self._scratch(B, M, F, T, K)
        self._check(self._L.assx_tilrma_loss(self.ctx, ptr(X), ptr(W), ptr(Tb), ptr(V), float(nu), float(eps), ptr(loss),
                                       ptr(ws), B, M, F, T, K, self.prec.code, self._st()), "assx_tilrma_loss")
        return loss

    # ------------------------------------------------------------------ AuxIVA
    def auxiva_weights(self, X, W, kind, eps=1e-12, with_loss=False, out=None):
        B, M, F, T = self._dims(X)
        r = out if out is not None else self.empty((B, M, T))
        loss = self.empty((B,), dtype=torch.float64) if with_loss else None
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_auxiva_weights(self.ctx, ptr(X), ptr(W), int(kind), float(eps), ptr(r), ptr(loss), ptr(ws),
                                          B, M, F, T, self.prec.code, self._st()), "assx_auxiva_weights")
        return r, loss

    def auxiva_spatial_update(self, X, W, r, eps=1e-12, threshold=1e12, status=None, U_out=None,
                              spatial=_lib.SPATIAL_IP, pair=(0, 1)):
        B, M, F, T = self._dims(X)
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_auxiva_spatial_update(self.ctx, int(spatial), int(pair[0]), int(pair[1]), ptr(X), ptr(W), ptr(r), float(eps), float(threshold),
                                                 ptr(U_out), ptr(status), ptr(ws), B, M, F, T, self.prec.code,
                                                 self._st()), "assx_auxiva_spatial_update")

    # ------------------------------------------------------------------ other callers of cov + IP (row f4)
    def idlma_space_update(self, X, W, dnn_output, domain=2, eps=1e-12, threshold=1e12, status=None):
        """GaussIDLMA.update_space_model: W (B,F,N,M) in place from the source variances dnn_output (B,N,F,T)."""
        B, M, F, T = self._dims(X)
        ws = self._scratch(B, M, F, T, 1)
        scratch = None if float(domain) == 2.0 else self.empty((B, M, F, T))
        self._check(self._L.assx_idlma_space_update(self.ctx, ptr(X), ptr(W), ptr(dnn_output), float(domain), float(eps),
                                                    float(threshold), ptr(scratch), ptr(status), ptr(ws), B, M, F, T,
                                                    self.prec.code, self._st()), "assx_idlma_space_update")
        return W

    def fastmnmf_update_diagonalizer(self, X, Q, Lambda, g, eps=1e-12, threshold=1e12, status=None):
        """FastMultichannelISNMF.update_diagonalizer: Q (B,F,M,M) in place; Lambda (B,N,F,T), g (B,N,F,M)."""
        B, M, F, T = self._dims(X)
        N = int(Lambda.shape[1])
        ws = self._scratch(B, M, F, T, 1)
        scratch = self.empty((B, M, F, T))
        self._check(self._L.assx_fastmnmf_update_diagonalizer(self.ctx, ptr(X), ptr(Q), ptr(Lambda), ptr(g), float(eps),
                                                              float(threshold), ptr(scratch), ptr(status), ptr(ws), B, M,
                                                              N, F, T, self.prec.code, self._st()),
                    "assx_fastmnmf_update_diagonalizer")
        return Q

    # ------------------------------------------------------------------ projection back
    def projection_back_scale(self, X, W, ref=0, status=None):
        B, M, F, T = self._dims(X)
        scale = self.empty((B, M, F), complex_=True)
        ws = self._scratch(B, M, F, T, 1)
        self._check(self._L.assx_projection_back_scale(self.ctx, ptr(X), ptr(W), int(ref), ptr(scale), ptr(status), ptr(ws),
                                                 B, M, F, T, self.prec.code, self._st()), "assx_projection_back_scale")
        return scale

    def projection_back(self, Y, reference, status=None):
        B, N, F, T = self._dims(Y)
        scale = self.empty((B, N, F), complex_=True)
        ws = self._scratch(B, N, F, T, 1)
        self._check(self._L.assx_projection_back(self.ctx, ptr(Y), ptr(reference), ptr(scale), ptr(status), ptr(ws), B, N, F,
                                           T, self.prec.code, self._st()), "assx_projection_back")
        return scale

    # ------------------------------------------------------------------ least-squares demixing filter
    def compute_demix_filter(self, Y, X, status=None):
        """W (B,F,M,M) = (Y X^H)(X X^H)^-1 per bin for Y, X (B,M,F,T)."""
        B, M, F, T = self._dims(X)
        W = self.empty((B, F, M, M), complex_=True)
        self._check(self._L.assx_compute_demix_filter(self.ctx, ptr(Y), ptr(X), ptr(W), ptr(status), B, M, F, T,
                                                      self.prec.code, self._st()), "assx_compute_demix_filter")
        return W

    # ------------------------------------------------------------------ pieces for the F-sharded mode (row f2)
    def ilrma_power_map(self, X, W, out=None):
        """P (B,N,F,T) real = |W x|^2."""
        B, M, F, T = self._dims(X)
        P = out if out is not None else self.empty((B, M, F, T))
        self._check(self._L.assx_ilrma_power_map(self.ctx, ptr(X), ptr(W), ptr(P), B, M, F, T, self.prec.code, self._st()),
                    "assx_ilrma_power_map")
        return P

    def nmf_half_sums(self, kind, half, X, Tb, V, domain=2, eps=1e-12, param=0.0, out=None):
        """(2, B, F*K) [half 0: basis] or (2, B, K*T) [half 1: activation] numerators / denominators of one half of a
        multiplicative update on X (B,F,T), not applied."""
        B, F, T = (int(s) for s in X.shape)
        K = int(Tb.shape[-1])
        count = F * K if int(half) == 0 else K * T
        sums = out if out is not None else self.empty((2, B, count))
        ws = self._nmf_scratch(B, F, T, K)
        self._check(self._L.assx_nmf_half_sums(self.ctx, int(kind), float(domain), float(param), float(eps), int(half),
                                               ptr(X), ptr(Tb), ptr(V), ptr(sums), ptr(ws), B, F, T, K, self.prec.code,
                                               self._st()), "assx_nmf_half_sums")
        return sums

    def nmf_apply_sums(self, kind, A, sums, domain=2, eps=1e-12):
        """A (B, ...) *= (num / max(den, eps)) ** exponent(kind, domain), sums (2, B, count)."""
        B = int(sums.shape[1])
        count = int(sums.shape[2])
        self._check(self._L.assx_nmf_apply_sums(self.ctx, int(kind), float(domain), float(eps), ptr(A), ptr(sums), B,
                                                count, self.prec.code, self._st()), "assx_nmf_apply_sums")
        return A

    def ordered_sum(self, parts, weights=None):
        """sum over the leading axis of `parts` (S, ...) in ascending order, optionally weighted (float64 (S,))."""
        S = int(parts.shape[0])
        count = int(parts[0].numel())
        if parts.dtype not in (torch.float64, torch.float32):
            raise ValueError("ordered_sum: float64 / float32 only")
        code = _lib.F64 if parts.dtype == torch.float64 else _lib.F32
        out = torch.empty(parts.shape[1:], dtype=parts.dtype, device=self.dev)
        self._check(self._L.assx_ordered_sum(self.ctx, ptr(parts.contiguous()), ptr(weights), ptr(out), S, count, code,
                                             self._st()), "assx_ordered_sum")
        return out

    # ------------------------------------------------------------------ NMF
    def _nmf_scratch(self, B, F, T, K):
        return self._ws.get(self._L.assx_nmf_workspace_bytes(B, F, T, K, self.prec.code))

    def nmf_update(self, kind, X, Tb, V, domain=2, eps=1e-12, param=0.0):
        B, F, T = (int(s) for s in X.shape)
        K = int(Tb.shape[-1])
        ws = self._nmf_scratch(B, F, T, K)
        self._check(self._L.assx_nmf_update_ex(self.ctx, int(kind), float(domain), float(param), float(eps), ptr(X), ptr(Tb),
                                         ptr(V), ptr(ws), B, F, T, K, self.prec.code, self._st()), "assx_nmf_update_ex")

    def nmf_loss(self, kind, X, Tb, V, domain=2, eps=1e-12, out=None, param=0.0):
        B, F, T = (int(s) for s in X.shape)
        K = int(Tb.shape[-1])
        loss = out if out is not None else self.empty((B,), dtype=torch.float64)
        ws = self._nmf_scratch(B, F, T, K)
        self._check(self._L.assx_nmf_loss_ex(self.ctx, int(kind), float(domain), float(param), float(eps), ptr(X), ptr(Tb),
                                       ptr(V), ptr(loss), ptr(ws), B, F, T, K, self.prec.code, self._st()),
                    "assx_nmf_loss_ex")
        return loss
