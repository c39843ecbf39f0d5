"""The diagonaliser update of FastMNMF on MI355X.

`FastMultichannelISNMF.update_diagonalizer` (/root/reference/src/bss/mnmf.py:848-888) is a weighted covariance per
CHANNEL m -- weights R[f,t,m] = sum_n Lambda[n,f,t] g[n,f,m] -- followed by the iterative-projection sweep with the
normaliser floored at eps: the very kernels of the ILRMA spatial update.  The rest of FastMNMF (NMF and spatial
covariance updates) is out of scope.
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from .. import _lib
from ..algorithm.projection_back import _engine

EPS = 1e-12
THRESHOLD = 1e+12


def source_variance(basis, activation, latent=None):
    """Lambda (n_sources, n_bins, n_frames) as update_diagonalizer forms it (mnmf.py:858-866): W @ H, or with a
    partitioning function (latent Z (n_sources, n_basis), shared W (n_bins, n_basis), H (n_basis, n_frames))
    (Z[:, None, :] * W[None]) @ H.  Host-side convenience: a caller that keeps its NMF model elsewhere passes
    `variance=` to update_diagonalizer instead."""
    if latent is not None:
        return (latent[:, None, :] * basis[None, :, :]) @ activation[None, :, :]
    return basis @ activation


def update_diagonalizer(input, diagonalizer, spatial_covariance, variance=None, basis=None, activation=None, latent=None,
                        eps=EPS, threshold=THRESHOLD, *, dtype='float64', device=None):
    """
    Args:
        input (n_channels, n_bins, n_frames) complex
        diagonalizer Q (n_bins, n_channels, n_channels) complex (not modified)
        spatial_covariance g (n_sources, n_bins, n_channels) real
        variance Lambda (n_sources, n_bins, n_frames) real, or (basis, activation[, latent]) to form it
    Returns:
        Q after the sweep over the channels (mnmf.py:872-886).
    """
    if variance is None:
        if basis is None or activation is None:
            raise ValueError("Specify `variance` or (`basis`, `activation`).")
        variance = source_variance(np.asarray(basis), np.asarray(activation), None if latent is None else np.asarray(latent))
    eng = _engine(dtype, device)
    X = to_device(input, eng.prec.cplx, eng.dev)
    Q = to_device(diagonalizer, eng.prec.cplx, eng.dev)
    L = to_device(variance, eng.prec.real, eng.dev)
    g = to_device(spatial_covariance, eng.prec.real, eng.dev)
    batched = X.dim() == 4
    if not batched:
        X, Q, L, g = X.unsqueeze(0), Q.unsqueeze(0), L.unsqueeze(0), g.unsqueeze(0)
    B, M, F, T = (int(v) for v in X.shape)
    N = int(L.shape[1])
    if tuple(Q.shape) != (B, F, M, M) or tuple(L.shape) != (B, N, F, T) or tuple(g.shape) != (B, N, F, M):
        raise ValueError("shapes do not match: input {}, diagonalizer {}, variance {}, spatial_covariance {}".format(
            tuple(X.shape), tuple(Q.shape), tuple(L.shape), tuple(g.shape)))
    Q = Q.contiguous().clone()
    status = eng.new_status(B)
    eng.fastmnmf_update_diagonalizer(X.contiguous(), Q, L.contiguous(), g.contiguous(), eps=eps, threshold=threshold,
                                     status=status)
    if int(status.max().item()) & _lib.STATUS_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    if isinstance(input, torch.Tensor) and isinstance(diagonalizer, torch.Tensor):
        return Q if batched else Q[0]
    Q = to_numpy(Q, np.complex128)
    return Q if batched else Q[0]
