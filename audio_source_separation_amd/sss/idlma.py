"""The spatial update of IDLMA on MI355X.

`GaussIDLMA.update_space_model` (/root/reference/src/sss/idlma.py:175-210) is the same weighted covariance +
iterative-projection sweep as Gauss-ILRMA's, with the source variances coming from a DNN instead of an NMF model.
The DNN (the rest of IDLMA) is out of scope; the caller supplies its output.
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from .. import _lib
from ..algorithm.projection_back import _engine

EPS = 1e-12
THRESHOLD = 1e+12


def update_space_model(input, demix_filter, dnn_output, domain=2, eps=EPS, threshold=THRESHOLD, *, dtype='float64',
                       device=None):
    """
    Args:
        input (n_channels, n_bins, n_frames) complex: mixture STFT
        demix_filter (n_bins, n_sources, n_channels) complex: current W (not modified)
        dnn_output (n_sources, n_bins, n_frames) real >= 0: the source model's output; R = dnn_output**(2/domain)
    Returns:
        demix_filter (n_bins, n_sources, n_channels) after one IP sweep (idlma.py:196-208).
        NumPy in -> NumPy out; device tensors in -> device tensor out; a leading utterance axis is allowed on all three.
    """
    eng = _engine(dtype, device)
    X = to_device(input, eng.prec.cplx, eng.dev)
    W = to_device(demix_filter, eng.prec.cplx, eng.dev)
    R = to_device(dnn_output, eng.prec.real, eng.dev)
    batched = X.dim() == 4
    if not batched:
        X, W, R = X.unsqueeze(0), W.unsqueeze(0), R.unsqueeze(0)
    if tuple(R.shape) != tuple(X.shape) or tuple(W.shape) != (X.shape[0], X.shape[2], X.shape[1], X.shape[1]):
        raise ValueError("shapes do not match: input {}, demix_filter {}, dnn_output {}".format(
            tuple(X.shape), tuple(W.shape), tuple(R.shape)))
    W = W.contiguous().clone()
    status = eng.new_status(X.shape[0])
    eng.idlma_space_update(X.contiguous(), W, R.contiguous(), domain=domain, eps=eps, threshold=threshold, status=status)
    if int(status.max().item()) & _lib.STATUS_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    if isinstance(input, torch.Tensor) and isinstance(demix_filter, torch.Tensor):
        return W if batched else W[0]
    W = to_numpy(W, np.complex128)
    return W if batched else W[0]
