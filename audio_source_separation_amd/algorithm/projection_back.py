"""projection_back on MI355X -- drop-in for `algorithm.projection_back.projection_back`
(/root/reference/src/algorithm/projection_back.py:3-34).  NumPy in, NumPy out; the statistics
sum_t y y^H, sum_t x_ref y^H and the per-bin N x N solve run on the device (assx_projection_back).
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from .. import _lib
from ..ops import Engine

_ENGINES = {}


def _engine(dtype, device):
    key = (dtype, str(device))
    if key not in _ENGINES:
        _ENGINES[key] = Engine(dtype=dtype, device=device)
    return _ENGINES[key]


def projection_back(Y, reference, *, dtype='float64', device=None):
    """
    Args:
        Y: (n_sources, n_bins, n_frames)
        reference: (n_bins, n_frames) or (n_channels, n_bins, n_frames)
    Returns:
        scale: (n_sources, n_bins) or (n_channels, n_sources, n_bins)
    """
    n_dims = reference.dim() if isinstance(reference, torch.Tensor) else np.ndim(reference)
    if n_dims not in (2, 3):
        raise ValueError("reference.ndim is expected 2 or 3, but given {}.".format(n_dims))
    eng = _engine(dtype, device)
    Yd = to_device(Y, eng.prec.cplx, eng.dev).unsqueeze(0).contiguous()
    Rd = to_device(reference, eng.prec.cplx, eng.dev)
    status = eng.new_status(1)
    if n_dims == 2:
        scale = eng.projection_back(Yd, Rd.unsqueeze(0).contiguous(), status)[0]
    else:
        scale = torch.stack([eng.projection_back(Yd, Rd[c].unsqueeze(0).contiguous(), status)[0]
                             for c in range(Rd.shape[0])])
    if int(status.item()) & _lib.STATUS_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    if isinstance(Y, torch.Tensor):
        return scale
    return to_numpy(scale, np.complex128)
