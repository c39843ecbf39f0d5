"""ctypes binding of libassx.so (the C-ABI declared in include/assx.h).

There is deliberately NO fallback: if the HIP library is missing the import fails loudly, and
every call checks its status code and raises with the library's own message.
"""
import ctypes
import os

# PyTorch-ROCm bundles its own HIP runtime (SONAME libamdhip64.so.7, same as /opt/rocm's).  The process must
# hold exactly ONE HIP runtime or device pointers / streams cannot be shared and the second runtime finds no
# device: importing torch first makes the loader bind libassx.so to the runtime torch already mapped.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASSX_LIB_PATH: another build of the same library (A/B measurements of kernel variants; never a fallback)
LIB_PATH = os.environ.get("ASSX_LIB_PATH") or os.path.join(_HERE, "csrc", "libassx.so")

F32, F64 = 0, 1
W_NONE, W_NT, W_NFT = 0, 1, 2
IVA_LAPLACE, IVA_GAUSS = 0, 1
NMF_EUC, NMF_KL, NMF_IS_MM, NMF_IS_ME = 0, 1, 2, 3
NMF_T, NMF_CAUCHY_NAIVE, NMF_CAUCHY_MM, NMF_CAUCHY_ME, NMF_CAUCHY_MM_FAST = 4, 5, 6, 7, 8
STATUS_SINGULAR, STATUS_COND_REJECT = 1, 2
SPATIAL_IP, SPATIAL_ISS, SPATIAL_IP2 = 0, 1, 2

_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
_sz = ctypes.c_size_t
_ll = ctypes.c_longlong

# name -> (restype, argtypes) ; must list EVERY symbol include/assx.h declares (tests check this)
SIGNATURES = {
    "assx_ctx_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "assx_ctx_destroy": (_i, [_vp]),
    "assx_last_error": (ctypes.c_char_p, [_vp]),
    "assx_version": (ctypes.c_char_p, []),
    "assx_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "assx_launch_order": (_i, [_i, _i, _i, _i, _vp, _i]),
    "assx_nmf_partition_query": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_demix": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_cov_accumulate": (_i, [_vp, _vp, _vp, _i, _d, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ip_update": (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _vp]),
    "assx_ilrma_source_update": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, ctypes.c_uint, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_expand_partitioned": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_source_update_partitioned": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_normalize_power_bins_partitioned": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_spatial_update": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ip2_update": (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_iss_update": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_cov_partials": (_i, [_vp, _vp, _vp, _vp, _d, _d, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_demix_power": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_power_from_cov": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "assx_ilrma_normalize_power": (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_normalize_power_bins": (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_normalize_pb": (_i, [_vp, _vp, _vp, _vp, _d, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_tilrma_source_update": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_tilrma_spatial_update": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_tilrma_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_auxiva_weights": (_i, [_vp, _vp, _vp, _i, _d, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_auxiva_spatial_update": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_idlma_space_update": (_i, [_vp, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_fastmnmf_update_diagonalizer": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "assx_projection_back_scale": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_projection_back": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_compute_demix_filter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "assx_nmf_update": (_i, [_vp, _i, _d, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_loss": (_i, [_vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_update_ex": (_i, [_vp, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_loss_ex": (_i, [_vp, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_iterate": (_i, [_vp, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_auxiva_iterate": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_iterate": (_i, [_vp, _i, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i, _i, _i, _i, _i, _i, _vp]),
    "assx_ilrma_power_map": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_half_sums": (_i, [_vp, _i, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_nmf_apply_sums": (_i, [_vp, _i, _d, _d, _vp, _vp, _i, _ll, _i, _vp]),
    "assx_ordered_sum": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp]),
    "assx_stft_num_frames": (_ll, [_ll, _i, _i]),
    "assx_istft_num_samples": (_ll, [_i, _i, _i]),
    "assx_stft_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "assx_stft": (_i, [_vp, _vp, _vp, _d, _vp, _vp, _i, _ll, _i, _i, _i, _i, _vp]),
    "assx_istft": (_i, [_vp, _vp, _vp, _d, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "assx_upload": (_i, [_vp, _vp, _i, _vp, _i, _sz, _vp]),
    "assx_download": (_i, [_vp, _vp, _i, _vp, _i, _sz, _vp]),
    "assx_shard_range": (None, [_sz, _i, _i, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "assx_comm_unique_id": (_i, [_vp]),
    "assx_comm_init": (_i, [_vp, _i, _i, _vp, ctypes.POINTER(_vp)]),
    "assx_comm_destroy": (_i, [_vp]),
    "assx_scatter": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _vp]),
    "assx_gather": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _vp]),
}
COMM_ID_BYTES = 128


class AssxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP library not built: %s is missing.  Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`audio_source_separation_amd/csrc/build.sh` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def version():
    return lib.assx_version().decode()


def check(ctx, rc, what):
    if rc != 0:
        msg = lib.assx_last_error(ctx).decode(errors="replace") if ctx else ""
        kind = "invalid argument" if rc < 0 else "HIP error"
        raise AssxError("%s failed (%s %d): %s" % (what, kind, rc, msg))
