"""ctypes binding of the C-ABI in include/ganspace_hip.h.

The product path has NO CPU fallback: if the shared library is missing or cannot be
loaded, :func:`load` raises ``RuntimeError`` (build it with
``python -m ganspace_amd._build`` / ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

GS_OK, GS_EINVAL, GS_EHIP, GS_ENOMEM, GS_ESTATE, GS_ENOTIMPL, GS_ENOCONV = 0, -1, -2, -3, -4, -5, -6
GS_MODE_EXACT, GS_MODE_FAITHFUL, GS_MODE_SMALLSIDE = 0, 1, 2
GS_PREC_F32, GS_PREC_BF16X3, GS_PREC_BF16X6, GS_PREC_BF16 = 0, 1, 2, 3
PRECISIONS = {"f32": GS_PREC_F32, "bf16x3": GS_PREC_BF16X3, "bf16x6": GS_PREC_BF16X6, "bf16": GS_PREC_BF16}

_vp, _i64, _int, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes): must list every symbol include/ganspace_hip.h declares
SIGNATURES = {
    "gs_version": (_int, []),
    "gs_last_error": (C.c_char_p, []),
    "gs_device_count": (_int, []),
    "gs_ipca_create": (_int, [_i64, _int, _int, _int, _int, C.POINTER(_vp)]),
    "gs_ipca_destroy": (_int, [_vp]),
    "gs_ipca_reset": (_int, [_vp]),
    "gs_ipca_update": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gs_ipca_update_resident": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gs_ipca_state_nbytes": (_i64, [_vp]),
    "gs_ipca_state_export": (_int, [_vp, _vp, _vp]),
    "gs_ipca_state_import": (_int, [_vp, _vp, _vp]),
    "gs_state_recenter": (_int, [_vp, _i64, _vp, _vp]),
    "gs_ipca_lowrank_nbytes": (_i64, [_vp]),
    "gs_ipca_lowrank_export": (_int, [_vp, _vp, _vp]),
    "gs_ipca_lowrank_merge": (_int, [_vp, _vp, _int, _vp]),
    "gs_ipca_info": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "gs_ipca_allreduce": (_int, [_vp, _vp, _vp]),
    "gs_ipca_finalize": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gs_ipca_last_sweeps": (_int, [_vp]),
    "gs_ipca_last_mults": (_int, [_vp]),
    "gs_ipca_components_device": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "gs_randomized_pca": (_int, [_vp, _i64, _i64, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "gs_column_moments": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gs_zgen_fill": (C.c_int, [C.c_uint32, _i64, _vp]),
    "gs_zgen_start": (_int, [_vp, _i64, _i64, _vp, _int, _int, C.POINTER(_vp)]),
    "gs_zgen_fill_truncnorm": (_int, [C.c_uint32, _i64, C.c_double, C.c_double, _f32, _vp]),
    "gs_zgen_start_truncnorm": (_int, [_vp, _i64, _i64, _vp, _int, _int, C.c_double, C.c_double, _f32, C.POINTER(_vp)]),
    "gs_zgen_wait": (_int, [_vp, _i64, C.POINTER(_vp)]),
    "gs_zgen_release": (_int, [_vp, _i64]),
    "gs_zgen_finish": (_int, [_vp]),
    "gs_zgen_device": (_int, [_vp, _i64, _i64, _vp, _i64, _int, C.c_double, C.c_double, _f32, _vp]),
    "gs_linear_set_resident": (_int, [_int]),
    "gs_zgen_segmented_nbytes": (_int, [_i64, _int, _int, _vp]),
    "gs_zgen_device_segmented": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _int, _int, _vp, _i64, _vp, _vp]),
    "gs_gram_accumulate": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gs_gram_accumulate_prec": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp]),
    "gs_gram_kernel_time": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "gs_ipca_profile_launches": (_int, [_vp, _int]),
    "gs_ipca_launch_profile": (_int, [_vp, _vp, _vp, _vp]),
    "gs_eigh_sym": (_int, [_vp, _vp, _int, _vp, _vp]),
    "gs_eigh_topk": (_int, [_vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp]),
    "gs_cholqr": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "gs_jacobi_small": (_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "gs_eig_tridiag": (_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "gs_gemm_f64": (_int, [_int, _int, _int, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, C.c_double, C.c_double, _vp, _vp,
                            _vp, _vp]),
    "gs_mapping_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _f32, _f32, _f32, _f32, _int, _i64, _vp]),
    "gs_linear_forward": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "gs_project_rows_nbytes": (_int, [_i64, _int, _int, C.POINTER(_i64)]),
    "gs_project_rows": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _int, _vp, _vp, _i64, _vp, _i64, _vp]),
    "gs_blocked_nbytes": (_int, [_i64, _i64, C.POINTER(_i64)]),
    "gs_block_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "gs_im2col3x3_blocked": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp]),
    "gs_gemm_blocked_nt": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _i64, _vp]),
    "gs_modconv3x3_patches": (_int, [_vp, _i64, _int, _int, _int, _vp, _int, _vp, _vp]),
    "gs_gemm_blocked_nt_styled": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _i64, _int, _vp, _vp, _f32, _vp, _f32, _f32, _int, _vp]),
}

_lib = None


class GanspaceHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ganspace_hip error {code}: {msg}")
        self.code = code


def load():
    """Load (once) and return the ctypes library; raise loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: its wheel bundles the HIP runtime under the same SONAME
    # (libamdhip64.so.7) as /opt/rocm, and streams / device pointers are only interchangeable
    # when both sides resolve to ONE runtime instance - the one torch already loaded.
    import torch  # noqa: F401
    path = os.environ.get("GANSPACE_HIP_LIB", _build.lib_path())
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m ganspace_amd._build`.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    if lib.gs_version() != 1:
        raise RuntimeError(f"ABI version mismatch: library reports {lib.gs_version()}")
    _lib = lib
    return lib


def check(rc: int):
    if rc != GS_OK:
        msg = load().gs_last_error()
        raise GanspaceHipError(rc, msg.decode() if msg else "")
    return rc


def current_stream_ptr():
    """hipStream_t of torch's current stream (0 = null stream if torch has no GPU)."""
    import torch
    if torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(0)
