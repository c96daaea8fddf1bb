"""ctypes binding of libcvhip.so — one Python function per C-ABI entry point (include/cvhip.h).

Fails loudly when the extension has not been built: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libcvhip.so"
_lib = None


class CvhipError(RuntimeError):
    """Non-zero return of a cvhip_* call; mirrors GpuError::Internal (gpu/vulkan.rs:1204-1272)."""

    def __init__(self, code: int, where: str, msg: str):
        super().__init__(f"{where}: {msg} (code {code})")
        self.code = code


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float)
MATCHES_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int)
_vp = C.c_void_p
_u32 = C.c_uint32

# name -> (restype, argtypes); kept in sync with include/cvhip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "cvhip_last_error": (C.c_char_p, []),
    "cvhip_abi_version": (_u32, []),
    "cvhip_device_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "cvhip_device_create_on_stream": (C.c_int, [C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "cvhip_device_destroy": (None, [_vp]),
    "cvhip_device_name": (C.c_char_p, [_vp]),
    "cvhip_device_synchronize": (C.c_int, [_vp]),
    "cvhip_ctx_create": (C.c_int, [_vp, _u32, _u32, _u32, _u32, C.c_int, C.POINTER(C.c_double), C.POINTER(_vp)]),
    "cvhip_ctx_destroy": (None, [_vp]),
    "cvhip_correlate_images": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, C.c_float, C.c_int, C.c_int,
                                         PROGRESS_FN, _vp]),
    "cvhip_cross_check_filter": (C.c_int, [_vp, C.c_float, C.c_int]),
    "cvhip_correlate_level": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, C.c_float, C.c_int, PROGRESS_FN,
                                        _vp]),
    "cvhip_complete": (C.c_int, [_vp, _vp, _vp]),
    "cvhip_complete_dir": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "cvhip_complete_packed": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "cvhip_triangulate_affine": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cvhip_extend_tracks": (C.c_int, [_vp, _vp, C.c_uint64, _u32, _vp, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cvhip_ctx_set_row_shard": (C.c_int, [_vp, _u32, _u32, ALLGATHER_FN, _vp]),
    "cvhip_ctx_set_row_band": (C.c_int, [_vp, _u32, _u32]),
    "cvhip_rccl_unique_id": (C.c_int, [_vp]),
    "cvhip_rccl_create": (C.c_int, [_vp, _vp, _u32, _u32, C.POINTER(_vp)]),
    "cvhip_rccl_destroy": (None, [_vp]),
    "cvhip_rccl_allgather": (C.c_int, [_vp, _vp, C.c_uint64]),
    "cvhip_rccl_gather": (C.c_int, [_vp, _vp, C.c_uint64, _u32]),
    "cvhip_ctx_set_row_shard_rccl": (C.c_int, [_vp, _vp]),
    "cvhip_ctx_gather_bands_rccl": (C.c_int, [_vp, _vp, C.c_int]),
    "cvhip_ctx_level_grid": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_u32),
                                       C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "cvhip_ctx_set_profiling": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cvhip_ctx_get_profile": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                        C.c_int]),
    "cvhip_ctx_get_kernel_times": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_u32), C.c_int]),
    "cvhip_ctx_get_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int]),
    "cvhip_ctx_set_range_mode": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_exact_scores": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_async_readback": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_result_bands": (C.c_int, [_vp, C.c_uint32]),
    "cvhip_ctx_get_result_bands": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "cvhip_ctx_set_search_version": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_borrow_inputs": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_fuse_level_calls": (C.c_int, [_vp, C.c_int]),
    "cvhip_ctx_set_stats_ahead": (C.c_int, [_vp, C.c_int]),
    "cvhip_downsample_box": (C.c_int, [_vp, _vp, _u32, _u32, _vp]),
    "cvhip_resize_lanczos3": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32]),
    "cvhip_orb_extract": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, C.POINTER(_u32), PROGRESS_FN, _vp]),
    "cvhip_orb_extract_batch": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp, PROGRESS_FN, _vp]),
    "cvhip_orb_set_orientation_guard": (C.c_int, [_vp, C.c_double]),
    "cvhip_match_points": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "cvhip_ransac_affine": (C.c_int, [_vp, _vp, _u32, C.c_uint64, _vp, C.POINTER(_u32), _vp]),
    "cvhip_ransac_perspective": (C.c_int, [_vp, _vp, _u32, C.c_double, C.c_uint64, _u32, _vp, C.POINTER(_u32), _vp]),
    "cvhip_ransac_set_pencil": (C.c_int, [_vp, C.c_int]),
    "cvhip_ransac_set_lm_pipeline": (C.c_int, [_vp, C.c_int]),
    "cvhip_ransac_set_count_mfma": (C.c_int, [_vp, C.c_int]),
    "cvhip_ransac_set_in_order": (C.c_int, [_vp, C.c_int]),
    "cvhip_ransac_perspective_models": (C.c_int, [_vp, _vp, _u32, _vp, _u32, C.c_double, _vp]),
    "cvhip_ransac_affine_models": (C.c_int, [_vp, _vp, _u32, _vp, _u32, C.c_double, _vp]),
    "cvhip_fits_model": (C.c_int, [_vp, _vp, _vp, _u32, C.c_double, _vp]),
    "cvhip_ransac_round_score": (C.c_int, [_vp, _vp, _u32, _vp, _u32, C.c_double, _vp, _vp]),
    "cvhip_ransac_rounds_pick": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u32, C.c_double, _u32, _vp, C.POINTER(_u32),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "cvhip_find_ransac": (C.c_int, [_vp, C.c_int, _vp, _u32, C.c_double, C.c_uint64, _vp, C.POINTER(_u32), _vp, PROGRESS_FN,
                                    MATCHES_FN, _vp]),
    "cvhip_optimize_perspective_f": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "cvhip_optimize_perspective_f_device": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "cvhip_ransac_score": (C.c_int, [_vp, _vp, _u32, _vp, _u32, C.c_double, _vp, _vp]),
}


def lib():
    """Load libcvhip.so (once).  Raises if it has not been built — no fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m cybervision_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
        try:
            # torch bundles its own ROCm runtime; loading it FIRST makes libcvhip.so bind to that
            # same libamdhip64.so.7, so one process never holds two HIP runtimes (the second one
            # would see "no HIP GPUs").
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, where: str):
    if rc != 0:
        msg = lib().cvhip_last_error()
        raise CvhipError(rc, where, msg.decode("utf-8", "replace") if msg else "")


NULL_PROGRESS = PROGRESS_FN()
NULL_MATCHES = MATCHES_FN()
NULL_ALLGATHER = ALLGATHER_FN()
