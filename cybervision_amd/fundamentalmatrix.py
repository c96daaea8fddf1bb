"""RANSAC hypothesis scoring (FundamentalMatrix::validate_f's all-matches fold,
src/fundamentalmatrix.rs:210-216, 452-471) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

RANSAC_T_AFFINE = 0.1                 # fundamentalmatrix.rs:22
RANSAC_T_PERSPECTIVE = 10.0 / 1000.0  # fundamentalmatrix.rs:23 (times max_dimension)


def ransac_score(device, F, matches, t: float):
    """F: [H, 3, 3] or [H, 9] row-major; matches: [N, 4] (x1, y1, x2, y2).
    -> (count[H] uint32, err_sum[H] float64)."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(-1, 9))
    matches = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    H, N = F.shape[0], matches.shape[0]
    cnt = np.zeros(H, dtype=np.uint32)
    err = np.zeros(H, dtype=np.float64)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(_lib.lib().cvhip_ransac_score(device.handle, p(F), H, p(matches), N, float(t), p(cnt), p(err)),
               "cvhip_ransac_score")
    return cnt, err


def find_ransac_affine(device, matches, seed: int = 0):
    """FundamentalMatrix::new(Affine, _).find_ransac(matches) entirely on the device
    (cvhip_ransac_affine).  -> (F[3, 3] float64, inlier_mask[N] bool).  Raises CvhipError (code -5) with
    the reference's RansacError text when no model is found."""
    matches = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    N = matches.shape[0]
    F = np.zeros(9, dtype=np.float64)
    mask = np.zeros(max(N, 1), dtype=np.uint8)
    cnt = C.c_uint32(0)
    _lib.check(_lib.lib().cvhip_ransac_affine(device.handle, C.c_void_p(matches.ctypes.data), N, seed,
                                              C.c_void_p(F.ctypes.data), C.byref(cnt), C.c_void_p(mask.ctypes.data)),
               "cvhip_ransac_affine")
    return F.reshape(3, 3), mask[:N].astype(bool)
