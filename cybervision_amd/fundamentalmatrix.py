"""FundamentalMatrix (src/fundamentalmatrix.rs:63-257) over the C ABI: RANSAC hypothesis generation,
validate_f, scoring and best-pick all run on the device; the final LM refit of optimize_result is the
library's host arithmetic, as in the reference.  No model arithmetic happens in this module - every
function forwards to libcvhip.so (the numpy restatement that cross-checks the device generator is test
infrastructure and lives with the CPU checker, not here)."""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib

RANSAC_T_AFFINE = 0.1                 # fundamentalmatrix.rs:22
RANSAC_T_PERSPECTIVE = 10.0 / 1000.0  # fundamentalmatrix.rs:23 (times max_dimension)


PENCIL_THIN_SVD, PENCIL_NULL_SPACE = 0, 1  # cvhip_ransac_set_pencil (include/cvhip.h); default: the reference's thin-SVD rows


def set_pencil(device, pencil: int):
    """cvhip_ransac_set_pencil: the basis of the 7-point pencil (fundamentalmatrix.rs:309-322) for this device handle."""
    _lib.check(_lib.lib().cvhip_ransac_set_pencil(device.handle, int(pencil)), "cvhip_ransac_set_pencil")


def set_lm_pipeline(device, enable):
    """cvhip_ransac_set_lm_pipeline (test hook): validate_f's LM as two passes on refilled lanes (2 or True: the default), as
    two passes with a root per thread (1) or as the scalar loop in one kernel (0 or False)."""
    mode = 2 if enable is True else int(enable)
    _lib.check(_lib.lib().cvhip_ransac_set_lm_pipeline(device.handle, mode), "cvhip_ransac_set_lm_pipeline")


def set_count_mfma(device, enable: bool):
    """cvhip_ransac_set_count_mfma: the counting screen's head as f32 matrix products (exact, measured slower: off by default)."""
    _lib.check(_lib.lib().cvhip_ransac_set_count_mfma(device.handle, int(bool(enable))), "cvhip_ransac_set_count_mfma")


def set_in_order(device, enable: bool):
    """cvhip_ransac_set_in_order (test hook): score the batches of rounds in order behind their events, without polling."""
    _lib.check(_lib.lib().cvhip_ransac_set_in_order(device.handle, int(bool(enable))), "cvhip_ransac_set_in_order")


class ProjectionMode(IntEnum):  # fundamentalmatrix.rs:35-39
    Affine = 0
    Perspective = 1


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _matches(matches):
    return np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))


def ransac_score(device, F, matches, t: float):
    """The all-matches fold of validate_f (:210-216) for H hypotheses.  F: [H, 3, 3] or [H, 9] row-major;
    matches: [N, 4] (x1, y1, x2, y2).  -> (count[H] uint32, err_sum[H] float64)."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(-1, 9))
    m = _matches(matches)
    H, N = F.shape[0], m.shape[0]
    cnt = np.zeros(H, dtype=np.uint32)
    err = np.zeros(H, dtype=np.float64)
    _lib.check(_lib.lib().cvhip_ransac_score(device.handle, _p(F), H, _p(m), N, float(t), _p(cnt), _p(err)),
               "cvhip_ransac_score")
    return cnt, err


def ransac_round_score(device, F, matches, t: float):
    """cvhip_ransac_round_score: one RANSAC round's scoring of caller-given hypotheses (counting kernel + ordered sums
    of the hypotheses tied at the maximum) -> (count[H] uint32, err_sum[H] float64, 0 where not computed)."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(-1, 9))
    m = _matches(matches)
    cnt = np.zeros(F.shape[0], dtype=np.uint32)
    err = np.zeros(F.shape[0], dtype=np.float64)
    _lib.check(_lib.lib().cvhip_ransac_round_score(device.handle, _p(F), F.shape[0], _p(m), m.shape[0], float(t), _p(cnt), _p(err)),
               "cvhip_ransac_round_score")
    return cnt, err


def ransac_rounds_pick(device, F, rounds: int, matches, t: float, min_count: int):
    """cvhip_ransac_rounds_pick: the device loops' rounds (pruned counting on the re-sorted list, candidates, finish
    kernel) over `rounds` consecutive slices of caller-given hypotheses.
    -> (index in F or -1, F_best [3, 3], count, mean error or NaN)."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(-1, 9))
    m = _matches(matches)
    out_F = np.zeros(9, dtype=np.float64)
    cnt, err, idx = C.c_uint32(0), C.c_double(0.0), C.c_int64(-1)
    _lib.check(_lib.lib().cvhip_ransac_rounds_pick(device.handle, _p(F), F.shape[0], int(rounds), _p(m), m.shape[0], float(t),
                                                   int(min_count), _p(out_F), C.byref(cnt), C.byref(err), C.byref(idx)),
               "cvhip_ransac_rounds_pick")
    return int(idx.value), out_F.reshape(3, 3), int(cnt.value), float(err.value)


def find_ransac_affine(device, matches, seed: int = 0):
    """FundamentalMatrix::new(Affine, _).find_ransac(matches) entirely on the device
    (cvhip_ransac_affine).  -> (F[3, 3] float64, inlier_mask[N] bool).  Raises CvhipError (code -5) with
    the reference's RansacError text when no model is found."""
    m = _matches(matches)
    N = m.shape[0]
    F = np.zeros(9, dtype=np.float64)
    mask = np.zeros(max(N, 1), dtype=np.uint8)
    cnt = C.c_uint32(0)
    _lib.check(_lib.lib().cvhip_ransac_affine(device.handle, _p(m), N, seed, _p(F), C.byref(cnt), _p(mask)),
               "cvhip_ransac_affine")
    return F.reshape(3, 3), mask[:N].astype(bool)


def optimize_perspective_f(F, inliers):
    """optimize_perspective_f (:391-426) through cvhip_optimize_perspective_f: the reference's own
    Levenberg-Marquardt loop (:515-621) and Jacobian (:473-512) on the 7 free parameters, then its rank test;
    None where the reference returns None.  Host arithmetic inside libcvhip, no device involved."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    m = _matches(inliers)
    out = np.zeros(9, dtype=np.float64)
    refined = C.c_int(0)
    _lib.check(_lib.lib().cvhip_optimize_perspective_f(_p(F), _p(m), len(m), _p(out), C.byref(refined)),
               "cvhip_optimize_perspective_f")
    return out.reshape(3, 3) if refined.value else None


def optimize_perspective_f_device(device, F, inliers):
    """The same refit on the device (cvhip_optimize_perspective_f_device) - what cvhip_find_ransac uses; equal to
    optimize_perspective_f bit for bit."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    m = _matches(inliers)
    out = np.zeros(9, dtype=np.float64)
    refined = C.c_int(0)
    _lib.check(_lib.lib().cvhip_optimize_perspective_f_device(device.handle, _p(F), _p(m), len(m), _p(out), C.byref(refined)),
               "cvhip_optimize_perspective_f_device")
    return out.reshape(3, 3) if refined.value else None


def perspective_models_device(device, matches, sample_idx, t: float):
    """cvhip_ransac_perspective_models: the device generator + validate_f's per-hypothesis checks on caller-chosen
    samples [B, 7] -> F [B, 3, 3, 3] (NaN where a root does not exist or is rejected)."""
    m = _matches(matches)
    idx = np.ascontiguousarray(np.asarray(sample_idx, dtype=np.uint32).reshape(-1, 7))
    out = np.zeros((len(idx), 3, 3, 3), dtype=np.float64)
    _lib.check(_lib.lib().cvhip_ransac_perspective_models(device.handle, _p(m), len(m), _p(idx), len(idx), float(t), _p(out)),
               "cvhip_ransac_perspective_models")
    return out


def affine_models_device(device, matches, sample_idx, t: float = RANSAC_T_AFFINE):
    """cvhip_ransac_affine_models: calculate_model_affine + validate_f's checks on caller-chosen samples [B, 4]
    -> F [B, 3, 3] (NaN where the sample is rejected)."""
    m = _matches(matches)
    idx = np.ascontiguousarray(np.asarray(sample_idx, dtype=np.uint32).reshape(-1, 4))
    out = np.zeros((len(idx), 3, 3), dtype=np.float64)
    _lib.check(_lib.lib().cvhip_ransac_affine_models(device.handle, _p(m), len(m), _p(idx), len(idx), float(t), _p(out)),
               "cvhip_ransac_affine_models")
    return out


def find_ransac_perspective_device(device, matches, max_dimension: float, seed: int = 0, rounds: int = 0, refit: bool = True):
    """The RANSAC loop of the perspective model on the device (cvhip_ransac_perspective: `rounds` rounds of 50 000
    samples, 0 = the reference's 20); refit=True adds optimize_result's LM refit and inlier re-selection
    (:246-256) through cvhip_optimize_perspective_f + cvhip_ransac_score - with rounds = 0 that is exactly what
    FundamentalMatrix.find_ransac below does in one call.
    -> (F [3, 3], inlier_mask [N] bool).  Raises CvhipError (code -5) with the reference's messages."""
    m = _matches(matches)
    N = len(m)
    F = np.zeros(9, dtype=np.float64)
    mask = np.zeros(max(N, 1), dtype=np.uint8)
    cnt = C.c_uint32(0)
    _lib.check(_lib.lib().cvhip_ransac_perspective(device.handle, _p(m), N, float(max_dimension), seed, rounds, _p(F),
                                                   C.byref(cnt), _p(mask)), "cvhip_ransac_perspective")
    F = F.reshape(3, 3)
    mask = mask[:N].astype(bool)
    if refit:
        Fo = optimize_perspective_f(F, m[mask])
        if Fo is not None:
            F = Fo
            mask = inlier_mask(device, F, m, RANSAC_T_PERSPECTIVE * float(max_dimension))
    return F, mask


def inlier_mask(device, F, matches, t: float):
    """fits_model (:452-458) of one F for every match (the inlier filter of optimize_result, :233-236, 248-254)
    on the device -> bool [N]."""
    m = _matches(matches)
    out = np.zeros(max(len(m), 1), dtype=np.uint8)
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    _lib.check(_lib.lib().cvhip_fits_model(device.handle, _p(F), _p(m), len(m), float(t), _p(out)), "cvhip_fits_model")
    return out[:len(m)].astype(bool)


class FundamentalMatrix:
    """FundamentalMatrix::new(projection, max_dimension) (:72-101); find_ransac (:103-147) -> (F, inlier mask)."""

    def __init__(self, projection: ProjectionMode, max_dimension: float):
        self.projection = ProjectionMode(projection)
        self.max_dimension = float(max_dimension)
        self.ransac_t = RANSAC_T_AFFINE if self.projection == ProjectionMode.Affine else RANSAC_T_PERSPECTIVE * self.max_dimension

    def find_ransac(self, device, point_matches, seed: int = 0, progress_listener=None):
        """-> (F [3, 3] float64, inliers [n, 4] uint32, inlier_mask [N] bool).  Raises CvhipError (code -5) with the
        reference's RansacError messages ("Not enough matches", "No reliable matches found").  progress_listener:
        the reference's `Option<&PL>` (:41-47, 103) - an object with report_status(pos) and report_matches(count)."""
        m = _matches(point_matches)
        N = len(m)
        F = np.zeros(9, dtype=np.float64)
        mask = np.zeros(max(N, 1), dtype=np.uint8)
        cnt = C.c_uint32(0)
        pl = progress_listener
        cb_s = _lib.PROGRESS_FN(lambda _u, v: pl.report_status(v)) if pl is not None else _lib.NULL_PROGRESS
        cb_m = _lib.MATCHES_FN(lambda _u, c: pl.report_matches(int(c))) if pl is not None else _lib.NULL_MATCHES
        _lib.check(_lib.lib().cvhip_find_ransac(device.handle, int(self.projection), _p(m), N, self.max_dimension, seed,
                                                _p(F), C.byref(cnt), _p(mask), cb_s, cb_m, None), "cvhip_find_ransac")
        mask = mask[:N].astype(bool)
        return F.reshape(3, 3), m[mask], mask
