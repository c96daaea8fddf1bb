"""ctypes front-end of libcvref.so — the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/cvref.h).  Nothing under cybervision_amd/ imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> Path:
    so = _HERE / "libcvref.so"
    srcs = [_HERE / n for n in ("cvref_corr.c", "cvref_orb.c", "cvref_ransac.c", "cvref.h", "cvref_orb_pattern.h")]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "libcvref.so"], stdout=subprocess.DEVNULL)
    return so


def _load(so: Path):
    if True:
        if True:
            L = C.CDLL(str(so))
        L.cvref_corr_new.restype = C.c_void_p
        L.cvref_corr_new.argtypes = [C.c_uint32] * 4 + [_f64p, C.c_int, C.c_int]
        L.cvref_corr_free.argtypes = [C.c_void_p]
        L.cvref_corr_correlate_images.restype = C.c_int
        L.cvref_corr_correlate_images.argtypes = [C.c_void_p, _u8p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32,
                                                  C.c_uint32, C.c_float]
        L.cvref_corr_step.restype = C.c_int
        L.cvref_corr_step.argtypes = [C.c_void_p, _u8p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.c_uint32,
                                      C.c_float, C.c_int]
        L.cvref_corr_cross_check.restype = C.c_int
        L.cvref_corr_cross_check.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.cvref_corr_end_level.argtypes = [C.c_void_p]
        L.cvref_corr_get.argtypes = [C.c_void_p, C.c_int, _i32p, _f32p]
        L.cvref_corr_candidates.restype = C.c_uint64
        L.cvref_corr_candidates.argtypes = [C.c_void_p]
        L.cvref_corr_optimal_scale_steps.restype = C.c_uint32
        L.cvref_corr_optimal_scale_steps.argtypes = [C.c_uint32, C.c_uint32]
        L.cvref_image_point_data.argtypes = [_u8p, C.c_uint32, C.c_uint32, _f32p, _f32p, C.c_int]
        L.cvref_orb_extract.restype = C.c_uint32
        L.cvref_orb_extract.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p]
        L.cvref_orb_adjust_contrast.argtypes = [_u8p, C.c_uint32]
        L.cvref_orb_fast.restype = C.c_uint32
        L.cvref_orb_fast.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u8p]
        L.cvref_orb_harris.restype = C.c_int
        L.cvref_orb_harris.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.cvref_orb_gaussian_kernel.argtypes = [C.c_uint32, _f64p]
        L.cvref_orb_gaussian_blur.argtypes = [_u8p, C.c_uint32, C.c_uint32, _f64p]
        L.cvref_orb_optimal_scale_steps.restype = C.c_uint32
        L.cvref_orb_optimal_scale_steps.argtypes = [C.c_uint32, C.c_uint32]
        L.cvref_match_points.restype = C.c_uint32
        L.cvref_match_points.argtypes = [_u32p, _u32p, C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_uint32, _u32p,
                                         _u32p]
        L.cvref_triangulate_affine.restype = C.c_uint64
        L.cvref_triangulate_affine.argtypes = [_i32p, C.c_uint32, C.c_uint32, _f64p, _u32p]
        L.cvref_extend_tracks.restype = C.c_uint64
        L.cvref_extend_tracks.argtypes = [_i32p, C.c_uint32, C.c_uint32, _i32p, C.c_uint64, C.c_uint32, _i32p, _u32p, _u32p]
        L.cvref_reprojection_error.restype = C.c_double
        L.cvref_reprojection_error.argtypes = [_f64p, _u32p]
        L.cvref_ransac_score.argtypes = [_f64p, C.c_uint32, _u32p, C.c_uint32, C.c_double, _u32p, _f64p]
        L.cvref_optimize_perspective_f.restype = C.c_int
        L.cvref_optimize_perspective_f.argtypes = [_f64p, _u32p, C.c_uint32, _f64p]
    return L


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libcvref.so"
        if not so.exists():
            build()
        _LIB = _load(so)
    return _LIB


_ALT = None


def alt_lib():
    """The sensitivity build (-DCVREF_ALT_ASSOC: the other association of nalgebra's three-term F*p / dot products)."""
    global _ALT
    if _ALT is None:
        subprocess.check_call(["make", "-C", str(_HERE), "libcvref_alt.so"], stdout=subprocess.DEVNULL)
        _ALT = _load(_HERE / "libcvref_alt.so")
    return _ALT


def default_threads() -> int:
    return max(1, min(os.cpu_count() or 1, 64))


class Corr:
    """PointCorrelations, CPU branch (src/correlation/mod.rs:150-245)."""

    def __init__(self, dims1, dims2, F, projection: int = 0, nthreads: int | None = None, alt: bool = False):
        self.w1, self.h1 = dims1
        self.w2, self.h2 = dims2
        self.nthreads = nthreads or default_threads()
        self._lib = alt_lib() if alt else lib()
        F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
        self._h = self._lib.cvref_corr_new(self.w1, self.h1, self.w2, self.h2, F, projection, self.nthreads)
        if not self._h:
            raise MemoryError("cvref_corr_new")

    def close(self):
        if self._h:
            self._lib.cvref_corr_free(self._h)
            self._h = None

    __del__ = close

    def correlate_images(self, img1, img2, scale: float):
        img1 = np.ascontiguousarray(img1, dtype=np.uint8)
        img2 = np.ascontiguousarray(img2, dtype=np.uint8)
        rc = self._lib.cvref_corr_correlate_images(self._h, img1, img1.shape[1], img1.shape[0], img2, img2.shape[1],
                                               img2.shape[0], scale)
        if rc:
            raise RuntimeError(f"cvref_corr_correlate_images rc={rc}")

    def step(self, img1, img2, scale: float, direction: int):
        img1 = np.ascontiguousarray(img1, dtype=np.uint8)
        img2 = np.ascontiguousarray(img2, dtype=np.uint8)
        rc = self._lib.cvref_corr_step(self._h, img1, img1.shape[1], img1.shape[0], img2, img2.shape[1], img2.shape[0],
                                   scale, direction)
        if rc:
            raise RuntimeError(f"cvref_corr_step rc={rc}")

    def cross_check(self, scale: float, direction: int):
        self._lib.cvref_corr_cross_check(self._h, scale, direction)

    def end_level(self):
        self._lib.cvref_corr_end_level(self._h)

    def get(self, direction: int = 0):
        w, h = (self.w1, self.h1) if direction == 0 else (self.w2, self.h2)
        xy = np.empty((h, w, 2), dtype=np.int32)
        corr = np.empty((h, w), dtype=np.float32)
        self._lib.cvref_corr_get(self._h, direction, xy, corr)
        return xy, corr

    @property
    def candidates(self) -> int:
        return int(self._lib.cvref_corr_candidates(self._h))


def correlate_dense(pyr1, pyr2, F, projection: int = 0, nthreads: int | None = None, alt: bool = False):
    """The level loop of correlate_dense (src/reconstruction.rs:554-588) over prebuilt pyramids
    (pyr[k] is the 1/2^k image).  Returns (xy, corr, candidates) for the forward grid."""
    steps = len(pyr1) - 1
    h1, w1 = pyr1[0].shape
    h2, w2 = pyr2[0].shape
    c = Corr((w1, h1), (w2, h2), F, projection, nthreads, alt)
    try:
        for i in range(steps + 1):
            k = steps - i
            c.correlate_images(pyr1[k], pyr2[k], 1.0 / float(1 << k))
        xy, corr = c.get(0)
        return xy, corr, c.candidates
    finally:
        c.close()


def image_point_data(img, nthreads: int = 1):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    avg = np.empty(img.shape, dtype=np.float32)
    std = np.empty(img.shape, dtype=np.float32)
    lib().cvref_image_point_data(img, img.shape[1], img.shape[0], avg, std, nthreads)
    return avg, std


def orb_extract(img, cap: int = 10000):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    xy = np.zeros((cap, 2), dtype=np.uint32)
    desc = np.zeros((cap, 8), dtype=np.uint32)
    n = lib().cvref_orb_extract(img, img.shape[1], img.shape[0], cap, xy, desc)
    return xy[:n].copy(), desc[:n].copy()


def orb_fast(img, cap: int = 1 << 22):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    xy = np.zeros((cap, 2), dtype=np.uint32)
    sc = np.zeros(cap, dtype=np.uint8)
    n = lib().cvref_orb_fast(img, img.shape[1], img.shape[0], cap, xy, sc)
    assert n <= cap
    return xy[:n].copy(), sc[:n].copy()


def orb_adjust_contrast(img):
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    lib().cvref_orb_adjust_contrast(out.reshape(-1), out.size)
    return out


def orb_harris(img, x: int, y: int):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = C.c_double(0.0)
    ok = lib().cvref_orb_harris(img, img.shape[1], img.shape[0], x, y, C.byref(out))
    return out.value if ok else None


def orb_gaussian_kernel(width: int):
    k = np.zeros(width, dtype=np.float64)
    lib().cvref_orb_gaussian_kernel(width, k)
    return k


def orb_gaussian_blur(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    w = img.shape[1]
    out = np.empty((w, w), dtype=np.float64)
    lib().cvref_orb_gaussian_blur(img, w, img.shape[0], out)
    return out


def match_points(xy1, desc1, xy2, desc2, threshold: int):
    xy1 = np.ascontiguousarray(xy1, dtype=np.uint32)
    xy2 = np.ascontiguousarray(xy2, dtype=np.uint32)
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint32)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint32)
    n1 = xy1.shape[0]
    out = np.zeros((max(n1, 1), 4), dtype=np.uint32)
    dist = np.zeros(max(n1, 1), dtype=np.uint32)
    n = lib().cvref_match_points(xy1, desc1, n1, xy2, desc2, xy2.shape[0], threshold, out, dist)
    return out[:n].copy(), dist[:n].copy()


def reprojection_error(F, m) -> float:
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    m = np.ascontiguousarray(np.asarray(m, dtype=np.uint32).reshape(4))
    return float(lib().cvref_reprojection_error(F, m))


def ransac_score(F, matches, t: float):
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(-1, 9))
    matches = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    H, N = F.shape[0], matches.shape[0]
    cnt = np.zeros(H, dtype=np.uint32)
    err = np.zeros(H, dtype=np.float64)
    lib().cvref_ransac_score(F, H, matches, N, t, cnt, err)
    return cnt, err


def optimize_perspective_f(F, matches):
    """optimize_perspective_f (fundamentalmatrix.rs:391-426) -> F [3, 3], or None where the reference returns None."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    matches = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    out = np.zeros(9, dtype=np.float64)
    ok = lib().cvref_optimize_perspective_f(F, matches, matches.shape[0], out)
    return out.reshape(3, 3) if ok else None


def triangulate_affine(xy):
    """AffineTriangulation::triangulate (triangulation.rs:268-330) -> (points3d[n, 3] f64, p2[n, 2] u32)."""
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    h, w = xy.shape[:2]
    cap = int((xy[..., 0] >= 0).sum())
    pts = np.zeros((max(cap, 1), 3), dtype=np.float64)
    p2 = np.zeros((max(cap, 1), 2), dtype=np.uint32)
    n = lib().cvref_triangulate_affine(xy, w, h, pts, p2)
    assert n == cap
    return pts[:n].copy(), p2[:n].copy()


def extend_tracks(xy, track_p1, max_dimension2: int):
    """Triangulation::extend_tracks (triangulation.rs:1330-1419) -> (track_p2[n_tracks, 2] int32 with -1 = nothing to
    add, new_p1[n, 2] uint32, new_p2[n, 2] uint32)."""
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    h, w = xy.shape[:2]
    track_p1 = np.ascontiguousarray(np.asarray(track_p1, dtype=np.int32).reshape(-1, 2))
    cap = max(int((xy[..., 0] >= 0).sum()), 1)
    tp2 = np.full((max(len(track_p1), 1), 2), -1, dtype=np.int32)
    n1 = np.zeros((cap, 2), dtype=np.uint32)
    n2 = np.zeros((cap, 2), dtype=np.uint32)
    n = lib().cvref_extend_tracks(xy, w, h, track_p1 if len(track_p1) else np.zeros((1, 2), dtype=np.int32), len(track_p1),
                                  max_dimension2, tp2, n1, n2)
    if n == 2 ** 64 - 1:
        raise IndexError("Index out of bounds")
    return tp2[:len(track_p1)].copy(), n1[:n].copy(), n2[:n].copy()
