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


# ---------------------------------------------------------------------------------------------
# Perspective model: host-side hypothesis generation (numpy), device-side scoring.
# Mirrors FundamentalMatrix::find_ransac for ProjectionMode::Perspective
# (src/fundamentalmatrix.rs:103-147, 155-229, 289-449); the reference's RNG is OS-seeded, so
# parity is statistical - the hypothesis scoring itself is the bit-exact kernel.
# ---------------------------------------------------------------------------------------------
TOP_INLIERS = 5000                   # fundamentalmatrix.rs:16
MIN_INLIER_DISTANCE = 10             # :17
RANSAC_K_PERSPECTIVE = 1_000_000     # :19
RANSAC_N_PERSPECTIVE = 7             # :21
RANSAC_D_PERSPECTIVE = 200           # :25
RANSAC_D_EARLY_EXIT_PERSPECTIVE = 50_000  # :27
RANSAC_CHECK_INTERVAL = 50_000       # :28
RANSAC_RANK_EPSILON_PERSPECTIVE = 0.001   # :30


def reprojection_error(F, matches):
    """fundamentalmatrix.rs:461-471, vectorised over matches [N, 4] (and over F [H, 3, 3] when given)."""
    F = np.asarray(F, dtype=np.float64)
    m = np.asarray(matches, dtype=np.float64).reshape(-1, 4)
    p1 = np.stack([m[:, 0], m[:, 1], np.ones(len(m))], axis=-1)
    p2 = np.stack([m[:, 2], m[:, 3], np.ones(len(m))], axis=-1)
    f_p1 = p1 @ np.swapaxes(F, -1, -2)          # (F p1)
    ft_p2 = p2 @ F                              # (F^T p2)
    num = np.sum(p2 * f_p1, axis=-1) ** 2
    den = f_p1[..., 0] ** 2 + f_p1[..., 1] ** 2 + ft_p2[..., 0] ** 2 + ft_p2[..., 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den


def choose_inliers(matches, count: int, rng):
    """choose_inliers (:155-175) for `count` samples at once: 7 matches from the top min(N, 5000), pairwise at
    least 10 px apart in all four coordinates (rejection sampling, vectorised: a sample with a conflict is
    redrawn whole - same distribution over accepted samples as drawing conflicting members again)."""
    m = np.asarray(matches, dtype=np.int64).reshape(-1, 4)
    limit = min(len(m), TOP_INLIERS)
    out = np.empty((count, RANSAC_N_PERSPECTIVE), dtype=np.int64)
    todo = np.arange(count)
    for _ in range(64):
        if len(todo) == 0:
            break
        idx = rng.integers(0, limit, size=(len(todo), RANSAC_N_PERSPECTIVE))
        pts = m[idx]                                              # [T, 7, 4]
        d = np.abs(pts[:, :, None, :] - pts[:, None, :, :])       # [T, 7, 7, 4]
        close = (d < MIN_INLIER_DISTANCE).any(axis=-1)
        close[:, np.arange(7), np.arange(7)] = False
        ok = ~close.any(axis=(1, 2))
        out[todo[ok]] = idx[ok]
        todo = todo[~ok]
    return out[np.setdiff1d(np.arange(count), todo)]


def calculate_model_perspective(samples):
    """calculate_model_perspective (:289-389) for a batch of 7-match samples [B, 7, 4] -> (F [K, 3, 3],
    sample index [K]): null space of the 7x9 system, det(a F1 + (1-a) F2) = 0 cubic, rank and
    sign-consistency checks, normalised by F[2][2]."""
    s = np.asarray(samples, dtype=np.float64)
    B = len(s)
    x1, y1, x2, y2 = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    one = np.ones_like(x1)
    A = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, one], axis=-1)   # [B, 7, 9]
    _, _, vt = np.linalg.svd(A, full_matrices=True)
    F1 = vt[:, 7, :].reshape(B, 3, 3)
    F2 = vt[:, 8, :].reshape(B, 3, 3)
    FF = np.stack([F1, F2], axis=1)                                                    # [B, 2, 3, 3]
    # d[i][j][k] = det([F_i col 0, F_j col 1, F_k col 2])  (vgg_singF_from_FF)
    d = np.empty((B, 2, 2, 2))
    for i in range(2):
        for j in range(2):
            for k in range(2):
                M = np.stack([FF[:, i, :, 0], FF[:, j, :, 1], FF[:, k, :, 2]], axis=-1)
                d[:, i, j, k] = np.linalg.det(M)
    c0 = (-d[:, 1, 0, 0] + d[:, 0, 1, 1] + d[:, 0, 0, 0] + d[:, 1, 1, 0] + d[:, 1, 0, 1] - d[:, 0, 1, 0]
          - d[:, 0, 0, 1] - d[:, 1, 1, 1])
    c1 = (d[:, 0, 0, 1] - 2.0 * d[:, 0, 1, 1] - 2.0 * d[:, 1, 0, 1] + d[:, 1, 0, 0] - 2.0 * d[:, 1, 1, 0]
          + d[:, 0, 1, 0] + 3.0 * d[:, 1, 1, 1])
    c2 = d[:, 1, 1, 0] + d[:, 0, 1, 1] + d[:, 1, 0, 1] - 3.0 * d[:, 1, 1, 1]
    c3 = d[:, 1, 1, 1]
    # real roots of c0 a^3 + c1 a^2 + c2 a + c3 via the companion matrix (find_roots_cubic, :351)
    good = np.abs(c0) > 1e-300
    comp = np.zeros((B, 3, 3))
    with np.errstate(divide="ignore", invalid="ignore"):
        comp[:, 0, 0] = -c1 / c0
        comp[:, 0, 1] = -c2 / c0
        comp[:, 0, 2] = -c3 / c0
    comp[:, 1, 0] = 1.0
    comp[:, 2, 1] = 1.0
    comp[~good] = 0.0
    ev = np.linalg.eigvals(comp)                                                       # [B, 3] complex
    real = good[:, None] & (np.abs(ev.imag) <= 1e-9 * np.maximum(1.0, np.abs(ev.real)))
    bi, ri = np.nonzero(real)
    a = ev.real[bi, ri]
    F = a[:, None, None] * F1[bi] + (1.0 - a)[:, None, None] * F2[bi]
    u, sv, vt2 = np.linalg.svd(np.swapaxes(F, 1, 2))
    rank_ok = (np.abs(sv[:, 1]) >= RANSAC_RANK_EPSILON_PERSPECTIVE) & (np.abs(sv[:, 2]) <= RANSAC_RANK_EPSILON_PERSPECTIVE)
    with np.errstate(divide="ignore", invalid="ignore"):
        F = F / F[:, 2:3, 2:3]
    # sign consistency (:372-383): e1 = last right singular vector of F^T, l1 = [e1]x x1, s = sum((F x2) * l1)
    e1 = vt2[:, 2, :]
    X1 = np.stack([x1[bi], y1[bi], one[bi]], axis=1)                                   # [K, 3, 7]
    X2 = np.stack([x2[bi], y2[bi], one[bi]], axis=1)
    l1 = np.cross(e1[:, :, None], X1, axisa=1, axisb=1, axisc=1)
    # nalgebra's column_sum() adds the COLUMNS (the seven points) - one total per component
    sgn = np.sum((F @ X2) * l1, axis=2)                                                # [K, 3]
    sign_ok = (sgn > 0.0).all(axis=1) | (sgn < 0.0).all(axis=1)
    keep = rank_ok & sign_ok & np.isfinite(F).all(axis=(1, 2))
    return F[keep], bi[keep]


def f_from_perspective_params(p):
    """:442-449 - det(F) = 0 by construction, 7 degrees of freedom."""
    x = -(-p[0] * p[4] + p[6] * p[2] * p[4] + p[3] * p[1] - p[6] * p[1] * p[5]) / (-p[3] * p[2] + p[0] * p[5])
    return np.array([[p[0], p[1], p[2]], [p[3], p[4], p[5]], [p[6], x, 1.0]])


def optimize_perspective_f(F, inliers):
    """optimize_perspective_f (:391-426) through cvhip_optimize_perspective_f: the reference's own
    Levenberg-Marquardt loop (:515-621) and Jacobian (:473-512) on the 7 free parameters, then its rank test;
    None where the reference returns None.  Host arithmetic inside libcvhip, no device involved."""
    F = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(9))
    m = np.ascontiguousarray(np.asarray(inliers, dtype=np.uint32).reshape(-1, 4))
    out = np.zeros(9, dtype=np.float64)
    refined = C.c_int(0)
    _lib.check(_lib.lib().cvhip_optimize_perspective_f(C.c_void_p(F.ctypes.data), C.c_void_p(m.ctypes.data), len(m),
                                                       C.c_void_p(out.ctypes.data), C.byref(refined)),
               "cvhip_optimize_perspective_f")
    return out.reshape(3, 3) if refined.value else None


def find_ransac_perspective(device, matches, max_dimension: float, seed: int = 0, k: int = RANSAC_K_PERSPECTIVE,
                            check_interval: int = RANSAC_CHECK_INTERVAL):
    """FundamentalMatrix::new(Perspective, max_dimension).find_ransac(matches): rounds of `check_interval`
    7-point hypotheses generated on the host, every surviving root scored against ALL matches on the device
    (cvhip_ransac_score - the fold of validate_f :210-216), best = most inliers then smallest mean error
    (:623-663), early exit above 50 000 inliers, final LM refit on the inliers (optimize_result :231-257).
    The per-hypothesis LM of validate_f (:205) is skipped: a 7-point solution has zero reprojection error on
    its own sample, which is that optimisation's fixed point.
    -> (F [3, 3], inlier_mask [N] bool).  Raises ValueError with the reference's messages."""
    m = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    t = RANSAC_T_PERSPECTIVE * float(max_dimension)
    if len(m) < RANSAC_D_PERSPECTIVE + RANSAC_N_PERSPECTIVE:
        raise ValueError("Not enough matches")
    rng = np.random.default_rng(seed)
    best = None  # (count, mean error, F)
    for _ in range(max(k // check_interval, 1)):
        idx = choose_inliers(m, check_interval, rng)
        F, _ = calculate_model_perspective(m[idx].astype(np.float64))
        if len(F):
            cnt, err = ransac_score(device, F, m, t)
            ok = cnt >= RANSAC_D_PERSPECTIVE + RANSAC_N_PERSPECTIVE
            if ok.any():
                mean = np.where(ok, err / np.maximum(cnt, 1), np.inf)
                order = np.lexsort((mean, -cnt.astype(np.int64)))
                j = order[0]
                cand = (int(cnt[j]), float(mean[j]), F[j])
                if best is None or (cand[0], -cand[1]) > (best[0], -best[1]):
                    best = cand
        if best is not None and best[0] > RANSAC_D_EARLY_EXIT_PERSPECTIVE:
            break
    if best is None:
        raise ValueError("No reliable matches found")
    Fb = best[2]
    err = reprojection_error(Fb, m)
    mask = np.isfinite(err) & (np.abs(err) <= t)
    Fo = optimize_perspective_f(Fb, m[mask])
    if Fo is not None:
        Fb = Fo
        err = reprojection_error(Fb, m)
        mask = np.isfinite(err) & (np.abs(err) <= t)
    return Fb, mask


def perspective_models_device(device, matches, sample_idx, t: float):
    """cvhip_ransac_perspective_models: the device generator on caller-chosen samples [B, 7] -> F [B, 3, 3, 3]
    (NaN where a root does not exist or fails a check)."""
    m = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    idx = np.ascontiguousarray(np.asarray(sample_idx, dtype=np.uint32).reshape(-1, 7))
    out = np.zeros((len(idx), 3, 3, 3), dtype=np.float64)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(_lib.lib().cvhip_ransac_perspective_models(device.handle, p(m), len(m), p(idx), len(idx), float(t), p(out)),
               "cvhip_ransac_perspective_models")
    return out


def find_ransac_perspective_device(device, matches, max_dimension: float, seed: int = 0, rounds: int = 0, refit: bool = True):
    """find_ransac for the perspective model with hypothesis generation, scoring and best-pick all on the device
    (cvhip_ransac_perspective); the final LM refit of optimize_result (:246-256) runs here on the host.
    -> (F [3, 3], inlier_mask [N] bool).  Raises CvhipError (code -5) with the reference's messages."""
    m = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 4))
    N = len(m)
    F = np.zeros(9, dtype=np.float64)
    mask = np.zeros(max(N, 1), dtype=np.uint8)
    cnt = C.c_uint32(0)
    _lib.check(_lib.lib().cvhip_ransac_perspective(device.handle, C.c_void_p(m.ctypes.data), N, float(max_dimension), seed,
                                                   rounds, C.c_void_p(F.ctypes.data), C.byref(cnt), C.c_void_p(mask.ctypes.data)),
               "cvhip_ransac_perspective")
    F = F.reshape(3, 3)
    mask = mask[:N].astype(bool)
    if refit:
        Fo = optimize_perspective_f(F, m[mask])
        if Fo is not None:
            t = RANSAC_T_PERSPECTIVE * float(max_dimension)
            err = reprojection_error(Fo, m)
            F, mask = Fo, np.isfinite(err) & (np.abs(err) <= t)
    return F, mask
