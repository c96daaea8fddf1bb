"""numpy restatement of the model-fitting half of src/fundamentalmatrix.rs.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (see
oracle/cvref.h).  Nothing under cybervision_amd/ imports it.

What is here: calculate_model_affine (:260-286), calculate_model_perspective (:289-389),
optimize_perspective_f / least_squares / f_jacobian (:391-426, 473-621), validate_f (:192-229),
choose_inliers (:155-175), the Ord of RansacIterationResult (:623-649) and the find_ransac loop
(:103-147) built from them.  Every function follows the cited lines statement by statement, but the linear
algebra underneath is LAPACK's (np.linalg.svd / solve / eigvals) - a DIFFERENT implementation from both the
reference's nalgebra and the device's Householder / Jacobi code.  That is the point: agreement between
this module and libcvhip.so on identical samples is agreement between two independent derivations, to a
stated tolerance (1e-7 relative on F; survivor sets equal except at the rank / sign thresholds).

Parity status: unpinned by the reference (no fixtures, no Rust toolchain - oracle/cvref.h).  The reference's
RNG is OS-seeded, so find_ransac itself is only statistically comparable.
"""
from __future__ import annotations

import numpy as np

TOP_INLIERS = 5000                        # fundamentalmatrix.rs:16
MIN_INLIER_DISTANCE = 10                  # :17
RANSAC_K = 1_000_000                      # :18-19
RANSAC_N_AFFINE, RANSAC_N_PERSPECTIVE = 4, 7          # :20-21
RANSAC_T_AFFINE, RANSAC_T_PERSPECTIVE = 0.1, 10.0 / 1000.0  # :22-23
RANSAC_D_AFFINE, RANSAC_D_PERSPECTIVE = 10, 200       # :24-25
RANSAC_D_EARLY_EXIT_AFFINE, RANSAC_D_EARLY_EXIT_PERSPECTIVE = 1000, 50_000  # :26-27
RANSAC_CHECK_INTERVAL = 50_000            # :28
RANSAC_RANK_EPSILON = 0.001               # :29-30


def _h(m):
    """matches [n, 4] (x1, y1, x2, y2) -> homogeneous p1 [n, 3], p2 [n, 3]."""
    m = np.asarray(m, dtype=np.float64).reshape(-1, 4)
    one = np.ones(len(m))
    return np.stack([m[:, 0], m[:, 1], one], axis=1), np.stack([m[:, 2], m[:, 3], one], axis=1)


def reprojection_error(F, matches):
    """:461-471 for matches [n, 4] -> [n]."""
    F = np.asarray(F, dtype=np.float64).reshape(3, 3)
    p1, p2 = _h(matches)
    f_p1 = p1 @ F.T
    ft_p2 = p2 @ F
    num = np.sum(p2 * f_p1, axis=1) ** 2
    den = f_p1[:, 0] ** 2 + f_p1[:, 1] ** 2 + ft_p2[:, 0] ** 2 + ft_p2[:, 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den


def fits_model(F, matches, t):
    """:452-458 -> bool [n]."""
    e = reprojection_error(F, matches)
    return np.isfinite(e) & ~(np.abs(e) > t)


def calculate_model_affine(sample):
    """:260-286.  sample [4, 4] -> F [3, 3] or None."""
    s = np.asarray(sample, dtype=np.float64).reshape(4, 4)
    a = np.stack([s[:, 2], s[:, 3], s[:, 0], s[:, 1]], axis=1)      # (x2, y2, x1, y1), :262-268
    mean = a.mean(axis=0)
    a = a - mean
    _, sv, vt = np.linalg.svd(a)
    if abs(sv[1]) < RANSAC_RANK_EPSILON:
        return None
    vtc = vt[3]
    e = vtc @ mean
    f = np.array([[0.0, 0.0, vtc[0]], [0.0, 0.0, vtc[1]], [vtc[2], vtc[3], -e]])
    with np.errstate(divide="ignore", invalid="ignore"):
        return f / f[2, 2]


def _real_cubic_roots(c0, c1, c2, c3):
    """roots::find_roots_cubic (roots 0.0.8) returns the real roots; np.roots drops leading zeros like the crate
    falls back to the quadratic / linear formulas."""
    co = np.array([c0, c1, c2, c3])
    if not np.isfinite(co).all():
        return []
    r = np.roots(co)
    return sorted(float(z.real) for z in r if abs(z.imag) <= 1e-9 * max(1.0, abs(z.real)))


PENCIL_THIN_SVD, PENCIL_NULL_SPACE = 0, 1   # cvhip_ransac_set_pencil's modes (include/cvhip.h)


def canonical_sign(v):
    """The sign convention of the thin-SVD pencil, shared with the device generator: a singular vector is only defined
    up to sign, nalgebra's choice falls out of its bidiagonalisation (source not in this image), and the pencil
    `root f1 + (1 - root) f2` - its roots and the SCALE the rank test :365-370 sees - depends on it.  Convention here:
    the entry of largest magnitude (the first of equals) is positive."""
    v = np.asarray(v, dtype=np.float64)
    return -v if v[int(np.argmax(np.abs(v)))] < 0.0 else v


def perspective_pencil(sample, pencil=PENCIL_THIN_SVD):
    """:289-323 -> (f1, f2) as [3, 3].  PENCIL_THIN_SVD (default) is the reference as written: `a.svd(false, true)` on
    an SMatrix<7, 9> gives nalgebra's THIN decomposition - v_t is DimMinimum<7, 9> x 9 = 7 x 9, singular values
    descending - so v_t.row(nrows - 2) and v_t.row(nrows - 1) are rows 5 and 6: the right singular vectors of the
    two SMALLEST of the seven singular values, NOT the two-dimensional null space of A.  PENCIL_NULL_SPACE is the
    7-point algorithm as published (rows 7 and 8 of the full 9 x 9 V'), kept as an option."""
    p1, p2 = _h(sample)
    x1, y1, x2, y2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    A = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones(7)], axis=1)
    if pencil == PENCIL_THIN_SVD:
        _, _, vt = np.linalg.svd(A, full_matrices=False)          # [7, 9]
        return canonical_sign(vt[5]).reshape(3, 3), canonical_sign(vt[6]).reshape(3, 3)
    _, _, vt = np.linalg.svd(A, full_matrices=True)
    return vt[7].reshape(3, 3), vt[8].reshape(3, 3)


def calculate_model_perspective(sample, pencil=PENCIL_THIN_SVD):
    """:289-389.  sample [7, 4] -> list of F [3, 3] (normalised by F[2][2], rank and sign checks applied)."""
    p1, p2 = _h(sample)
    f1, f2 = perspective_pencil(sample, pencil)
    ff = (f1, f2)
    d = np.empty((2, 2, 2))
    for i in range(2):
        for j in range(2):
            for k in range(2):
                d[i, j, k] = np.linalg.det(np.stack([ff[i][:, 0], ff[j][:, 1], ff[k][:, 2]], axis=1))
    c0 = -d[1, 0, 0] + d[0, 1, 1] + d[0, 0, 0] + d[1, 1, 0] + d[1, 0, 1] - d[0, 1, 0] - d[0, 0, 1] - d[1, 1, 1]
    c1 = d[0, 0, 1] - 2.0 * d[0, 1, 1] - 2.0 * d[1, 0, 1] + d[1, 0, 0] - 2.0 * d[1, 1, 0] + d[0, 1, 0] + 3.0 * d[1, 1, 1]
    c2 = d[1, 1, 0] + d[0, 1, 1] + d[1, 0, 1] - 3.0 * d[1, 1, 1]
    c3 = d[1, 1, 1]
    out = []
    for root in _real_cubic_roots(c0, c1, c2, c3):
        f = root * f1 + (1.0 - root) * f2
        _, sv, vt2 = np.linalg.svd(f.T)
        if abs(sv[1]) < RANSAC_RANK_EPSILON or abs(sv[2]) > RANSAC_RANK_EPSILON:
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            f = f / f[2, 2]
        e1 = vt2[2]
        l1 = np.cross(e1[None, :], p1)                    # [e1]x x1, per point
        s = np.sum((p2 @ f.T) * l1, axis=0)               # column_sum(): one total per component
        if (s > 0.0).all() or (s < 0.0).all():
            out.append(f)
    return out


def params_from_perspective_f(F):   # :429-440
    F = np.asarray(F, dtype=np.float64).reshape(3, 3)
    return F.reshape(9)[:7].copy()


def f_from_perspective_params(p):   # :442-449
    with np.errstate(divide="ignore", invalid="ignore"):
        x = -(-p[0] * p[4] + p[6] * p[2] * p[4] + p[3] * p[1] - p[6] * p[1] * p[5]) / (-p[3] * p[2] + p[0] * p[5])
    return np.array([[p[0], p[1], p[2]], [p[3], p[4], p[5]], [p[6], x, 1.0]])


def f_jacobian(F, matches):
    """:473-512 for all matches -> [n, 7].  As written there: c = d = (F p1)_0 + (F p1)_1 + (F' p2)_0 + (F' p2)_1."""
    p1, p2 = _h(matches)
    f_p1 = p1 @ F.T
    ft_p2 = p2 @ F
    c = f_p1[:, 0] + f_p1[:, 1] + ft_p2[:, 0] + ft_p2[:, 1]
    dd = c
    total = np.sum(p2 * f_p1, axis=1)                     # p2' F p1
    J = np.empty((len(p1), 7))
    for i in range(7):
        row, col = divmod(i, 3)
        a = p2[:, row] * p1[:, col]
        x = F[row, col]
        b = total - a * x                                 # p2' (F with F[row][col] = 0) p1
        with np.errstate(divide="ignore", invalid="ignore"):
            J[:, i] = 2.0 * (a * x + b) * (a * dd - b * c * c * x) / (c * c * x * x + dd)
    return J


def least_squares(params, residual_fn, jacobian_fn):
    """:515-621 -> params or None (Err)."""
    residual = residual_fn(params)
    J = jacobian_fn(params)
    g = J.T @ residual
    if abs(g.max()) <= 1e-12:
        return params
    mu = 1e-3 * np.max(np.diag(J.T @ J))
    nu = 2.0
    found = False
    params = params.copy()
    with np.errstate(all="ignore"):
        for _ in range(1000):
            A = J.T @ J + mu * np.eye(len(params))
            try:
                delta = np.linalg.solve(A, g)
            except np.linalg.LinAlgError:
                return None
            if np.linalg.norm(delta) <= 1e-12 * (np.linalg.norm(params) + 1e-12):
                found = True
                break
            new_params = params + delta
            new_residual = residual_fn(new_params)
            rr, nrr = residual @ residual, new_residual @ new_residual
            rho = (rr - nrr) / (delta @ (delta * mu + g))
            if rho > 0.0:
                converged = np.sqrt(rr) - np.sqrt(nrr) < 0.0 * np.sqrt(rr)
                residual, params = new_residual, new_params
                J = jacobian_fn(params)
                g = J.T @ residual
                if converged or abs(g.max()) <= 1e-12:
                    found = True
                    break
                mu *= max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)
                nu = 2.0
            else:
                mu *= nu
                nu *= 2.0
            if np.linalg.norm(residual) <= 1e-12:
                found = True
                break
    return params if found else None


def optimize_perspective_f(F, inliers):
    """:391-426 -> F [3, 3] or None."""
    inliers = np.asarray(inliers, dtype=np.float64).reshape(-1, 4)
    p = least_squares(params_from_perspective_f(F),
                      lambda q: reprojection_error(f_from_perspective_params(q), inliers),
                      lambda q: f_jacobian(f_from_perspective_params(q), inliers))
    if p is None:
        return None
    f = f_from_perspective_params(p)
    if not np.isfinite(f).all():
        return None  # nalgebra's SVD does not converge on non-finite input; the hypothesis dies either way
    sv = np.linalg.svd(f.T, compute_uv=False)
    if abs(sv[1]) < RANSAC_RANK_EPSILON or abs(sv[2]) > RANSAC_RANK_EPSILON:
        return None
    return f


def validate_f(F, sample, matches, t, min_count, perspective):
    """:192-229 -> (F, matches_count, best_error) or None."""
    F = np.asarray(F, dtype=np.float64).reshape(3, 3)
    if not np.isfinite(F).all():
        return None
    if perspective:
        F = optimize_perspective_f(F, sample)
        if F is None:
            return None
    if not fits_model(F, sample, t).all():
        return None
    e = reprojection_error(F, matches)
    ok = np.isfinite(e) & ~(np.abs(e) > t)
    count = int(ok.sum())
    if count < min_count:
        return None
    return F, count, float(e[ok].sum() / count)


def better(a, b):
    """Ord for RansacIterationResult (:623-649): is (count, error) a strictly greater than b?"""
    if a[0] != b[0]:
        return a[0] > b[0]
    fa, fb = np.isfinite(a[1]), np.isfinite(b[1])
    if fa != fb:
        return bool(fa)
    if not fa:
        return False
    return a[1] < b[1]


def choose_inliers(matches, n, rng):
    """:155-175 with numpy's Generator in place of SmallRng -> n match indices."""
    m = np.asarray(matches, dtype=np.int64).reshape(-1, 4)
    limit = min(len(m), TOP_INLIERS)
    idx = []
    while len(idx) < n:
        i = int(rng.integers(0, limit))
        if all((np.abs(m[i] - m[j]) >= MIN_INLIER_DISTANCE).all() for j in idx):
            idx.append(i)
    return np.array(idx, dtype=np.int64)


def ransac_iteration(matches, sample_idx, t, min_count, perspective, pencil=PENCIL_THIN_SVD):
    """:177-190 for one given sample -> list of (F, count, error)."""
    m = np.asarray(matches).reshape(-1, 4)
    sample = m[np.asarray(sample_idx)]
    if perspective:
        models = calculate_model_perspective(sample, pencil)
    else:
        f = calculate_model_affine(sample)
        models = [] if f is None else [f]
    out = []
    for f in models:
        r = validate_f(f, sample, m, t, min_count, perspective)
        if r is not None:
            out.append(r)
    return out


def find_ransac(matches, perspective, max_dimension=0.0, rng=None, iterations=RANSAC_K, check_interval=RANSAC_CHECK_INTERVAL,
                pencil=PENCIL_THIN_SVD):
    """:103-147 + optimize_result (:231-257) -> (F, inlier mask) or raises ValueError with the reference's text."""
    m = np.asarray(matches).reshape(-1, 4)
    n = RANSAC_N_PERSPECTIVE if perspective else RANSAC_N_AFFINE
    d = RANSAC_D_PERSPECTIVE if perspective else RANSAC_D_AFFINE
    t = RANSAC_T_PERSPECTIVE * max_dimension if perspective else RANSAC_T_AFFINE
    early = RANSAC_D_EARLY_EXIT_PERSPECTIVE if perspective else RANSAC_D_EARLY_EXIT_AFFINE
    if len(m) < d + n:
        raise ValueError("Not enough matches")
    rng = rng or np.random.default_rng(0)
    best = None
    for _ in range(max(iterations // check_interval, 1)):
        for _ in range(check_interval):
            for r in ransac_iteration(m, choose_inliers(m, n, rng), t, d + n, perspective, pencil):
                if best is None or better((r[1], r[2]), (best[1], best[2])):
                    best = r
        if best is not None and best[1] > early:
            break
    if best is None:
        raise ValueError("No reliable matches found")
    F = best[0]
    mask = fits_model(F, m, t)
    if perspective:
        Fo = optimize_perspective_f(F, m[mask])
        if Fo is not None:
            F = Fo
        mask = fits_model(F, m, t)
    return F, mask
