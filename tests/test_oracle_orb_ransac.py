"""CPU tests of the ORB, matcher and RANSAC-scoring oracle (oracle/cvref_orb.c, cvref_ransac.c):
hand-made known-answer cases and independent numpy re-derivations."""
import math

import numpy as np

from cybervision_amd import synth


def corner_image():
    """Isolated bright pixels on a dark background: each is a FAST corner (its whole ring is
    darker), nothing else is (a ring never holds more than one bright pixel)."""
    img = np.full((64, 64), 30, dtype=np.uint8)
    for (x, y), v in {(20, 20): 220, (43, 20): 200, (20, 43): 180, (43, 43): 160, (10, 50): 44}.items():
        img[y, x] = v
    return img


def test_adjust_contrast(oracle):
    img = np.array([[10, 20], [30, 110]], dtype=np.uint8)
    out = oracle.orb_adjust_contrast(img)
    coeff = np.float32(255.0) / np.float32(100.0)
    want = np.round(coeff * (img.astype(np.float32) - 10)).astype(np.uint8)
    assert (out == want).all() and out.min() == 0 and out.max() == 255
    flat = np.full((4, 4), 9, dtype=np.uint8)
    assert (oracle.orb_adjust_contrast(flat) == flat).all()  # min >= max: untouched (orb.rs:464-466)


def test_fast_finds_isolated_corners(oracle):
    xy, score = oracle.orb_fast(corner_image())
    # contrast 14 (44 vs 30) is below FAST_THRESHOLD = 15 (c < v - t needs 30 < 44 - 15): not a corner
    assert xy.tolist() == [[20, 20], [43, 20], [20, 43], [43, 43]]  # scan order (y, then x)
    # score = largest t in [15, 254] with is_keypoint(t) (bisection, orb.rs:122-133): ring darker than
    # v - t  <=>  30 < v - t  <=>  t <= v - 31
    assert score.tolist() == [220 - 31, 200 - 31, 180 - 31, 160 - 31]


def test_fast_nms_drops_equal_neighbours(oracle):
    """QUIRK (orb.rs:149-184): a neighbour with score >= own suppresses, so two adjacent corners of
    equal score remove each other; with different scores only the weaker one goes."""
    img = np.full((32, 32), 30, dtype=np.uint8)
    img[10, 10] = img[10, 11] = 200
    assert len(oracle.orb_fast(img)[0]) == 0
    img[10, 11] = 210
    xy, score = oracle.orb_fast(img)
    assert xy.tolist() == [[11, 10]] and score.tolist() == [210 - 31]


def test_gaussian_kernel(oracle):
    for width in (7, 11):
        k = oracle.orb_gaussian_kernel(width)
        sigma = (width - 1) / 6.0
        want = [math.exp(-((i - width // 2) ** 2) / (2.0 * sigma * sigma)) / (math.sqrt(2.0 * math.pi) * sigma)
                for i in range(width)]
        assert k.tolist() == want


def test_harris_quirk_and_bounds(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)
    assert oracle.orb_harris(img, 5, 20) is None and oracle.orb_harris(img, 20, 5) is None  # < 6 from edge
    assert oracle.orb_harris(img, 43, 20) is not None and oracle.orb_harris(img, 44, 20) is None
    # independent re-derivation with the 7-wide tap indexing (orb.rs:209-225, 254-255)
    kg = oracle.orb_gaussian_kernel(7)
    sx = [-1.0, 0.0, 1.0, -2.0, 0.0, 2.0, -1.0, 0.0, 1.0]
    sy = [-1.0, -2.0, -1.0, 0.0, 0.0, 0.0, 1.0, 2.0, 1.0]
    x, y = 21, 17
    a = b = c = 0.0
    for ky in range(7):
        for kx in range(7):
            px, py = x + kx - 3, y + ky - 3
            dx = dy = 0.0
            for i in range(9):
                v = float(img[py + i // 7 - 3, px + i % 7 - 3])
                dx += sx[i] * v / 255.0
                dy += sy[i] * v / 255.0
            g = kg[kx] * kg[ky]
            a += dx * dx * g
            b += dy * dy * g
            c += dx * dy * g
    want = a * b - c * c - 0.04 * ((a + b) * (a + b))
    assert oracle.orb_harris(img, x, y) == want


def test_gaussian_blur_validity_and_values(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(40, 32), dtype=np.uint8)  # h > w: grid is w x w (orb.rs:293)
    out = oracle.orb_gaussian_blur(img)
    assert out.shape == (32, 32)
    valid = ~np.isnan(out)
    want_valid = np.zeros_like(valid)
    want_valid[10:30, 5:27] = True  # 10 <= y < h-10 (=30) and y < w (=32); 5 <= x < w-5
    assert (valid == want_valid).all()
    kg = oracle.orb_gaussian_kernel(11)
    x, y = 12, 15
    col = []
    for yy in range(y - 5, y + 6):
        s = 0.0
        for i in range(11):
            s += kg[i] * float(img[yy, x + i - 5])
        col.append(s)
    s = 0.0
    for i in range(11):
        s += kg[i] * col[i]
    assert out[y, x] == s


def test_extract_points_on_textured_blocks(oracle):
    a, _, _ = synth.make_pair(256, 256, seed=9)
    img = synth.add_blocks(a, count=60)
    xy, desc = oracle.orb_extract(img)
    assert 50 < len(xy) <= 10000
    assert (xy[:, 0] >= 15).all() and (xy[:, 0] < 256 - 15).all() and (xy[:, 1] >= 15).all()
    assert len({tuple(p) for p in xy.tolist()}) == len(xy)
    assert desc.any() and (desc != 0xFFFFFFFF).any()


def test_matcher_first_minimum_and_stable_sort(oracle):
    d = lambda *bits: [sum(1 << b for b in bits)] + [0] * 7  # noqa: E731
    xy1 = np.array([[1, 1], [2, 2], [3, 3]], dtype=np.uint32)
    desc1 = np.array([d(0, 1, 2), d(), d(*range(32))], dtype=np.uint32)
    xy2 = np.array([[10, 10], [20, 20], [30, 30]], dtype=np.uint32)
    desc2 = np.array([d(0), d(1), d()], dtype=np.uint32)
    m, dist = oracle.match_points(xy1, desc1, xy2, desc2, 32)
    # query 0: distances 2, 2, 3 -> first minimum (train 0); query 1: 1, 1, 0 -> train 2;
    # query 2: 31, 31, 32 -> train 0 (first of the 31s); sorted by distance, stable
    assert m.tolist() == [[2, 2, 30, 30], [1, 1, 10, 10], [3, 3, 10, 10]]
    assert dist.tolist() == [0, 2, 31]
    m, dist = oracle.match_points(xy1, desc1, xy2, desc2, 1)
    assert m.tolist() == [[2, 2, 30, 30]]


def test_reprojection_error_known_answers(oracle):
    F = synth.F_HORIZONTAL  # x2^T F x1 = y1 - y2
    assert oracle.reprojection_error(F, [10, 20, 300, 20]) == 0.0
    # one pixel off the epipolar line: (y1-y2)^2 / (1 + 1) = 0.5
    assert oracle.reprojection_error(F, [10, 20, 300, 21]) == 0.5
    rng = np.random.default_rng(1)
    Fr = rng.normal(size=(3, 3))
    m = [11, 23, 37, 5]
    p1, p2 = np.array([11.0, 23.0, 1.0]), np.array([37.0, 5.0, 1.0])
    want = (p2 @ Fr @ p1) ** 2 / ((Fr @ p1)[0] ** 2 + (Fr @ p1)[1] ** 2 + (Fr.T @ p2)[0] ** 2 + (Fr.T @ p2)[1] ** 2)
    assert math.isclose(oracle.reprojection_error(Fr, m), want, rel_tol=1e-12)


def test_ransac_score_counts(oracle):
    rng = np.random.default_rng(2)
    n = 500
    x1 = rng.integers(0, 1000, size=n)
    y1 = rng.integers(0, 1000, size=n)
    x2 = x1 + rng.integers(-50, 50, size=n)
    y2 = y1.copy()
    y2[::5] += rng.integers(3, 40, size=len(y2[::5]))  # 20 % outliers off the line
    m = np.stack([x1, y1, x2, y2], axis=1).astype(np.uint32)
    cnt, err = oracle.ransac_score(np.stack([synth.F_HORIZONTAL, synth.f_tilt(30.0)]), m, 0.1)
    assert cnt[0] == n - len(y2[::5]) and err[0] == 0.0
    assert cnt[1] < cnt[0]


def test_perspective_seven_point_solver_recovers_exact_geometry(oracle, oracle_fm):
    """The oracle's numpy restatement of calculate_model_perspective (fundamentalmatrix.rs:289-389) with the textbook
    null-space pencil (PENCIL_NULL_SPACE, the optional mode): on exact correspondences one of the cubic's roots is
    the true F (zero reprojection error on every point), it passes the reference's rank and sign checks, and the
    vectorised reprojection error agrees with the C restatement."""
    import cases

    fm = oracle_fm
    m, _, exact, F_true = cases.perspective_matches(n=400, outlier_frac=0.0)
    for k in range(5):
        F = fm.calculate_model_perspective(exact[7 * k:7 * k + 7], fm.PENCIL_NULL_SPACE)
        assert len(F) >= 1
        errs = np.array([np.abs(fm.reprojection_error(Fi, exact)).max() for Fi in F])
        assert errs.min() < 1e-12
        Fb = F[int(np.argmin(errs))]
        assert Fb[2, 2] == 1.0 and np.abs(Fb - F_true).max() < 1e-6 * np.abs(F_true).max()
    got = fm.reprojection_error(F_true, m[:50])
    want = np.array([oracle.reprojection_error(F_true, mm) for mm in m[:50]])
    assert np.allclose(got, want, rtol=1e-6, atol=1e-12)  # the numerator cancels ~10 digits; summation orders differ
    # the 7-parameter form used by the final refit reproduces F and keeps det(F) = 0 (:429-449)
    p = fm.params_from_perspective_f(F_true)
    assert np.abs(fm.f_from_perspective_params(p) - F_true).max() < 1e-9 * np.abs(F_true).max()
    # sampling honours the 10 px separation in all four coordinates (:155-175)
    rng = np.random.default_rng(1)
    for _ in range(50):
        pts = m[fm.choose_inliers(m, 7, rng)].astype(np.int64)
        d = np.abs(pts[:, None, :] - pts[None, :, :])
        d[np.arange(7), np.arange(7)] = 1000
        assert (d >= fm.MIN_INLIER_DISTANCE).all()


def _hestenes_svd(W):
    """One-sided Jacobi on the columns of W [m, n] -> (column norms = singular values, normalised columns = left
    singular vectors of W).  Plain Python; the device generator uses the same method."""
    W = np.array(W, dtype=np.float64)
    n = W.shape[1]
    for _ in range(60):
        rotated = False
        for p in range(n):
            for q in range(p + 1, n):
                al, be, ga = W[:, p] @ W[:, p], W[:, q] @ W[:, q], W[:, p] @ W[:, q]
                if abs(ga) <= 1e-17 * np.sqrt(al * be):
                    continue
                rotated = True
                ze = (be - al) / (2.0 * ga)
                t = np.copysign(1.0, ze) / (abs(ze) + np.sqrt(ze * ze + 1.0))
                c = 1.0 / np.sqrt(t * t + 1.0)
                sn = t * c
                W[:, p], W[:, q] = c * W[:, p] - sn * W[:, q], sn * W[:, p] + c * W[:, q]
        if not rotated:
            break
    sv = np.linalg.norm(W, axis=0)
    return sv, W / sv


def test_reference_pencil_is_rows_5_and_6_of_the_thin_svd(oracle_fm):
    """calculate_model_perspective as the reference WRITES it (:309-322): `a.svd(false, true)` of the 7 x 9 system is
    nalgebra's thin decomposition (v_t: 7 x 9), so rows nrows-2 / nrows-1 are rows 5 and 6 - the right singular
    vectors of the two smallest of the SEVEN singular values.  Pinned here without LAPACK: a plain-Python one-sided
    Jacobi SVD gives the same two vectors (modulo sign) and singular values; orthonormal, canonical sign; they are NOT
    null vectors (|A v| = sigma > 0), so the pencil's members have det 0 but do not fit the sample - validate_f's LM
    (:201-205) starts from there for every root.  The null-space mode differs and does fit."""
    import cases

    fm = oracle_fm
    m, _, exact, _ = cases.perspective_matches(n=400, outlier_frac=0.0)
    for k in range(8):
        sample = m[7 * k:7 * k + 7]
        p1, p2 = fm._h(sample)
        A = np.stack([p2[:, 0] * p1[:, 0], p2[:, 0] * p1[:, 1], p2[:, 0], p2[:, 1] * p1[:, 0], p2[:, 1] * p1[:, 1], p2[:, 1],
                      p1[:, 0], p1[:, 1], np.ones(7)], axis=1)
        f1, f2 = fm.perspective_pencil(sample)                       # default = the reference's
        v1, v2 = f1.reshape(9), f2.reshape(9)
        assert abs(v1 @ v1 - 1) < 1e-12 and abs(v2 @ v2 - 1) < 1e-12 and abs(v1 @ v2) < 1e-9
        assert v1[np.argmax(np.abs(v1))] > 0 and v2[np.argmax(np.abs(v2))] > 0
        s1, s2 = np.linalg.norm(A @ v1), np.linalg.norm(A @ v2)     # = sigma_6, sigma_7
        sv, V = _hestenes_svd(A.T)                                   # independent of LAPACK: one-sided Jacobi
        order = np.argsort(-sv)
        assert s1 >= s2 > 0 and abs(s1 - sv[order[5]]) <= 1e-9 * sv[order[5]] and abs(s2 - sv[order[6]]) <= 1e-9 * sv[order[6]]
        for v, col in ((v1, order[5]), (v2, order[6])):
            assert min(np.abs(v - V[:, col]).max(), np.abs(v + V[:, col]).max()) < 1e-8
        n1, n2 = fm.perspective_pencil(sample, fm.PENCIL_NULL_SPACE)
        assert np.linalg.norm(A @ n1.reshape(9)) < 1e-6 and np.linalg.norm(A @ n2.reshape(9)) < 1e-6
        for F in fm.calculate_model_perspective(sample):
            assert F[2, 2] == 1.0 and abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9
    # on EXACT correspondences A has rank <= 7 with a genuine null space, yet the thin rows still skip it: the textbook
    # mode recovers the geometry (test above), the reference's mode does not
    fits = 0
    for k in range(5):
        for F in fm.calculate_model_perspective(exact[7 * k:7 * k + 7]):
            fits += int(np.abs(fm.reprojection_error(F, exact)).max() < 1e-9)
    assert fits == 0


def test_affine_four_point_model_known_answers(oracle_fm):
    """calculate_model_affine (:260-286): four exact correspondences of a known affine geometry give back its F
    (normalised by F22) to rounding; a degenerate sample (second singular value < 1e-3) gives nothing."""
    fm = oracle_fm
    F_true = synth.f_tilt(12.0)
    th = np.radians(12.0)
    rng = np.random.default_rng(2)
    for _ in range(20):
        x1, y1 = rng.uniform(100, 1900, 4), rng.uniform(100, 1900, 4)
        dist = rng.uniform(-80, 80, 4)
        s = np.stack([x1, y1, x1 + dist * np.cos(th), y1 + dist * np.sin(th)], axis=1)
        F = fm.calculate_model_affine(s)
        assert F is not None and F[2, 2] == 1.0 and (F[:2, :2] == 0).all()
        # same line direction; every point of the generating geometry has zero error
        d = np.array([F[0, 2], F[1, 2]]) / np.hypot(F[0, 2], F[1, 2])
        want = np.array([F_true[0, 2], F_true[1, 2]])
        assert min(np.abs(d - want).max(), np.abs(d + want).max()) < 1e-9
        assert np.abs(fm.reprojection_error(F, s)).max() < 1e-18
    same = np.tile(np.array([[10.0, 20.0, 30.0, 40.0]]), (4, 1))
    assert fm.calculate_model_affine(same) is None
    line = np.stack([np.arange(4.0), np.zeros(4), np.arange(4.0), np.zeros(4)], axis=1) * 1e-4  # rank 1, tiny
    assert fm.calculate_model_affine(line) is None


def test_validate_f_and_result_ordering(oracle_fm):
    """validate_f (:192-229) on the planted perspective geometry: 7-point hypotheses from clean samples survive the LM
    step, the rank test on the re-parametrised matrix and the sample-fit test, and count the planted inliers; a
    non-finite F, a rank-1 F and a sample that does not fit are rejected.  Ord (:623-649): more matches win, then the
    smaller finite error."""
    import cases

    fm = oracle_fm
    m, truth, _, F_true = cases.perspective_matches(n=1500, outlier_frac=0.3)
    t = fm.RANSAC_T_PERSPECTIVE * 2048.0
    rng = np.random.default_rng(5)
    clean = np.flatnonzero(truth[:1000])
    survivors = 0
    for _ in range(30):
        while True:
            idx = rng.choice(clean, 7, replace=False)
            pts = m[idx].astype(np.int64)
            d = np.abs(pts[:, None, :] - pts[None, :, :])
            d[np.arange(7), np.arange(7)] = 1000
            if (d >= 10).all():
                break
        for F in fm.calculate_model_perspective(m[idx], fm.PENCIL_NULL_SPACE):
            r = fm.validate_f(F, m[idx], m, t, 207, True)
            if r is None:
                continue
            survivors += 1
            Fv, cnt, err = r
            assert Fv[2, 2] == 1.0 and abs(np.linalg.det(Fv)) < 1e-9 * np.abs(Fv).max() ** 3 + 1e-18
            assert cnt == int(fm.fits_model(Fv, m, t).sum()) and 0.0 <= err <= t
    assert survivors >= 10
    bad = F_true.copy()
    bad[0, 1] = np.nan
    assert fm.validate_f(bad, m[:7], m, t, 207, True) is None
    tiny = np.array([[1e-9, 2e-9, 1e-7], [3e-9, 1e-9, 2e-7], [1e-7, 2e-7, 1.0]])  # s[1] < 1e-3 after the parameter map
    assert fm.validate_f(tiny, m[:0], m, t, 0, True) is None
    assert fm.validate_f(synth.F_HORIZONTAL, np.array([[10, 10, 50, 400]] * 4), m, 0.1, 0, False) is None  # sample off the line
    assert fm.better((10, 0.5), (9, 0.1)) and not fm.better((9, 0.1), (10, 0.5))
    assert fm.better((10, 0.1), (10, 0.5)) and not fm.better((10, 0.5), (10, 0.5))
    assert fm.better((10, 0.5), (10, np.nan)) and not fm.better((10, np.nan), (10, 0.5)) and not fm.better((10, np.nan), (10, np.inf))


def test_perspective_refit_product_vs_independent_numpy_lm(oracle_fm):
    """cvhip_optimize_perspective_f (host arithmetic inside libcvhip) against the oracle's numpy restatement of the
    same reference lines (:391-426, 473-621) - numpy's summation order and LAPACK's LU instead of the restated
    nalgebra ones, so this is an independent derivation and the comparison is to tolerance, not bits: same
    accept/reject, F within 1e-6 of its largest entry."""
    import cases
    from cybervision_amd import fundamentalmatrix

    m, truth, _, F_true = cases.perspective_matches(n=3000, outlier_frac=0.3)
    inl = m[truth]
    rng = np.random.default_rng(7)
    compared = 0
    for scale in (0.0, 1e-6, 1e-3):
        for n in (7, 40, 1001):
            F0 = F_true * (1.0 + scale * rng.standard_normal((3, 3)))
            want = oracle_fm.optimize_perspective_f(F0, inl[:n])
            got = fundamentalmatrix.optimize_perspective_f(F0, inl[:n])
            assert (want is None) == (got is None), (scale, n)
            if want is not None:
                compared += 1
                assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), (scale, n, np.abs(got - want).max())
    assert compared >= 6


def test_least_squares_first_step_known_answer(oracle_fm):
    """One Levenberg-Marquardt step by hand (:515-573) in exact rational arithmetic on eight integer matches: the
    gradient J'r, the initial damping mu = 1e-3 max diag(J'J) and the step solve((J'J + mu I), J'r) of the numpy
    restatement agree with the exact values to 1e-9 - a pin that does not share code with either implementation."""
    from fractions import Fraction as Q

    m = np.array([[10, 20, 13, 22], [200, 50, 190, 57], [300, 400, 311, 395], [40, 330, 52, 341],
                  [500, 120, 489, 131], [250, 250, 262, 244], [90, 470, 99, 480], [410, 60, 400, 71]], dtype=np.int64)
    p = [Q(1, 1000000), Q(-3, 1000000), Q(2, 1000), Q(4, 1000000), Q(1, 2000000), Q(-7, 1000), Q(-3, 1000)]

    def F_of(q):
        x = -(-q[0] * q[4] + q[6] * q[2] * q[4] + q[3] * q[1] - q[6] * q[1] * q[5]) / (-q[3] * q[2] + q[0] * q[5])
        return [[q[0], q[1], q[2]], [q[3], q[4], q[5]], [q[6], x, Q(1)]]

    def res_and_jac(q):
        F = F_of(q)
        rs, Js = [], []
        for x1, y1, x2, y2 in m.tolist():
            p1, p2 = [Q(x1), Q(y1), Q(1)], [Q(x2), Q(y2), Q(1)]
            fp1 = [sum(F[i][j] * p1[j] for j in range(3)) for i in range(3)]
            ftp2 = [sum(F[i][j] * p2[i] for i in range(3)) for j in range(3)]
            tot = sum(p2[i] * fp1[i] for i in range(3))
            rs.append(tot * tot / (fp1[0] ** 2 + fp1[1] ** 2 + ftp2[0] ** 2 + ftp2[1] ** 2))
            c = fp1[0] + fp1[1] + ftp2[0] + ftp2[1]
            row = []
            for i in range(7):
                r, k = divmod(i, 3)
                a, x = p2[r] * p1[k], F[r][k]
                b = tot - a * x
                row.append(2 * (a * x + b) * (a * c - b * c * c * x) / (c * c * x * x + c))
            Js.append(row)
        return rs, Js

    rs, Js = res_and_jac(p)
    g = [sum(Js[i][j] * rs[i] for i in range(8)) for j in range(7)]
    JtJ = [[sum(Js[i][a] * Js[i][b] for i in range(8)) for b in range(7)] for a in range(7)]
    mu = Q(1, 1000) * max(JtJ[i][i] for i in range(7))
    # exact solve by fraction Gauss-Jordan
    A = [[JtJ[a][b] + (mu if a == b else 0) for b in range(7)] + [g[a]] for a in range(7)]
    for c in range(7):
        piv = next(r for r in range(c, 7) if A[r][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        A[c] = [v / A[c][c] for v in A[c]]
        for r in range(7):
            if r != c and A[r][c] != 0:
                A[r] = [vr - A[r][c] * vc for vr, vc in zip(A[r], A[c])]
    delta = [float(A[r][7]) for r in range(7)]
    fm = oracle_fm
    pf = np.array([float(v) for v in p])
    Jn = fm.f_jacobian(fm.f_from_perspective_params(pf), m)
    rn = fm.reprojection_error(fm.f_from_perspective_params(pf), m)
    assert np.allclose(rn, [float(v) for v in rs], rtol=1e-9, atol=0)
    assert np.allclose(Jn, [[float(v) for v in row] for row in Js], rtol=1e-8, atol=0)
    gn = Jn.T @ rn
    assert np.allclose(gn, [float(v) for v in g], rtol=1e-8, atol=0)
    mun = 1e-3 * np.max(np.diag(Jn.T @ Jn))
    assert abs(mun - float(mu)) <= 1e-9 * float(mu)
    dn = np.linalg.solve(Jn.T @ Jn + mun * np.eye(7), gn)
    assert np.allclose(dn, delta, rtol=1e-6, atol=0)


def test_perspective_refit_product_equals_c_restatement_bitwise(oracle):
    """cvhip_optimize_perspective_f (host arithmetic inside libcvhip: the reference's LM loop, Jacobian, rank test -
    fundamentalmatrix.rs:391-426, 473-621) against oracle/cvref_ransac.c.  The two restate the SAME evaluation order
    (nalgebra's dot / LU as published), so bit equality here guards against transcription slips and compiler
    contraction only - it is not independent evidence (that is test_perspective_refit_product_vs_independent_numpy_lm
    and test_least_squares_first_step_known_answer above)."""
    import cases
    from cybervision_amd import fundamentalmatrix

    m, truth, _, F_true = cases.perspective_matches(n=4000, outlier_frac=0.3)
    inl = m[truth]
    rng = np.random.default_rng(7)
    outcomes = set()
    for scale in (0.0, 1e-6, 1e-3, 0.2):
        for n in (7, 40, 1001, len(inl)):
            F0 = F_true * (1.0 + scale * rng.standard_normal((3, 3)))
            want = oracle.optimize_perspective_f(F0, inl[:n])
            got = fundamentalmatrix.optimize_perspective_f(F0, inl[:n])
            assert (want is None) == (got is None), (scale, n)
            outcomes.add(want is None)
            if want is not None:
                assert (got.view(np.uint64) == want.view(np.uint64)).all(), (scale, n, np.abs(got - want).max())
                # f_from_perspective_params (:442-449): F22 = 1 and det F = 0 by construction
                assert got[2, 2] == 1.0 and abs(np.linalg.det(got)) < 1e-12 * np.abs(got).max() ** 3 + 1e-18
    # the parameter map pins F22 = 1: a matrix whose other entries are tiny is rank 1 after it; with no inliers the
    # loop returns its start (J'r = 0) and the rank test s[1] >= 1e-3 rejects it in both (:418-423)
    tiny = np.array([[1e-9, 2e-9, 1e-7], [3e-9, 1e-9, 2e-7], [1e-7, 2e-7, 1.0]])
    assert oracle.optimize_perspective_f(tiny, inl[:0]) is None
    assert fundamentalmatrix.optimize_perspective_f(tiny, inl[:0]) is None
    assert outcomes == {False}


def test_perspective_refit_is_the_identity_on_an_exact_fit(oracle, oracle_fm):
    """least_squares returns its start when max(J'r) <= 1e-12 (:548-550): on matches that satisfy p2' F p1 = 0 exactly
    the refit only re-expresses F through its 7 parameters (F22 = 1, F21 from det F = 0)."""
    import cases
    from cybervision_amd import fundamentalmatrix

    # F = [e]x-like integer matrix and integer points on its epipolar lines: x2 = x1 + d, y2 = y1 (rectified pair)
    F = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
    rng = np.random.default_rng(3)
    x1, y1, d = rng.integers(50, 900, 60), rng.integers(50, 900, 60), rng.integers(1, 40, 60)
    m = np.stack([x1, y1, x1 + d, y1], axis=1).astype(np.uint32)
    assert max(abs(oracle.reprojection_error(F, mm)) for mm in m) == 0.0
    # the parameter map divides by (-p3 p2 + p0 p5) = 0 here: x = -0/0 = NaN -> every residual is NaN, J'r is NaN,
    # |max(J'r)| <= 1e-12 is false, the solve yields NaN, ... : whatever comes out, both restatements agree
    want, got = oracle.optimize_perspective_f(F, m), fundamentalmatrix.optimize_perspective_f(F, m)
    assert (want is None) == (got is None)
    # a generic exact case: the planted geometry with sub-pixel-exact (unrounded would be exact) points replaced by
    # points that satisfy the integer-rounded F exactly is not constructible; use the 7-point property instead:
    mm, truth, _, _ = cases.perspective_matches(n=400, outlier_frac=0.0, seed=9)
    Fs = oracle_fm.calculate_model_perspective(mm[:7], oracle_fm.PENCIL_NULL_SPACE)
    assert len(Fs) >= 1
    for F7 in Fs:
        got = fundamentalmatrix.optimize_perspective_f(F7, mm[:7])
        want = oracle.optimize_perspective_f(F7, mm[:7])
        assert (want is None) == (got is None)
        if got is not None:
            assert (got.view(np.uint64) == want.view(np.uint64)).all()
            # zero residual on its own sample: the refit moves nothing beyond the re-parametrisation
            assert np.allclose(got, F7 / F7[2, 2], rtol=1e-6, atol=1e-9)
