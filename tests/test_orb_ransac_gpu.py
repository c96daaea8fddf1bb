"""GPU parity tests for ORB extraction, the keypoint matcher and RANSAC hypothesis scoring:
C-ABI results must equal the CPU oracle's bit for bit (coordinates, descriptors, distances,
inlier counts and error sums)."""
import numpy as np
import pytest

from cybervision_amd import fundamentalmatrix, orb, pointmatching, synth

pytestmark = pytest.mark.gpu


def orb_image(w, h, seed=9, blocks=120):
    a, _, _ = synth.make_pair(w, h, seed=seed)
    return synth.add_blocks(a, count=blocks, seed=seed + 70)


@pytest.mark.parametrize("dims", [(256, 256), (300, 200), (200, 300), (512, 384)])
def test_orb_matches_oracle(gpu_device, oracle, dims):
    img = orb_image(*dims)
    want_xy, want_desc = oracle.orb_extract(img)
    got_xy, got_desc = orb.extract_points(gpu_device, img)
    assert len(want_xy) > 30
    assert got_xy.shape == want_xy.shape, f"{len(got_xy)} keypoints, oracle has {len(want_xy)}"
    assert (got_xy == want_xy).all(), "keypoint coordinates / order differ"
    assert (got_desc == want_desc).all(), "BRIEF descriptors differ"


def test_orb_keypoint_cap_and_ties(gpu_device, oracle):
    """More than MAX_KEYPOINTS corners: the Harris-ranked top 10 000 survive (orb.rs:76-81)."""
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, size=(700, 900), dtype=np.uint8)  # white noise: corners everywhere
    want_xy, want_desc = oracle.orb_extract(img)
    got_xy, got_desc = orb.extract_points(gpu_device, img)
    assert 5000 < len(want_xy) <= 10000
    assert got_xy.shape == want_xy.shape and (got_xy == want_xy).all() and (got_desc == want_desc).all()


def test_orb_degenerate_images(gpu_device, oracle):
    flat = np.full((128, 160), 100, dtype=np.uint8)
    xy, desc = orb.extract_points(gpu_device, flat)
    assert len(xy) == 0 and len(oracle.orb_extract(flat)[0]) == 0
    # corners only near the border: FAST finds them, Harris/BRIEF bounds drop them all
    img = np.full((64, 64), 30, dtype=np.uint8)
    img[4, 4] = img[4, 59] = img[59, 4] = 250
    assert len(oracle.orb_extract(img)[0]) == 0
    assert len(orb.extract_points(gpu_device, img)[0]) == 0


def test_orb_device_resident_image(gpu_device, oracle):
    import torch

    img = orb_image(320, 256, seed=4)
    got_xy, got_desc = orb.extract_points(gpu_device, torch.from_numpy(img).cuda())
    want_xy, want_desc = oracle.orb_extract(img)
    assert (got_xy == want_xy).all() and (got_desc == want_desc).all()


def test_matcher_matches_oracle(gpu_device, oracle):
    a, b, _ = synth.make_pair(512, 384, seed=31)
    img1 = synth.add_blocks(a, count=150, seed=5)
    img2 = synth.add_blocks(b, count=150, seed=5)
    k1 = oracle.orb_extract(img1)
    k2 = oracle.orb_extract(img2)
    for thr in (pointmatching.THRESHOLD_AFFINE, pointmatching.THRESHOLD_PERSPECTIVE, 0, 256):
        want_m, want_d = oracle.match_points(k1[0], k1[1], k2[0], k2[1], thr)
        got_m, got_d = pointmatching.match_points(gpu_device, k1[0], k1[1], k2[0], k2[1], thr)
        assert got_m.shape == want_m.shape and (got_m == want_m).all() and (got_d == want_d).all()
    assert len(oracle.match_points(k1[0], k1[1], k2[0], k2[1], 48)[0]) > 20


def test_matcher_tie_rules(gpu_device, oracle):
    """Many equal distances: first minimum per query and stable sort by distance."""
    rng = np.random.default_rng(7)
    base = rng.integers(0, 2 ** 32, size=(40, 8), dtype=np.uint64).astype(np.uint32)
    desc1 = np.repeat(base, 30, axis=0)[:1100]
    desc2 = np.concatenate([base, base, base ^ np.uint32(1)])[:117]
    xy1 = np.stack([np.arange(len(desc1)), np.arange(len(desc1)) * 2], axis=1).astype(np.uint32)
    xy2 = np.stack([np.arange(len(desc2)) + 7, np.arange(len(desc2))], axis=1).astype(np.uint32)
    want_m, want_d = oracle.match_points(xy1, desc1, xy2, desc2, 32)
    got_m, got_d = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, 32)
    assert len(want_m) == len(desc1)
    assert (got_m == want_m).all() and (got_d == want_d).all()


def test_matcher_ties_across_candidate_chunks(gpu_device, oracle):
    """The candidate list is split across workgroups (512-descriptor chunks) that meet in an atomicMin on
    (distance, index): equal distances in DIFFERENT chunks must still resolve to the first index, as the reference's
    serial scan does.  Every candidate descriptor occurs 75 times, spread over all chunks; queries hit them exactly, at
    distance 1 and at distance 2."""
    rng = np.random.default_rng(17)
    base = rng.integers(0, 2 ** 32, size=(40, 8), dtype=np.uint64).astype(np.uint32)
    desc2 = np.tile(base, (75, 1))                       # 3000 candidates, duplicates 40 apart
    desc1 = np.concatenate([base, base ^ np.uint32(1), base ^ np.uint32(3), np.roll(base, 1, axis=0)])
    xy1 = np.stack([np.arange(len(desc1)), np.arange(len(desc1)) + 3], axis=1).astype(np.uint32)
    xy2 = np.stack([np.arange(len(desc2)), 2 * np.arange(len(desc2))], axis=1).astype(np.uint32)
    for thr in (0, 1, 64):
        want_m, want_d = oracle.match_points(xy1, desc1, xy2, desc2, thr)
        got_m, got_d = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, thr)
        assert got_m.shape == want_m.shape and (got_m == want_m).all() and (got_d == want_d).all()
    assert (want_m[:, 2] < 40).all()  # (x2 = the candidate's index: always a first occurrence)


def test_matcher_matrix_pipe_path(gpu_device, oracle):
    """Large descriptor sets go through match_mfma_kernel (Hamming distance as an int8 matrix product): random
    descriptors with planted near-duplicates at every small distance, duplicates of one candidate spread over MORE than
    65 536 candidates (several splits of the candidate list, first index must win), ragged sizes, thresholds 0 .. 256."""
    rng = np.random.default_rng(23)

    def rand_desc(n):
        return rng.integers(0, 2 ** 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)

    # (a) random sets with near-duplicates: query i is candidate perm[i] with (i % 40) bits flipped
    n1, n2 = 2500, 2300
    desc2 = rand_desc(n2)
    desc1 = desc2[rng.integers(0, n2, size=n1)].copy()
    for i in range(n1):
        for b in rng.choice(256, size=i % 40, replace=False):
            desc1[i, b // 32] ^= np.uint32(1) << np.uint32(b % 32)
    desc1[-50:] = rand_desc(50)  # and some with no close candidate at all
    xy1 = np.stack([np.arange(n1), np.arange(n1) + 3], axis=1).astype(np.uint32)
    xy2 = np.stack([np.arange(n2), 2 * np.arange(n2)], axis=1).astype(np.uint32)
    for thr in (0, 5, 48, 64, 200, 256):
        want_m, want_d = oracle.match_points(xy1, desc1, xy2, desc2, thr)
        got_m, got_d = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, thr)
        assert got_m.shape == want_m.shape and (got_m == want_m).all() and (got_d == want_d).all(), thr
    # (b) ties across the splits of a long candidate list: 40 base descriptors, each 1 801 times among 72 040 candidates
    base = rand_desc(40)
    desc2 = np.tile(base, (1801, 1))
    desc1 = np.concatenate([base, base ^ np.uint32(1), base ^ np.uint32(3), np.roll(base, 1, axis=0)])
    xy1 = np.stack([np.arange(len(desc1)), np.arange(len(desc1)) + 3], axis=1).astype(np.uint32)
    xy2 = np.stack([np.arange(len(desc2)), 2 * np.arange(len(desc2))], axis=1).astype(np.uint32)
    for thr in (0, 1, 64):
        want_m, want_d = oracle.match_points(xy1, desc1, xy2, desc2, thr)
        got_m, got_d = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, thr)
        assert got_m.shape == want_m.shape and (got_m == want_m).all() and (got_d == want_d).all(), thr
    assert (want_m[:, 2] < 40).all()  # (x2 = the candidate's index: always a first occurrence)
    # (c) all-zero and all-one descriptors (popcounts 0 and 256: the key's bias)
    desc1 = np.zeros((2100, 8), dtype=np.uint32)
    desc1[1::2] = 0xFFFFFFFF
    desc2 = np.zeros((2050, 8), dtype=np.uint32)
    desc2[::3] = 0xFFFFFFFF
    desc2[5] = 0x0000FFFF
    xy1 = np.stack([np.arange(len(desc1)), np.arange(len(desc1))], axis=1).astype(np.uint32)
    xy2 = np.stack([np.arange(len(desc2)), np.arange(len(desc2))], axis=1).astype(np.uint32)
    for thr in (0, 128, 256):
        want_m, want_d = oracle.match_points(xy1, desc1, xy2, desc2, thr)
        got_m, got_d = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, thr)
        assert got_m.shape == want_m.shape and (got_m == want_m).all() and (got_d == want_d).all(), thr


def ransac_inputs(n=3000, hyp=700, seed=2):
    rng = np.random.default_rng(seed)
    x1 = rng.integers(0, 2000, size=n)
    y1 = rng.integers(0, 2000, size=n)
    x2 = np.clip(x1 + rng.integers(-60, 60, size=n), 0, None)
    y2 = y1.copy()
    out = rng.random(n) < 0.3
    y2[out] = rng.integers(0, 2000, size=int(out.sum()))
    m = np.stack([x1, y1, x2, y2], axis=1).astype(np.uint32)
    F = np.repeat(synth.F_HORIZONTAL[None], hyp, axis=0).copy()
    F += rng.normal(size=F.shape) * (10.0 ** rng.uniform(-6, -1, size=(hyp, 1, 1)))
    F[0] = synth.F_HORIZONTAL
    F[1] = 0.0                      # 0/0 -> NaN errors: nothing fits (fits_model rejects non-finite)
    F[2, 0, 0] = np.inf
    return F, m


@pytest.mark.parametrize("t", [0.1, 0.01 * 2000])
def test_ransac_score_matches_oracle_bitwise(gpu_device, oracle, t):
    F, m = ransac_inputs()
    want_c, want_e = oracle.ransac_score(F, m, t)
    got_c, got_e = fundamentalmatrix.ransac_score(gpu_device, F, m, t)
    assert (got_c == want_c).all(), "inlier counts differ"
    assert (got_e.view(np.uint64) == want_e.view(np.uint64)).all(), "error sums differ in the last bits"
    assert want_c[0] > 1500 and want_c[1] == 0 and want_c[2] == 0


def test_ransac_score_ragged_sizes(gpu_device, oracle):
    for n, hyp in [(1, 1), (63, 65), (1025, 129), (2048, 64)]:
        F, m = ransac_inputs(n=n, hyp=max(hyp, 3), seed=n)
        F = F[:hyp] if hyp < 3 else F
        want_c, want_e = oracle.ransac_score(F, m, 0.1)
        got_c, got_e = fundamentalmatrix.ransac_score(gpu_device, F, m, 0.1)
        assert (got_c == want_c).all() and (got_e.view(np.uint64) == want_e.view(np.uint64)).all()


@pytest.mark.parametrize("mfma", [True, False])
def test_round_counting_kernel_is_exact(gpu_device, oracle, oracle_fm, mfma):
    """(mfma: the screen's head as f32 matrix products - ransac_count_mfma_kernel, an option - or the vector kernel alone, the
    default; cvhip_ransac_set_count_mfma.)
    The counting kernel of a RANSAC round decides most (hypothesis, match) pairs in packed f32 against error bounds
    and only the guard band in f64: its counts must be the f64 counts - on good hypotheses whose inliers sit right at
    the threshold (t chosen AS the error of individual matches, and one ulp either side), on perturbed, random,
    huge, tiny and non-finite hypotheses; the ordered sums of the hypotheses tied at the maximum must be the oracle's."""
    import cases

    fundamentalmatrix.set_count_mfma(gpu_device, mfma)
    try:
        _round_counting_cases(gpu_device, oracle, oracle_fm)
    finally:
        fundamentalmatrix.set_count_mfma(gpu_device, False)


def _round_counting_cases(gpu_device, oracle, oracle_fm):
    import cases

    m, truth, _, _ = cases.perspective_matches(n=5000, outlier_frac=0.3)
    F0, _ = fundamentalmatrix.find_ransac_perspective_device(gpu_device, m, 2048.0, seed=2, rounds=1, refit=False)
    rng = np.random.default_rng(8)
    hyps = [F0]
    for scale in (1e-7, 1e-5, 1e-3, 1e-1):
        for _ in range(6):
            hyps.append(F0 * (1.0 + scale * rng.standard_normal((3, 3))))
    hyps += [rng.standard_normal((3, 3)) for _ in range(6)]
    hyps += [F0 * 1e-9, F0 * 1e-30, F0 * 1e9, F0 * 1e200, F0 * 1e-200]  # screen switched off by its range guards
    bad = F0.copy()
    bad[1, 1] = np.nan
    inf = F0.copy()
    inf[0, 2] = np.inf
    hyps += [bad, inf, np.zeros((3, 3))]
    hyps += [F0, F0]  # exact duplicates: tied at the maximum -> their ordered sums are computed
    F = np.stack(hyps)
    errs = np.sort(oracle_fm.reprojection_error(F0, m))
    ts = [20.48, 0.5, 1e-3]
    for q in (0.3, 0.5, 0.7):  # thresholds ON a match's error: err == t is an inlier, one ulp below t is not
        e = float(errs[int(q * len(errs))])
        ts += [e, float(np.nextafter(e, 0.0)), float(np.nextafter(e, np.inf))]
    for t in ts:
        want_c, want_e = oracle.ransac_score(F, m, t)
        full_c, _ = fundamentalmatrix.ransac_score(gpu_device, F, m, t)
        got_c, got_e = fundamentalmatrix.ransac_round_score(gpu_device, F, m, t)
        assert (full_c == want_c).all()
        assert (got_c == want_c).all(), (t, np.nonzero(got_c != want_c)[0], got_c[got_c != want_c], want_c[got_c != want_c])
        top = want_c.max()
        tied = np.nonzero(want_c == top)[0]
        if top > 0 and len(tied) > 1:
            assert (got_e[tied].view(np.uint64) == want_e[tied].view(np.uint64)).all()
        assert (got_e[want_c != top] == 0.0).all()
    assert want_c[0] > 1000
    # coordinates at the top of the u16 range (the bounds scale with the largest coordinate) and hypotheses of
    # arbitrary shape; thresholds again placed on individual errors of each hypothesis in turn
    rng = np.random.default_rng(12)
    big = rng.integers(0, 65536, size=(3000, 4)).astype(np.uint32)
    Fb = np.stack([rng.standard_normal((3, 3)) * rng.choice([1e-6, 1e-3, 1.0, 1e3], size=(3, 3)) for _ in range(12)])
    for hsel in range(0, 12, 3):
        eb = np.sort(oracle_fm.reprojection_error(Fb[hsel], big))
        for q in (0.2, 0.6):
            t = float(eb[int(q * len(eb))])
            if not (np.isfinite(t) and t > 0):
                continue
            for tt in (t, float(np.nextafter(t, 0.0)), float(np.nextafter(t, np.inf))):
                want_c, _ = oracle.ransac_score(Fb, big, tt)
                got_c, _ = fundamentalmatrix.ransac_round_score(gpu_device, Fb, big, tt)
                assert (got_c == want_c).all(), (hsel, q, tt, got_c, want_c)
    # ragged sizes around the 128-match step of the kernel
    for n in (1, 63, 64, 65, 127, 128, 129, 1000):
        want_c, _ = oracle.ransac_score(F, m[:n], 20.48)
        got_c, _ = fundamentalmatrix.ransac_round_score(gpu_device, F, m[:n], 20.48)
        assert (got_c == want_c).all(), n


def affine_matches(n=4000, outlier_frac=0.35, seed=5):
    """Matches of a known affine geometry x2^T F x1 = 0 with F = f_tilt(12 deg): the point in image 2 lies on
    the line through (x1, y1) with that direction, at a random signed distance (integer rounding noise)."""
    import math

    rng = np.random.default_rng(seed)
    th = math.radians(12.0)
    x1 = rng.integers(100, 1900, size=n).astype(np.float64)
    y1 = rng.integers(100, 1900, size=n).astype(np.float64)
    dist = rng.uniform(-80, 80, size=n)
    x2 = x1 + dist * math.cos(th)
    y2 = y1 + dist * math.sin(th)
    out = rng.random(n) < outlier_frac
    x2[out] = rng.integers(0, 2000, size=int(out.sum()))
    y2[out] = rng.integers(0, 2000, size=int(out.sum()))
    m = np.stack([x1, y1, np.round(x2), np.round(y2)], axis=1).clip(0, None).astype(np.uint32)
    return m, ~out, synth.f_tilt(12.0)


def test_device_affine_ransac_recovers_known_model(gpu_device, oracle):
    m, truth_inlier, F_true = affine_matches()
    F, mask = fundamentalmatrix.find_ransac_affine(gpu_device, m, seed=7)
    # affine form, normalised by f[2][2] (fundamentalmatrix.rs:284-285)
    assert F[2, 2] == 1.0 and (F[:2, :2] == 0).all()
    # direction of the epilines: (F[0][2], F[1][2]) ~ (-sin, cos) up to scale
    d = np.array([F[0, 2], F[1, 2]])
    d /= np.linalg.norm(d)
    want = np.array([F_true[0, 2], F_true[1, 2]])
    assert min(np.linalg.norm(d - want), np.linalg.norm(d + want)) < 5e-3
    # the inlier set is essentially the planted one; rounding to integer pixels costs ~10 %
    assert (mask & truth_inlier).sum() > 0.8 * truth_inlier.sum()  # t = 0.1 rejects the worst roundings
    assert (mask & ~truth_inlier).sum() < 0.05 * (~truth_inlier).sum()
    # the returned mask is exactly fits_model of the returned F (optimize_result, :231-239)
    cnt, _ = oracle.ransac_score(F, m, fundamentalmatrix.RANSAC_T_AFFINE)
    assert cnt[0] == mask.sum()
    errs = np.array([oracle.reprojection_error(F, mm) for mm in m[:200]])
    assert ((np.abs(errs) <= 0.1) == mask[:200]).all()
    # reproducible for a fixed seed, different (but equally good) for another
    F2, mask2 = fundamentalmatrix.find_ransac_affine(gpu_device, m, seed=7)
    assert (F2 == F).all() and (mask2 == mask).all()
    F3, mask3 = fundamentalmatrix.find_ransac_affine(gpu_device, m, seed=8)
    assert abs(int(mask3.sum()) - int(mask.sum())) < 0.05 * mask.sum()


def _samples(m, n, count, seed):
    """`count` samples of n match indices honouring choose_inliers' 10 px rule (:155-175), vectorised."""
    rng = np.random.default_rng(seed)
    mm = m.astype(np.int64)
    out = []
    while len(out) < count:
        idx = rng.integers(0, min(len(m), 5000), size=(4 * count, n))
        pts = mm[idx]
        d = np.abs(pts[:, :, None, :] - pts[:, None, :, :])
        close = (d < 10).any(axis=-1)
        close[:, np.arange(n), np.arange(n)] = False
        out.extend(idx[~close.any(axis=(1, 2))].tolist())
    return np.array(out[:count], dtype=np.uint32)


def test_device_affine_generator_matches_oracle_per_sample(gpu_device, oracle_fm):
    """calculate_model_affine + validate_f's checks on the device (one-sided Jacobi SVD of the centred 4x4) against
    the oracle's numpy restatement (LAPACK SVD) on identical 4-point samples: the same samples survive and every
    coefficient of F agrees to 1e-9 of the largest one."""
    m, _, _ = affine_matches(n=3000, outlier_frac=0.3, seed=8)
    idx = _samples(m, 4, 3000, seed=1)
    got = fundamentalmatrix.affine_models_device(gpu_device, m, idx, fundamentalmatrix.RANSAC_T_AFFINE)
    alive = 0
    for s_idx, Fg in zip(idx, got):
        sample = m[s_idx]
        F = oracle_fm.calculate_model_affine(sample)
        ok = F is not None and np.isfinite(F).all() and oracle_fm.fits_model(F, sample, 0.1).all()
        assert ok == bool(np.isfinite(Fg).all()), (s_idx, F, Fg)
        if ok:
            alive += 1
            assert np.abs(Fg - F).max() <= 1e-9 * np.abs(F).max(), np.abs(Fg - F).max() / np.abs(F).max()
    assert alive > 2500
    # degenerate samples: the same point four times, and four collinear-in-4D points (rank 1)
    deg = np.array([[5, 5, 5, 5], [0, 1, 2, 3]], dtype=np.uint32)
    mm = np.concatenate([np.array([[10, 20, 30, 40], [11, 21, 31, 41], [12, 22, 32, 42], [13, 23, 33, 43]], dtype=np.uint32),
                         np.tile(np.array([[7, 7, 7, 7]], dtype=np.uint32), (4, 1))])
    out = fundamentalmatrix.affine_models_device(gpu_device, mm, np.array([[4, 5, 6, 7], [0, 1, 2, 3]], dtype=np.uint32))
    assert np.isnan(out[0]).all() and oracle_fm.calculate_model_affine(mm[4:8]) is None
    del deg


@pytest.fixture
def null_space_pencil(gpu_device):
    """The textbook pencil (CVHIP_PENCIL_NULL_SPACE) for one test; the handle goes back to the reference's default."""
    fundamentalmatrix.set_pencil(gpu_device, fundamentalmatrix.PENCIL_NULL_SPACE)
    yield gpu_device
    fundamentalmatrix.set_pencil(gpu_device, fundamentalmatrix.PENCIL_THIN_SVD)


def _compare_generator_with_oracle(gpu_device, oracle_fm, m, idx, t, pencil):
    """-> (oracle hypotheses, device hypotheses, matched to 1e-7 of the largest entry, unmatched examples, device array)"""
    got = fundamentalmatrix.perspective_models_device(gpu_device, m, idx, t)          # [B, 3, 3, 3]
    n_ref = n_dev = matched = 0
    unmatched = []
    for b, s_idx in enumerate(idx):
        sample = m[s_idx]
        want = []
        for F in oracle_fm.calculate_model_perspective(sample, pencil):
            if not np.isfinite(F).all():
                continue
            Fo = oracle_fm.optimize_perspective_f(F, sample)
            if Fo is not None and oracle_fm.fits_model(Fo, sample, t).all():
                want.append(Fo)
        cand = got[b][np.isfinite(got[b][:, 0, 0])]
        n_ref += len(want)
        n_dev += len(cand)
        for Fo in want:
            rel = [np.abs(c - Fo).max() / np.abs(Fo).max() for c in cand]
            if rel and min(rel) < 1e-7:
                matched += 1
            else:
                unmatched.append((b, min(rel) if rel else None))
    return n_ref, n_dev, matched, unmatched, got


def test_device_perspective_generator_and_validate_f_match_oracle_per_sample(null_space_pencil, oracle_fm):
    """The device's 7-point generator in the TEXTBOOK mode (null-space pencil) INCLUDING validate_f's per-hypothesis part
    (finite test, optimize_perspective_f over the sample with its rank test on the re-parametrised matrix, sample-fit
    test; fundamentalmatrix.rs:192-209, 391-426) against the oracle's numpy restatement (LAPACK SVD, np.roots,
    np.linalg.solve) on identical samples: the same hypotheses survive - up to a handful that sit on a rank / sign
    threshold - and the surviving matrices agree to 1e-7 of their largest entry."""
    import cases

    gpu_device = null_space_pencil
    m, _, _, _ = cases.perspective_matches(n=3000, outlier_frac=0.2, seed=11)
    idx = _samples(m, 7, 2500, seed=4)
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    n_ref, n_dev, matched, unmatched, got = _compare_generator_with_oracle(gpu_device, oracle_fm, m, idx, t, oracle_fm.PENCIL_NULL_SPACE)
    assert n_ref > 300, n_ref
    assert abs(n_dev - n_ref) <= max(3, 0.005 * n_ref), (n_dev, n_ref)
    assert matched >= n_ref - max(3, 0.005 * n_ref), (matched, n_ref, unmatched[:5])
    # every surviving device hypothesis is in the re-parametrised form: F22 = 1 exactly, det = 0 to rounding
    alive = got[np.isfinite(got[:, :, 0, 0])]
    assert (alive[:, 2, 2] == 1.0).all()
    dets = np.abs(np.linalg.det(alive / np.abs(alive).max(axis=(1, 2), keepdims=True)))
    assert dets.max() < 1e-9


def test_device_reference_pencil_matches_thin_svd_rows_per_sample(gpu_device, oracle_fm):
    """calculate_model_perspective AS THE REFERENCE WRITES IT (:309-322: rows 5 and 6 of nalgebra's thin 7 x 9 v_t, the
    default mode) + validate_f's per-hypothesis part on the device against the oracle on identical samples.  The
    oracle takes np.linalg.svd(A, full_matrices=False) rows 5 / 6 in the same sign convention; everything downstream
    is the reference's: cubic, rank test before the F22 scale, sign test, then optimize_perspective_f's LM on EVERY
    root (none of them fits its sample at the start), rank test on the re-parametrised matrix, sample-fit test.
    (a) with the sample-fit test disarmed (t = 1e300) every root that leaves the LM is compared - the pencil, the roots
    and the LM trajectory (all steps rejected until the step vanishes, or accepted ones) to 1e-7;
    (b) with the reference's t the few hypotheses that fit their own sample are the same ones."""
    import cases

    m, _, _, _ = cases.perspective_matches(n=3000, outlier_frac=0.2, seed=11)
    idx = _samples(m, 7, 1500, seed=4)
    n_ref, n_dev, matched, unmatched, got = _compare_generator_with_oracle(gpu_device, oracle_fm, m, idx, 1e300, oracle_fm.PENCIL_THIN_SVD)
    print(f"thin-SVD pencil, t = inf: oracle {n_ref}, device {n_dev}, matched {matched}; unmatched e.g. {unmatched[:8]}")
    assert n_ref > 1500, n_ref
    assert abs(n_dev - n_ref) <= max(3, 0.01 * n_ref), (n_dev, n_ref)
    assert matched >= n_ref - max(3, 0.02 * n_ref), (matched, n_ref, unmatched[:5])
    alive = got[np.isfinite(got[:, :, 0, 0])]
    assert (alive[:, 2, 2] == 1.0).all()
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    idx = _samples(m, 7, 6000, seed=5)
    n_ref, n_dev, matched, unmatched, got = _compare_generator_with_oracle(gpu_device, oracle_fm, m, idx, t, oracle_fm.PENCIL_THIN_SVD)
    print(f"thin-SVD pencil, reference t: oracle {n_ref}, device {n_dev}, matched {matched}; unmatched e.g. {unmatched[:8]}")
    # the two-kernel form of the loop gives the scalar loop's values, bit for bit (NaN slots included)
    fundamentalmatrix.set_lm_pipeline(gpu_device, False)
    try:
        scalar = fundamentalmatrix.perspective_models_device(gpu_device, m, idx, t)
        scalar_inf = fundamentalmatrix.perspective_models_device(gpu_device, m, idx[:1500], 1e300)
    finally:
        fundamentalmatrix.set_lm_pipeline(gpu_device, True)
    assert (scalar.view(np.uint64) == got.view(np.uint64)).all()
    piped_inf = fundamentalmatrix.perspective_models_device(gpu_device, m, idx[:1500], 1e300)
    assert (scalar_inf.view(np.uint64) == piped_inf.view(np.uint64)).all() and np.isfinite(piped_inf[:, :, 0, 0]).sum() > 1500
    # ... and so does the root-per-thread form of the two passes (mode 1; the default above is mode 2, the refilled lanes)
    fundamentalmatrix.set_lm_pipeline(gpu_device, 1)
    try:
        threads = fundamentalmatrix.perspective_models_device(gpu_device, m, idx, t)
        threads_inf = fundamentalmatrix.perspective_models_device(gpu_device, m, idx[:1500], 1e300)
    finally:
        fundamentalmatrix.set_lm_pipeline(gpu_device, True)
    assert (scalar.view(np.uint64) == threads.view(np.uint64)).all() and (scalar_inf.view(np.uint64) == threads_inf.view(np.uint64)).all()
    assert n_ref >= 20, n_ref                      # a few per cent of the roots fit their own sample
    assert abs(n_dev - n_ref) <= max(2, 0.05 * n_ref), (n_dev, n_ref)
    assert matched >= n_ref - max(2, 0.05 * n_ref), (matched, n_ref, unmatched[:5])
    # the two modes are different algorithms: the null-space pencil's survivors are many
    fundamentalmatrix.set_pencil(gpu_device, fundamentalmatrix.PENCIL_NULL_SPACE)
    try:
        other = fundamentalmatrix.perspective_models_device(gpu_device, m, idx[:500], t)
    finally:
        fundamentalmatrix.set_pencil(gpu_device, fundamentalmatrix.PENCIL_THIN_SVD)
    assert np.isfinite(other[:, :, 0, 0]).sum() > 2 * np.isfinite(got[:500, :, 0, 0]).sum()


def test_find_ransac_one_call_both_models(gpu_device, oracle, oracle_fm):
    """FundamentalMatrix::new(..).find_ransac (:72-147, 231-257) as the single call the pipeline makes
    (cvhip_find_ransac): the planted geometry is recovered for both models, the returned inliers are exactly
    fits_model of the returned F (the oracle's), and for the perspective model F is the LM refit of a hypothesis."""
    import cases

    m, truth, _, _ = cases.perspective_matches(n=4000, outlier_frac=0.3)
    fmx = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Perspective, 2048.0)
    F, inliers, mask = fmx.find_ransac(gpu_device, m, seed=3)
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    assert F[2, 2] == 1.0 and abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9
    assert (mask & truth).sum() > 0.97 * truth.sum() and (mask & ~truth).sum() < 0.1 * (~truth).sum()
    assert np.median(np.abs(oracle_fm.reprojection_error(F, m[truth]))) < 1.0
    cnt, _ = oracle.ransac_score(F, m, t)
    assert cnt[0] == mask.sum() == len(inliers) and (inliers == m[mask]).all()
    assert (fundamentalmatrix.inlier_mask(gpu_device, F, m, t) == mask).all()
    ma, truth_a, _ = affine_matches()
    Fa, inl_a, mask_a = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Affine, 2000.0).find_ransac(gpu_device, ma, seed=7)
    F2, mask2 = fundamentalmatrix.find_ransac_affine(gpu_device, ma, seed=7)
    assert (Fa == F2).all() and (mask_a == mask2).all() and len(inl_a) == mask2.sum()
    from cybervision_amd._lib import CvhipError
    with pytest.raises(CvhipError) as ei:
        fmx.find_ransac(gpu_device, m[:100])
    assert ei.value.code == -5 and "Not enough matches" in str(ei.value)


def test_device_perspective_ransac_recovers_planted_geometry(gpu_device, oracle, oracle_fm):
    import cases

    m, truth, _, _ = cases.perspective_matches(n=4000, outlier_frac=0.3)
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    F0, mask0 = fundamentalmatrix.find_ransac_perspective_device(gpu_device, m, 2048.0, seed=5, rounds=2, refit=False)
    assert F0[2, 2] == 1.0
    cnt, _ = oracle.ransac_score(F0, m, t)
    assert cnt[0] == mask0.sum()  # the mask is exactly fits_model of the returned F
    assert (mask0 & truth).sum() > 0.95 * truth.sum()
    F, mask = fundamentalmatrix.find_ransac_perspective_device(gpu_device, m, 2048.0, seed=5, rounds=2)
    # optimize_result (:231-257): the reference's LM refit on the winner's inliers, deterministic -> the oracle's bits
    want = oracle.optimize_perspective_f(F0, m[mask0])
    assert (F == (F0 if want is None else want)).all()
    assert (mask & truth).sum() > 0.97 * truth.sum() and (mask & ~truth).sum() < 0.1 * (~truth).sum()
    assert np.median(np.abs(oracle_fm.reprojection_error(F, m[truth]))) < 1.0
    F2, mask2 = fundamentalmatrix.find_ransac_perspective_device(gpu_device, m, 2048.0, seed=5, rounds=2, refit=False)
    assert (F2 == F0).all() and (mask2 == mask0).all()  # reproducible for a fixed seed
    from cybervision_amd._lib import CvhipError
    with pytest.raises(CvhipError) as ei:
        fundamentalmatrix.find_ransac_perspective_device(gpu_device, m[:100], 2048.0)
    assert ei.value.code == -5 and "Not enough matches" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", [1, 3, 8])
def test_device_rounds_pick_ord_maximum(gpu_device, rounds):
    """The device loops' round machinery (cvhip_ransac_rounds_pick: counting kernel on its re-sorted list with the
    abandonment bounds, the candidate list, ransac_round_finish_kernel) on caller-given hypotheses against Ord's maximum
    (fundamentalmatrix.rs:623-649) over the full fold of every hypothesis (cvhip_ransac_score): the same hypothesis wins
    - generated models with their dead slots, near-duplicates of the good ones, and exact duplicates that tie."""
    import cases

    m, truth, _, _ = cases.perspective_matches(n=6000, outlier_frac=0.35)
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    rng = np.random.default_rng(11 + rounds)
    good = np.flatnonzero(truth)
    idx = np.stack([rng.choice(len(m), 7, replace=False) for _ in range(1500)] +
                   [rng.choice(good, 7, replace=False) for _ in range(500)]).astype(np.uint32)
    F = fundamentalmatrix.perspective_models_device(gpu_device, m, idx, t).reshape(-1, 9)  # NaN rows = dead slots
    cnt, err = fundamentalmatrix.ransac_score(gpu_device, np.nan_to_num(F, nan=0.0), m, t)
    cnt = np.where(np.isfinite(F).all(axis=1), cnt, 0)
    order = np.argsort(-cnt.astype(np.int64), kind="stable")
    top = F[order[:40]]
    pert = top * (1.0 + 1e-7 * rng.standard_normal(top.shape))
    pert[:, 8] = 1.0
    F = np.concatenate([F, pert, top[:5], top[:5]])  # exact duplicates of the five best: ties in count AND error
    F = F[rng.permutation(len(F))]
    live = np.isfinite(F).all(axis=1)
    cnt, err = fundamentalmatrix.ransac_score(gpu_device, np.nan_to_num(F, nan=0.0), m, t)
    cnt = np.where(live, cnt, 0).astype(np.int64)
    min_count = 207
    # Ord: more matches win; among equal counts the smaller mean error; the first of equals stays
    best = -1
    for h in np.flatnonzero(cnt >= min_count):
        if best < 0 or cnt[h] > cnt[best] or (cnt[h] == cnt[best] and err[h] / cnt[h] < err[best] / cnt[best]):
            best = int(h)
    assert best >= 0 and (cnt == cnt[best]).sum() >= 2  # the planted duplicates do tie
    got_idx, got_F, got_cnt, got_err = fundamentalmatrix.ransac_rounds_pick(gpu_device, F, rounds, m, t, min_count)
    assert got_cnt == cnt[best]
    assert (got_F.reshape(9) == F[best]).all(), (got_idx, best)
    first_equal = int(np.flatnonzero((F == F[best]).all(axis=1))[0])
    assert got_idx == first_equal
    if np.isfinite(got_err):
        assert abs(got_err - err[best] / cnt[best]) <= 1e-9 * abs(err[best] / cnt[best])
    # nothing reaches an impossible minimum: no winner
    assert fundamentalmatrix.ransac_rounds_pick(gpu_device, F, rounds, m, t, len(m) + 1)[0] == -1


def test_device_refit_equals_host_refit_bit_for_bit(gpu_device, oracle):
    """cvhip_optimize_perspective_f_device (one workgroup, eight threads per long dot product) against
    cvhip_optimize_perspective_f (one host thread) and the oracle: same loop, same accumulation order -> same bits.
    Inlier counts around the multiples of 8 (the partial-sum blocks and their tail), a start the loop accepts steps
    from, a start it returns at once, and a degenerate one (None)."""
    import cases

    m, truth, _, _ = cases.perspective_matches(n=3000, outlier_frac=0.2)
    F0, mask0 = fundamentalmatrix.find_ransac_perspective_device(gpu_device, m, 2048.0, seed=9, rounds=1, refit=False)
    inl = m[mask0]
    assert len(inl) > 1500
    rng = np.random.default_rng(4)
    starts = [F0]
    for scale in (1e-3, 3e-2):  # perturbed starts: several accepted and rejected steps before the loop ends
        P = F0 * (1.0 + scale * rng.standard_normal((3, 3)))
        starts.append(P / P[2, 2])
    refined_some = False
    for F in starts:
        for n in (len(inl), 1024, 1001, 808, 807, 801, 64, 15, 8, 7, 3, 1, 0):
            host = fundamentalmatrix.optimize_perspective_f(F, inl[:n])
            dev = fundamentalmatrix.optimize_perspective_f_device(gpu_device, F, inl[:n])
            assert (host is None) == (dev is None), (n, host, dev)
            if host is not None:
                refined_some = True
                assert (host.view(np.uint64) == dev.view(np.uint64)).all(), (n, host - dev)
    assert refined_some
    want = oracle.optimize_perspective_f(F0, inl)
    got = fundamentalmatrix.optimize_perspective_f_device(gpu_device, F0, inl)
    assert (want is None) == (got is None) and (want is None or (want == got).all())
    # degenerate start: rank test / solve failure -> None on both
    Z = np.zeros((3, 3))
    Z[2, 2] = 1.0
    assert fundamentalmatrix.optimize_perspective_f(Z, inl[:50]) is None
    assert fundamentalmatrix.optimize_perspective_f_device(gpu_device, Z, inl[:50]) is None


def test_device_affine_ransac_error_reporting(gpu_device):
    from cybervision_amd._lib import CvhipError

    with pytest.raises(CvhipError) as ei:
        fundamentalmatrix.find_ransac_affine(gpu_device, np.zeros((5, 4), dtype=np.uint32))
    assert ei.value.code == -5 and "Not enough matches" in str(ei.value)
    rng = np.random.default_rng(1)
    noise = rng.integers(0, 2000, size=(300, 4)).astype(np.uint32)  # no consistent geometry at t = 0.1
    with pytest.raises(CvhipError) as ei:
        fundamentalmatrix.find_ransac_affine(gpu_device, noise, seed=3)
    assert ei.value.code == -5 and "No reliable matches" in str(ei.value)


def test_orb_multiscale_driver_matches_oracle(gpu_device, oracle):
    """match_keypoints' per-image loop (reconstruction.rs:407-459): levels coarse to fine, coordinates
    mapped back with (x as f32 / scale) as usize, lists concatenated."""
    img = orb_image(1100, 900, seed=13, blocks=500)
    steps = orb.optimal_scale_steps(1100, 900)
    assert steps == 1 and oracle.lib().cvref_orb_optimal_scale_steps(1100, 900) == 1
    pyr = synth.box_pyramid(img, steps)
    got_xy, got_desc = orb.extract_points_multiscale(gpu_device, pyr)
    want_xy, want_desc = [], []
    for i in range(steps + 1):
        k = steps - i
        xy, desc = oracle.orb_extract(pyr[k])
        scale = np.float32(1.0 / (1 << k))
        want_xy.append(np.floor(xy.astype(np.float32) / scale).astype(np.uint32))
        want_desc.append(desc)
    want_xy, want_desc = np.concatenate(want_xy), np.concatenate(want_desc)
    assert len(want_xy) > 1000
    assert (got_xy == want_xy).all() and (got_desc == want_desc).all()


def test_progress_listeners_of_orb_and_ransac(gpu_device):
    """`Option<&PL>` of orb::extract_points (orb.rs:43-53) and find_ransac (fundamentalmatrix.rs:41-47, 103) through the
    ABI: positions rise through the reference's stage boundaries to 1.0, report_matches carries the running maximum of
    the inlier counts, and a listener changes nothing about the results."""
    import cases

    img = orb_image(640, 480)
    pos = []
    xy, desc = orb.extract_points(gpu_device, img, progress=pos.append)
    xy0, desc0 = orb.extract_points(gpu_device, img)
    assert (xy == xy0).all() and (desc == desc0).all() and len(xy) > 100
    assert pos == sorted(pos) and pos[0] == pytest.approx(0.20) and pos[-1] == 1.0
    assert [round(p, 2) for p in pos] == [0.20, 0.25, 0.35, 0.70, 1.0]
    pos = []
    assert len(orb.extract_points(gpu_device, np.full((64, 64), 7, dtype=np.uint8), progress=pos.append)[0]) == 0
    assert pos[-1] == 1.0  # no corners: the listener still sees the end

    class Listener:
        def __init__(self):
            self.status, self.matches = [], []

        def report_status(self, p):
            self.status.append(p)

        def report_matches(self, c):
            self.matches.append(c)

    m, truth, _, _ = cases.perspective_matches(n=4000, outlier_frac=0.3)
    fmx = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Perspective, 2048.0)
    pl = Listener()
    F, inliers, mask = fmx.find_ransac(gpu_device, m, seed=3, progress_listener=pl)
    F0, _, mask0 = fmx.find_ransac(gpu_device, m, seed=3)
    assert (F == F0).all() and (mask == mask0).all()
    assert len(pl.status) == 20 and pl.status == sorted(pl.status) and pl.status[-1] == 1.0  # 20 rounds, no early exit
    # the early exit cannot fire (4000 matches < 50 000): the rounds are scored in batches of four, the listener gets the
    # running maximum once per scored batch (five) and once more with the final result
    assert len(pl.matches) == 6 and pl.matches == sorted(pl.matches) and 0.9 * truth.sum() < pl.matches[-1] <= len(m)
    assert pl.matches[-1] >= mask0.sum() * 0.9  # (the last count is the winner's before the refit)
    ma, truth_a, _ = affine_matches()
    pl = Listener()
    fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Affine, 2000.0).find_ransac(gpu_device, ma, seed=7, progress_listener=pl)
    assert 1 <= len(pl.status) <= 20 and pl.status == sorted(pl.status)   # early exit above 1000 inliers (:135-141)
    assert pl.matches == sorted(pl.matches) and pl.matches[-1] > 1000


def test_ransac_scheduler_paths_agree(gpu_device):
    """The round scheduler of the perspective loop (ransac_rounds): batches scored as their generators finish (the host
    polls, bounded), in order behind their events (the branch the bounded polling falls back to, forced through
    cvhip_ransac_set_in_order), and with the reference's listener attached (fundamentalmatrix.rs:112-142,
    reconstruction.rs:510-518 always passes one) - the same matrix and mask every way, for both pencils."""
    import cases

    class Listener:
        def __init__(self):
            self.status, self.matches = [], []

        def report_status(self, p):
            self.status.append(p)

        def report_matches(self, c):
            self.matches.append(c)

    m, truth, _, _ = cases.perspective_matches(n=6000, outlier_frac=0.4, seed=21)
    fmx = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Perspective, 2048.0)
    for pencil in (fundamentalmatrix.PENCIL_THIN_SVD, fundamentalmatrix.PENCIL_NULL_SPACE):
        fundamentalmatrix.set_pencil(gpu_device, pencil)
        try:
            F0, _, mask0 = fmx.find_ransac(gpu_device, m, seed=9)
            fundamentalmatrix.set_in_order(gpu_device, True)
            try:
                F1, _, mask1 = fmx.find_ransac(gpu_device, m, seed=9)
                pl = Listener()
                F2, _, mask2 = fmx.find_ransac(gpu_device, m, seed=9, progress_listener=pl)
            finally:
                fundamentalmatrix.set_in_order(gpu_device, False)
            pl3 = Listener()
            F3, _, mask3 = fmx.find_ransac(gpu_device, m, seed=9, progress_listener=pl3)
            # ... and with the counting screen's head as f32 matrix products (cvhip_ransac_set_count_mfma: phase 1 on
            # ransac_count_mfma_kernel, survivors in the vector kernel) - the same exact counts, hence the same winner
            fundamentalmatrix.set_count_mfma(gpu_device, True)
            try:
                F4, _, mask4 = fmx.find_ransac(gpu_device, m, seed=9)
            finally:
                fundamentalmatrix.set_count_mfma(gpu_device, False)
        finally:
            fundamentalmatrix.set_pencil(gpu_device, fundamentalmatrix.PENCIL_THIN_SVD)
        for F, mask in ((F1, mask1), (F2, mask2), (F3, mask3), (F4, mask4)):
            assert (F.view(np.uint64) == F0.view(np.uint64)).all() and (mask == mask0).all(), pencil
        assert len(pl.matches) == 6 and len(pl3.matches) == 6 and pl.matches[-1] == pl3.matches[-1]
        assert (mask0 & truth).sum() > 0.9 * truth.sum(), (pencil, (mask0 & truth).sum(), truth.sum())


def test_orb_batch_equals_single_extractions(gpu_device, oracle):
    """cvhip_orb_extract_batch: several images of different sizes in one call - among them one without any corner and
    one above the keypoint cap - give exactly what one cvhip_orb_extract per image gives (and so the oracle's
    keypoints); likewise the multi-scale drivers built on it."""
    imgs = [orb_image(400, 300, seed=3), np.full((80, 120), 9, dtype=np.uint8), orb_image(1024, 768, seed=5, blocks=2500),
            orb_image(97, 131, seed=6)]
    single = [orb.extract_points(gpu_device, im) for im in imgs]
    pos = []
    batch = orb.extract_points_batch(gpu_device, imgs, progress=pos.append)
    assert [round(p, 2) for p in pos] == [0.20, 0.25, 0.35, 0.70, 1.0]
    for (xy_a, d_a), (xy_b, d_b) in zip(single, batch):
        assert xy_a.shape == xy_b.shape and (xy_a == xy_b).all() and (d_a == d_b).all()
    assert len(batch[1][0]) == 0 and len(batch[2][0]) > 5000
    want_xy, want_desc = oracle.orb_extract(imgs[0])
    assert (batch[0][0] == want_xy).all() and (batch[0][1] == want_desc).all()
    import torch
    dev_imgs = [torch.from_numpy(im).cuda() for im in imgs]
    for (xy_a, d_a), (xy_b, d_b) in zip(single, orb.extract_points_batch(gpu_device, dev_imgs)):
        assert (xy_a == xy_b).all() and (d_a == d_b).all()
    base = orb_image(1100, 900, seed=8, blocks=900)
    steps = orb.optimal_scale_steps(1100, 900)
    pyr = synth.box_pyramid(base, steps)
    a = orb.extract_points_multiscale(gpu_device, pyr, batched=False)
    b = orb.extract_points_multiscale(gpu_device, pyr, batched=True)
    c = orb.extract_points_multiscale_set(gpu_device, [pyr, pyr[1:]])
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (c[0][0] == a[0]).all() and (c[0][1] == a[1]).all()
    assert len(c[1][0]) > 100


def test_orb_batch_of_more_than_16_images(gpu_device):
    """Up to 16 images go out as one launch per kernel, larger batches image by image over the handle's streams: both
    forms give what single extractions give (several calls, so that the handle's choice of 3 / 2 / 1 chains is exercised)."""
    sizes = [(200 + 13 * i, 150 + 7 * (i % 5)) for i in range(18)]
    imgs = [orb_image(w, h, seed=40 + i, blocks=60) for i, (w, h) in enumerate(sizes)]
    single = [orb.extract_points(gpu_device, im) for im in imgs]
    assert sum(len(xy) for xy, _ in single) > 2000
    for n in (16, 18, 18, 18, 18, 17):
        for (xy_a, d_a), (xy_b, d_b) in zip(single[:n], orb.extract_points_batch(gpu_device, imgs[:n])):
            assert xy_a.shape == xy_b.shape and (xy_a == xy_b).all() and (d_a == d_b).all()


def test_orb_orientation_guard_settings_agree(gpu_device, oracle):
    """The orientation step: device atan2 / sin / cos where every rounded sample offset is provably the libm one, the
    host's libm otherwise (cvhip_orb_set_orientation_guard).  Default guard, a guard so wide that most images take the
    host path, every image forced through it, and the guard off must all give the oracle's keypoints and descriptors
    (the oracle calls libm)."""
    imgs = [orb_image(640, 480, seed=21), orb_image(333, 517, seed=22, blocks=300)]
    want = [oracle.orb_extract(im) for im in imgs]
    try:
        for guard in (1e-9, 0.02, 0.5, 0.0):
            orb.set_orientation_guard(gpu_device, guard)
            got = orb.extract_points_batch(gpu_device, imgs)
            for (xy, desc), (wxy, wdesc) in zip(got, want):
                assert xy.shape == wxy.shape and (xy == wxy).all() and (desc == wdesc).all(), f"guard {guard}"
    finally:
        orb.set_orientation_guard(gpu_device, 1e-9)
