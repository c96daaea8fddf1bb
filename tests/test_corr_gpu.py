"""GPU parity tests of the dense-correlation path: every call goes through the C ABI
(libcvhip.so via cybervision_amd.correlation) and is compared bit-for-bit with the CPU oracle
and with the committed golden fixtures.  Integer outputs (match coordinates) and float scores
are both required to be identical: the kernels keep the reference's f32/f64 operation order."""
from pathlib import Path

import numpy as np
import pytest

import cases
from cybervision_amd import correlation, synth

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_same_grid(got, want, what):
    gxy, gc = got
    wxy, wc = want
    diff = np.nonzero((gxy != wxy).any(axis=-1))
    assert diff[0].size == 0, f"{what}: {diff[0].size} match cells differ, first at (y, x) = " \
                              f"({diff[0][0]}, {diff[1][0]}): got {gxy[diff[0][0], diff[1][0]]} " \
                              f"want {wxy[diff[0][0], diff[1][0]]}"
    valid = wxy[..., 0] >= 0
    assert (bits(gc)[valid] == bits(wc)[valid]).all(), f"{what}: scores differ in the last bits"
    assert np.isnan(gc[~valid]).all()


def run_gpu(dev, c, fused=True, both=False, version=None, counters=None, range_mode=None, exact_scores=None):
    """exact_scores: None = on exactly when the reverse grid is read (its scores are only the reference's bits under
    cvhip_ctx_set_exact_scores; the forward grid of the last level always is)."""
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pc = correlation.PointCorrelations(dev, (w1, h1), (w2, h2), c["F"], correlation.ProjectionMode(c["projection"]))
    pc.set_exact_scores(both if exact_scores is None else exact_scores)
    if version is not None:
        pc.set_search_version(version)
    if range_mode is not None:
        pc.set_range_mode(range_mode)
    if counters is not None:
        pc.set_profiling(False, True)
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=fused)
        fwd = pc.complete(correlation.CorrelationDirection.Forward)
        if counters is not None:
            counters.update(pc.get_counters())
        if both:
            return fwd, pc.complete(correlation.CorrelationDirection.Reverse)
        return fwd
    finally:
        pc.close()


def run_oracle(oracle, c, both=False):
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    oc = oracle.Corr((w1, h1), (w2, h2), c["F"], c["projection"], 8)
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            oc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        if both:
            return oc.get(0), oc.get(1)
        return oc.get(0)
    finally:
        oc.close()


def test_device_name(gpu_device):
    assert "gfx950" in gpu_device.name() or "MI3" in gpu_device.name()


@pytest.mark.parametrize("name", cases.CASES)
def test_matches_oracle_bit_exact(gpu_device, oracle, name):
    c = cases.make_case(name)
    got_f, got_r = run_gpu(gpu_device, c, both=True)
    want_f, want_r = run_oracle(oracle, c, both=True)
    assert_same_grid(got_f, want_f, f"{name} forward")
    assert_same_grid(got_r, want_r, f"{name} reverse")


@pytest.mark.parametrize("name", cases.CASES)
def test_exact_kernel_v1_matches_oracle(gpu_device, oracle, name):
    """The plain kernel that sends every candidate through the exact serial f32 chain."""
    c = cases.make_case(name)
    assert_same_grid(run_gpu(gpu_device, c, version=1), run_oracle(oracle, c), f"{name} v1")


@pytest.mark.parametrize("name", cases.CASES)
def test_candidate_filter_v2_matches_oracle(gpu_device, oracle, name):
    """Version 2 (the per-candidate integer filter) is the per-workgroup fallback of the default box
    filter (version 3), so it has to stay exact on its own."""
    c = cases.make_case(name)
    got_f, got_r = run_gpu(gpu_device, c, both=True, version=2)
    want_f, want_r = run_oracle(oracle, c, both=True)
    assert_same_grid(got_f, want_f, f"{name} v2 forward")
    assert_same_grid(got_r, want_r, f"{name} v2 reverse")


@pytest.mark.parametrize("name", ["h256", "sem320x200", "tilt05_400x300", "tilt3_200x150", "persp_240x180", "ragged_dims",
                                  "vert_200x260"])
def test_box_filter_declines_per_workgroup(gpu_device, oracle, name):
    """Version 4 = the box filter launched for every geometry: on unrectified pairs it settles the workgroups
    whose pixels all have rectangular candidate sets and hands the rest to the candidate filter, so one grid
    is written by both kernels."""
    c = cases.make_case(name)
    got_f, got_r = run_gpu(gpu_device, c, both=True, version=4)
    want_f, want_r = run_oracle(oracle, c, both=True)
    assert_same_grid(got_f, want_f, f"{name} v4 forward")
    assert_same_grid(got_r, want_r, f"{name} v4 reverse")


def adversarial_case(kind):
    """Inputs built to stress the filter's decision rule: exact ties (periodic texture: many
    candidates with IDENTICAL scores, the first must win), near-threshold scores and windows whose
    stdev is close to MIN_STDEV (largest relative rounding error of the reference's f32 chain)."""
    rng = np.random.default_rng(77)
    h, w = 160, 200
    if kind == "periodic":
        base = rng.integers(0, 256, size=(8, 4), dtype=np.uint8)
        a = np.tile(base, (h // 8, w // 4))
        b = a.copy()
    elif kind == "periodic16":  # two or three exact ties per pixel: within the contender list, shared out over idle lanes
        h, w = 192, 320
        base = rng.integers(0, 256, size=(16, 8), dtype=np.uint8)
        a = np.tile(base, (h // 16, w // 8))
        b = a.copy()
    elif kind == "low_contrast":
        a = (100 + rng.integers(0, 4, size=(h, w))).astype(np.uint8)   # stdev ~ 1.1
        b = np.roll(a, 3, axis=1)
    elif kind == "noisy":
        t, t2, _ = synth.make_pair(w, h, seed=5)
        n = rng.integers(-40, 41, size=(h, w))
        a = t
        b = np.clip(t2.astype(np.int64) + n, 0, 255).astype(np.uint8)  # scores hover around the 0.6 threshold
    else:
        raise KeyError(kind)
    steps = synth.optimal_scale_steps(w, h)
    return dict(img1=np.ascontiguousarray(a), img2=np.ascontiguousarray(b), F=synth.F_HORIZONTAL, projection=0,
                steps=steps)


@pytest.mark.parametrize("kind", ["periodic", "periodic16", "low_contrast", "noisy"])
def test_filter_decision_rule_on_adversarial_inputs(gpu_device, oracle, kind):
    c = adversarial_case(kind)
    cnt = {}
    got = run_gpu(gpu_device, c, both=True, counters=cnt)
    want = run_oracle(oracle, c, both=True)
    assert_same_grid(got[0], want[0], f"{kind} forward")
    assert_same_grid(got[1], want[1], f"{kind} reverse")
    if kind == "periodic":  # ties must actually have been exercised
        assert cnt["multi_contender_pixels"] + cnt["whole_corridor_pixels"] > 1000, cnt
    if kind == "periodic16":  # ... and as lists of contenders, not only as whole-corridor pixels
        assert cnt["multi_contender_pixels"] > 1000, cnt
    # the launches without counters (another instantiation of the same kernels), and the stepped ones forced
    for version in (3, 4):
        again = run_gpu(gpu_device, c, both=True, version=version)
        assert_same_grid(again[0], want[0], f"{kind} forward, version {version}, no counters")
        assert_same_grid(again[1], want[1], f"{kind} reverse, version {version}, no counters")


def test_filter_statistics(gpu_device):
    """On ordinary textured input nearly every pixel needs exactly one exact evaluation."""
    c = cases.make_case("h256")
    cnt = {}
    run_gpu(gpu_device, c, counters=cnt)
    assert cnt["exact_evals"] < 0.1 * cnt["candidates"], cnt
    assert cnt["whole_corridor_pixels"] < 0.01 * 256 * 256 * 2 * 3, cnt


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_matches_golden_fixture(gpu_device, name):
    g = np.load(GOLDEN / f"corr_{name}.npz")
    c = dict(img1=g["img1"], img2=g["img2"], F=g["F"], projection=int(g["projection"]), steps=int(g["steps"]))
    got_f, got_r = run_gpu(gpu_device, c, both=True)
    assert_same_grid(got_f, (g["fwd_xy"].astype(np.int32), g["fwd_corr"]), f"{name} forward vs golden")
    assert_same_grid(got_r, (g["rev_xy"].astype(np.int32), g["rev_corr"]), f"{name} reverse vs golden")


def test_per_pass_calls_equal_fused_level_call(gpu_device):
    """The reference's four backend calls per level (mod.rs:224-240) and cvhip_correlate_level
    give the same grids."""
    c = cases.make_case("tilt3_200x150")
    assert_same_grid(run_gpu(gpu_device, c, fused=False), run_gpu(gpu_device, c, fused=True), "per-pass vs fused")


@pytest.mark.parametrize("name", ["h256", "sem320x200", "ragged_dims"])
def test_matrix_pipe_filter_matches_oracle(gpu_device, oracle, name):
    """Search version 5: the rectified affine launches as int8 matrix products (search4_mfma_kernel: sixteen positions x
    sixteen pixels x four target rows per v_mfma_i32_16x16x64_i8, the window sums biased by 128 and corrected exactly) -
    the same integers as the box filter, so both directions' grids are the oracle's bit for bit; and on a larger pair with
    disparity discontinuities (workgroups that decline, pixels with more events than their queue holds)."""
    c = cases.make_case(name)
    assert_same_grid(run_gpu(gpu_device, c, version=5, both=True)[0], run_oracle(oracle, c, both=True)[0], f"{name} v5 forward")
    assert_same_grid(run_gpu(gpu_device, c, version=5, both=True)[1], run_oracle(oracle, c, both=True)[1], f"{name} v5 reverse")
    if name == "h256":
        a, b, _ = synth.make_pair(1024, 768, seed=77)
        big = {"img1": a, "img2": b, "F": synth.F_HORIZONTAL, "projection": 0, "steps": synth.optimal_scale_steps(1024, 768)}
        assert_same_grid(run_gpu(gpu_device, big, version=5), run_gpu(gpu_device, big, version=3), "1024x768 v5 vs v3")


@pytest.mark.parametrize("name", ["h256", "sem320x200", "periodic", "periodic16", "low_contrast", "noisy", "big"])
def test_two_column_box_walk_matches_oracle(gpu_device, oracle, name):
    """search3_box2_kernel (search version 6): the rectified box walk with two image columns per lane - pair sums through
    one prefix sum, S12 = R(l) - Q(l - 6) / Q(l) - R(l - 5) - computes the one-column walk's integers, so both directions'
    grids are the oracle's bit for bit: ordinary pairs, the adversarial inputs of the decision rule (exact ties: contender
    lists through the queue, whole-corridor pixels, tiles on the 58-wide work-list entries), with and without the candidate
    counter (whose count must be the oracle's), and a larger pair with disparity discontinuities against version 3."""
    if name == "big":
        a, b, _ = synth.make_pair(1100, 700, seed=77)
        big = {"img1": a, "img2": b, "F": synth.F_HORIZONTAL, "projection": 0, "steps": synth.optimal_scale_steps(1100, 700)}
        cnt6, cnt3 = {}, {}
        v6, v3 = run_gpu(gpu_device, big, version=6, both=True, counters=cnt6), run_gpu(gpu_device, big, version=3, both=True, counters=cnt3)
        assert_same_grid(v6[0], v3[0], "1100x700 v6 vs v3 forward")
        assert_same_grid(v6[1], v3[1], "1100x700 v6 vs v3 reverse")
        assert cnt6["candidates"] == cnt3["candidates"]
        assert_same_grid(run_gpu(gpu_device, big, version=6), run_gpu(gpu_device, big, version=3), "1100x700 v6 vs v3, default mode")
        return
    c = adversarial_case(name) if name in ("periodic", "periodic16", "low_contrast", "noisy") else cases.make_case(name)
    want = run_oracle(oracle, c, both=True)
    cnt = {}
    got = run_gpu(gpu_device, c, version=6, both=True, counters=cnt)
    assert_same_grid(got[0], want[0], f"{name} v6 forward")
    assert_same_grid(got[1], want[1], f"{name} v6 reverse")
    cnt3 = {}
    run_gpu(gpu_device, c, version=3, both=True, counters=cnt3)
    assert cnt["candidates"] == cnt3["candidates"], (cnt, cnt3)
    again = run_gpu(gpu_device, c, version=6, both=True)
    assert_same_grid(again[0], want[0], f"{name} v6 forward, no counters")
    assert_same_grid(again[1], want[1], f"{name} v6 reverse, no counters")
    default = run_gpu(gpu_device, c, version=6)   # scores of the observable pass only
    assert_same_grid(default, want[0], f"{name} v6 forward, default score mode")


@pytest.mark.parametrize("name", ["tilt3_200x150", "persp_240x180", "ragged_dims"])
def test_fused_level_calls_equal_independent_calls(gpu_device, oracle, name):
    """cvhip_ctx_set_fuse_level_calls: the reference's four calls per level, in the reference's order, executed as one
    level call (forward call: images in + statistics; reverse call: both search passes; second cross-check call: both
    filters) give the grids of the independent calls and of the oracle - host images through the upload ring, device
    images copied and borrowed, both directions' grids, several pairs on one context's parked buffers."""
    import torch

    c = cases.make_case(name)
    want = run_oracle(oracle, c, both=True)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pads = [[torch.zeros(l.size + 64, dtype=torch.uint8, device="cuda") for l in p] for p in (p1, p2)]
    dev_p = []
    for p, bufs in zip((p1, p2), pads):
        lv = []
        for l, b in zip(p, bufs):
            b[:l.size].copy_(torch.from_numpy(l.reshape(-1)))
            lv.append(b[:l.size].view(l.shape[0], l.shape[1]))
        dev_p.append(lv)
    torch.cuda.synchronize()
    for mode in ("host", "device", "borrowed", "host again"):
        pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"], correlation.ProjectionMode(c["projection"]))
        pc.set_exact_scores(True)
        pc.set_fuse_level_calls(True)
        a, b = (p1, p2) if mode.startswith("host") else dev_p
        if mode == "borrowed":
            pc.set_borrow_inputs(True)
            pc.set_stats_ahead(True)
        try:
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(a[k], b[k], 1.0 / float(1 << k), fused=False)
            got = (pc.complete(correlation.CorrelationDirection.Forward), pc.complete(correlation.CorrelationDirection.Reverse))
        finally:
            pc.close()
        assert_same_grid(got[0], want[0], f"{name} fused calls ({mode}) forward")
        assert_same_grid(got[1], want[1], f"{name} fused calls ({mode}) reverse")


def test_host_images_into_fresh_contexts(gpu_device, oracle):
    """Host level images are uploaded on the handle's copy stream into a pool that the context's creation clears on the
    CONTEXT's stream: the first upload must wait for that clearing (a race that a randomised sweep caught in 13 of 250
    cases - whole grids wrong - and no fixed-size test did: contexts of dimensions seen before reuse parked buffers whose
    clearing is long done).  Fresh dimensions every time, host images, level call and four fused calls."""
    rng = np.random.default_rng(7)
    for it in range(14):
        w, h = int(rng.integers(96, 260)), int(rng.integers(96, 200))
        a, b, _ = synth.make_pair(w, h, seed=100 + it)
        c = {"img1": a, "img2": b, "F": synth.F_HORIZONTAL, "projection": 0, "steps": synth.optimal_scale_steps(w, h)}
        want = run_oracle(oracle, c)
        p1, p2 = cases.pyramids(c)
        pc = correlation.PointCorrelations(gpu_device, (w, h), (w, h), c["F"])
        try:
            if it & 1:
                pc.set_fuse_level_calls(True)
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=not (it & 1))
            assert_same_grid(pc.complete(), want, f"fresh context {w}x{h}")
        finally:
            pc.close()


@pytest.mark.parametrize("bands", [1, 2])
def test_fused_level_calls_out_of_order(gpu_device, oracle, bands):
    """A caller that has promised the reference's call order and departs from it still gets every call executed: grids
    read between the calls (complete() after each), a forward call whose reverse call never comes, a reverse call with
    OTHER images than the forward call's, cross-checks in the other order.  With result bands the last level's launches
    are held back until its second filter call: every departure runs what was asked for so far."""
    c = cases.make_case("sem320x200")
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    F, D = correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    pc.set_exact_scores(True)
    pc.set_fuse_level_calls(True)
    pc.set_result_bands(bands)
    oc = oracle.Corr((w1, h1), (w2, h2), c["F"], 0, 8)
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            s = 1.0 / float(1 << k)
            variant = (i + bands - 1) % 3
            pc.correlate_images_step(p1[k], p2[k], s, F)
            oc.step(p1[k], p2[k], s, 0)
            if variant == 0:   # the grid is read before the reverse call: the pending forward pass runs alone
                assert_same_grid(pc.complete(F), oc.get(0), f"level {k} fwd search (read early)")
            if variant == 1:   # the reverse call passes a COPY of the images: not recognised as the level's reverse call
                pc.correlate_images_step(p2[k].copy(), p1[k].copy(), s, D)
            else:
                pc.correlate_images_step(p2[k], p1[k], s, D)
            oc.step(p2[k], p1[k], s, 1)
            assert_same_grid(pc.complete(D), oc.get(1), f"level {k} rev search")
            if variant == 2:   # reverse filter first
                pc.cross_check_filter(s, D)
                oc.cross_check(s, 1)
                pc.cross_check_filter(s, F)
                oc.cross_check(s, 0)
            else:
                pc.cross_check_filter(s, F)
                oc.cross_check(s, 0)
                if variant == 0:
                    assert_same_grid(pc.complete(F), oc.get(0), f"level {k} fwd cross-check (read early)")
                pc.cross_check_filter(s, D)
                oc.cross_check(s, 1)
            assert_same_grid(pc.complete(F), oc.get(0), f"level {k} fwd")
            assert_same_grid(pc.complete(D), oc.get(1), f"level {k} rev")
            pc.first_pass = False
            oc.end_level()
        # a forward call that is never followed by anything but the context's destruction
        pc.first_pass = True
        pc.correlate_images_step(p1[c["steps"]], p2[c["steps"]], 1.0 / float(1 << c["steps"]), F)
    finally:
        pc.close()
        oc.close()


@pytest.mark.parametrize("name,bands,expect", [("sem320x200", 2, 2), ("sem320x200", 16, 2), ("h256", 2, 2), ("ragged_dims", 3, None),
                                               ("tilt3_200x150", 2, None), ("persp_240x180", 4, 1)])
def test_result_bands_give_the_same_grid(gpu_device, oracle, name, bands, expect):
    """cvhip_ctx_set_result_bands: the last level searched and filtered in row bands, each expanded and copied out to the
    HOST destination on the copy stream under the search of the bands behind it - the grid (and the reverse grid) of the
    unbanded level, i.e. the oracle's, bit for bit; through the level call and through the reference's four calls (host
    images); into device destinations (no banded copy); on a context used for a second pair.  Geometries that are not
    row-local (perspective pairs, tilted lines beyond the bound) and bands too low for the filter's reach fall back to
    fewer bands or none, which cvhip_ctx_get_result_bands reports."""
    import torch

    c = cases.make_case(name)
    want = run_oracle(oracle, c, both=True)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    F, D = correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse
    for mode in ("level call", "four calls", "device destination", "level call, pinned"):
        pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"], correlation.ProjectionMode(c["projection"]))
        pc.set_exact_scores(True)
        pc.set_result_bands(bands)
        if mode == "four calls":
            pc.set_fuse_level_calls(True)
        try:
            for rep in range(2):
                pc.first_pass = True
                for i in range(c["steps"] + 1):
                    k = c["steps"] - i
                    pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=mode != "four calls")
                live = pc.result_bands()
                if expect is not None:
                    assert live == expect, (mode, live)
                if mode == "device destination":
                    xy = torch.empty((h1, w1, 2), dtype=torch.int32, device="cuda")
                    co = torch.empty((h1, w1), dtype=torch.float32, device="cuda")
                    pc.complete(F, xy, co)
                    torch.cuda.synchronize()
                    got = (xy.cpu().numpy(), co.cpu().numpy())
                elif mode.endswith("pinned"):
                    xy = torch.empty((h1, w1, 2), dtype=torch.int32).pin_memory()
                    co = torch.empty((h1, w1), dtype=torch.float32).pin_memory()
                    xy.fill_(7)
                    pc.complete(F, xy, co)
                    got = (xy.numpy().copy(), co.numpy().copy())
                else:
                    got = pc.complete(F)
                assert_same_grid(got, want[0], f"{name} {bands} bands ({mode}, pair {rep}) forward")
                assert_same_grid(pc.complete(D), want[1], f"{name} {bands} bands ({mode}, pair {rep}) reverse")
                assert_same_grid(pc.complete(F), want[0], f"{name} {bands} bands ({mode}, pair {rep}) forward again")
                # the 8-byte cells of cvhip_complete_packed: the same grid
                cells, co = pc.complete_packed(F)
                assert_same_grid((pc.unpack_cells(cells), co), want[0], f"{name} {bands} bands ({mode}, pair {rep}) packed")
        finally:
            pc.close()


def test_result_bands_with_async_readback_on_one_context(gpu_device):
    """Result bands requested together with cvhip_ctx_set_async_readback on ONE context over four pairs (the
    band expansions read the level's planes on the copy stream while the context's stream may already rewrite them for the
    next pair).  Under asynchronous readback the level is not banded, and a banded grid completed after the mode was
    switched on is ordered by an event behind its last expansion: every pair's page-locked grid must be the synchronous
    one, bit for bit.  Two different pairs alternate so that a stale plane would show."""
    import torch

    pairs = []
    for seed in (5, 6):
        a, b, _ = synth.make_pair(1280, 1024, seed=seed)
        c = {"img1": a, "img2": b, "F": synth.F_HORIZONTAL, "projection": 0, "steps": synth.optimal_scale_steps(1280, 1024)}
        pairs.append((c, cases.pyramids(c), run_gpu(gpu_device, c)))
    for switch_late in (False, True):
        pc = correlation.PointCorrelations(gpu_device, (1280, 1024), (1280, 1024), synth.F_HORIZONTAL)
        pc.set_result_bands(4)
        host = [(torch.empty((1024, 1280, 2), dtype=torch.int32).pin_memory(), torch.empty((1024, 1280), dtype=torch.float32).pin_memory())
                for _ in range(4)]
        try:
            if not switch_late:
                pc.set_async_readback(True)
            for it in range(4):
                c, (p1, p2), _ = pairs[it & 1]
                pc.first_pass = True
                for i in range(c["steps"] + 1):
                    k = c["steps"] - i
                    pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
                assert pc.result_bands() == (4 if switch_late else 1)
                if switch_late:
                    pc.set_async_readback(True)   # the level went out in bands; the completion is asynchronous
                host[it][0].fill_(7)
                pc.complete(out_xy=host[it][0].numpy(), out_corr=host[it][1].numpy())
                if switch_late:
                    pc.set_async_readback(False)
            gpu_device.synchronize()
            for it in range(4):
                want = pairs[it & 1][2]
                valid = want[0][..., 0] >= 0
                assert (host[it][0].numpy() == want[0]).all(), (switch_late, it)
                assert (host[it][1].numpy().view(np.uint32)[valid] == want[1].view(np.uint32)[valid]).all(), (switch_late, it)
        finally:
            pc.close()


@pytest.mark.parametrize("tilt", [0.0, 1.5])
def test_result_bands_large_pair(gpu_device, tilt):
    """Four and eight result bands on a 1536 x 1280 pair with disparity discontinuities against the unbanded level (both on
    the device: the oracle would take minutes), host destinations.  Tilted by 1.5 degrees the bands are the stepped
    launches', one per direction with the second on the handle's side stream (levels from 1024^2), and the filter's reach
    is ~50 rows."""
    a, b, _ = synth.make_pair(1536, 1280, seed=91, tilt_deg=tilt)
    c = {"img1": a, "img2": b, "F": synth.f_tilt(tilt) if tilt else synth.F_HORIZONTAL, "projection": 0,
         "steps": synth.optimal_scale_steps(1536, 1280)}
    want = run_gpu(gpu_device, c, both=True)
    p1, p2 = cases.pyramids(c)
    for bands in ((4, 8, 0) if tilt == 0.0 else (3, 5)):   # (0: the library's choice - bands of half a megapixel or more: 3 here)
        pc = correlation.PointCorrelations(gpu_device, (1536, 1280), (1536, 1280), c["F"])
        pc.set_exact_scores(True)
        pc.set_result_bands(bands)
        pc.set_fuse_level_calls(bands in (8, 5))
        try:
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=bands in (4, 3, 0))
            assert pc.result_bands() == (bands if bands else 3)
            cells, co = pc.complete_packed()
            assert_same_grid((pc.unpack_cells(cells), co), want[0], f"{bands} bands forward, packed")
            assert_same_grid(pc.complete(), want[0], f"{bands} bands forward")
            assert_same_grid(pc.complete(correlation.CorrelationDirection.Reverse), want[1], f"{bands} bands reverse")
        finally:
            pc.close()


def test_each_level_matches_oracle(gpu_device, oracle):
    """Stage-by-stage: after every search pass and every cross-check the device grids equal
    the oracle's (catches compensating errors that a final-grid comparison could hide)."""
    c = cases.make_case("sem320x200")
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    F, D = correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    pc.set_exact_scores(True)  # coarser levels and the reverse grid are read here
    oc = oracle.Corr((w1, h1), (w2, h2), c["F"], 0, 8)
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            s = 1.0 / float(1 << k)
            pc.correlate_images_step(p1[k], p2[k], s, F)
            oc.step(p1[k], p2[k], s, 0)
            assert_same_grid(pc.complete(F), oc.get(0), f"level {k} fwd search")
            pc.correlate_images_step(p2[k], p1[k], s, D)
            oc.step(p2[k], p1[k], s, 1)
            assert_same_grid(pc.complete(D), oc.get(1), f"level {k} rev search")
            pc.cross_check_filter(s, F)
            oc.cross_check(s, 0)
            assert_same_grid(pc.complete(F), oc.get(0), f"level {k} fwd cross-check")
            cells, co = pc.complete_packed(F)  # (coarser levels: the coordinates scaled back, as complete()'s)
            assert_same_grid((pc.unpack_cells(cells), co), oc.get(0), f"level {k} fwd cross-check, packed cells")
            pc.cross_check_filter(s, D)
            oc.cross_check(s, 1)
            assert_same_grid(pc.complete(D), oc.get(1), f"level {k} rev cross-check")
            pc.first_pass = False
            oc.end_level()
    finally:
        pc.close()
        oc.close()


@pytest.mark.parametrize("name", ["h256", "sem320x200", "tilt3_200x150", "persp_240x180", "vert_200x260", "tilt60_150x200"])
@pytest.mark.parametrize("version", [2, 3])
def test_default_score_mode_positions_exact_everywhere(gpu_device, oracle, name, version):
    """Default mode (cvhip_ctx_set_exact_scores off): passes whose scores cannot reach the caller skip the exact chain
    for pixels with one clear contender.  The forward grid - positions and score bits - and the reverse POSITIONS
    must still be the oracle's; the reverse scores (never observable in the reference, mod.rs:208-215) are not computed
    and are reported as NaN.  Far fewer exact evaluations are spent."""
    c = cases.make_case(name)
    want_f, want_r = run_oracle(oracle, c, both=True)
    cnt, cnt_all = {}, {}
    got_f, got_r = run_gpu(gpu_device, c, both=True, version=version, counters=cnt, exact_scores=False)
    run_gpu(gpu_device, c, both=True, version=version, counters=cnt_all, exact_scores=True)
    assert_same_grid(got_f, want_f, f"{name} forward, default score mode")
    assert (got_r[0] == want_r[0]).all(), f"{name}: reverse match positions differ in default score mode"
    assert np.isnan(got_r[1]).all()
    assert cnt["candidates"] == cnt_all["candidates"]
    assert cnt["exact_evals"] <= cnt_all["exact_evals"]
    if name in ("h256", "sem320x200", "tilt3_200x150"):  # (steep cases spend theirs in the whole-corridor fallback)
        assert cnt["exact_evals"] < 0.75 * cnt_all["exact_evals"]


def test_candidate_counter_matches_oracle(gpu_device, oracle):
    c = cases.make_case("h256")
    p1, p2 = cases.pyramids(c)
    pc = correlation.PointCorrelations(gpu_device, (256, 256), (256, 256), c["F"])
    try:
        pc.set_profiling(True, True)
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        prof = pc.get_profile()
    finally:
        pc.close()
    _, _, cand = oracle.correlate_dense(p1, p2, c["F"], 0, 8)
    assert prof["candidates"] == cand
    # one search launch per level: the forward and the reverse pass of a level share their launches
    assert prof["launches"] == c["steps"] + 1 and prof["search_ms"] > 0.0


@pytest.mark.parametrize("name", ["h256", "sem320x200", "tilt3_200x150", "persp_240x180", "ragged_dims", "vert_200x260",
                                  "flat"])
def test_search_range_sum_path_equals_chain(gpu_device, oracle, name):
    """estimate_search_range from integer box sums (range mode 0, the default) against the f64 chain alone (=1)
    and against the test hooks that send every third block of a sum tile through the chain, over the staged tile
    (=2) or over global memory (=3): the same
    matches, scores and - because every corridor bound enters it - the same candidate count as the oracle."""
    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    _, _, cand = oracle.correlate_dense(p1, p2, c["F"], c["projection"], 8)
    want = run_oracle(oracle, c, both=True)
    for mode in (0, 1, 2, 3):
        cnt = {}
        got = run_gpu(gpu_device, c, both=True, counters=cnt, range_mode=mode)
        assert_same_grid(got[0], want[0], f"{name} range mode {mode} forward")
        assert_same_grid(got[1], want[1], f"{name} range mode {mode} reverse")
        assert cnt["candidates"] == cand, f"{name} range mode {mode}"


@pytest.mark.parametrize("tilt", [0.0, 3.0, 90.0])
def test_search_range_sum_path_equals_chain_1024(gpu_device, tilt):
    a, b, _ = synth.make_pair(1024, 1024, seed=53, tilt_deg=tilt, sem_style=tilt == 0.0)
    steps = synth.optimal_scale_steps(1024, 1024)
    c = dict(img1=a, img2=b, F=synth.f_tilt(tilt), projection=0, steps=steps)
    res = {}
    for mode in (1, 0, 2, 3):
        cnt = {}
        res[mode] = run_gpu(gpu_device, c, both=True, counters=cnt, range_mode=mode), cnt["candidates"]
    for mode in (0, 2, 3):
        for d in (0, 1):
            assert_same_grid(res[mode][0][d], res[1][0][d], f"tilt {tilt} range mode {mode} dir {d}")
        assert res[mode][1] == res[1][1] and res[mode][1] > 100_000_000


def test_device_resident_inputs_and_outputs(gpu_device, oracle):
    """Images and result buffers may live in HBM (torch tensors): same bits as host buffers."""
    import torch

    c = cases.make_case("tilt3_200x150")
    p1, p2 = cases.pyramids(c)
    d1 = [torch.from_numpy(p).cuda() for p in p1]
    d2 = [torch.from_numpy(p).cuda() for p in p2]
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
        oxy = torch.empty((h1, w1, 2), dtype=torch.int32, device="cuda")
        oc = torch.empty((h1, w1), dtype=torch.float32, device="cuda")
        pc.complete(out_xy=oxy, out_corr=oc)
        gpu_device.synchronize()  # device destinations are written in stream order on the context's own stream
        got = (oxy.cpu().numpy(), oc.cpu().numpy())
    finally:
        pc.close()
    assert_same_grid(got, run_oracle(oracle, c), "device-resident")


def test_borrowed_device_inputs(gpu_device, oracle):
    """cvhip_ctx_set_borrow_inputs: padded device images are used in place (no staging copy) - same bits."""
    import torch

    c = cases.make_case("ragged_dims")
    p1, p2 = cases.pyramids(c)

    def resident(p):
        buf = torch.zeros(p.size + 64, dtype=torch.uint8, device="cuda")
        buf[:p.size].copy_(torch.from_numpy(p).reshape(-1))
        return buf[:p.size].view(p.shape[0], p.shape[1])

    d1, d2 = [resident(p) for p in p1], [resident(p) for p in p2]
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    try:
        pc.set_borrow_inputs(True)
        torch.cuda.synchronize()
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
        got = pc.complete()
    finally:
        pc.close()
    assert_same_grid(got, run_oracle(oracle, c), "borrowed inputs")


def test_stats_ahead_same_bits(gpu_device, oracle):
    """cvhip_ctx_set_stats_ahead: the window statistics of borrowed level images run on a side stream under the coarse
    levels' search - the grid is the oracle's, also when one context runs two pyramids back to back."""
    import torch

    c = cases.make_case("ragged_dims")
    p1, p2 = cases.pyramids(c)

    def resident(p):
        buf = torch.zeros(p.size + 64, dtype=torch.uint8, device="cuda")
        buf[:p.size].copy_(torch.from_numpy(p).reshape(-1))
        return buf[:p.size].view(p.shape[0], p.shape[1])

    d1, d2 = [resident(p) for p in p1], [resident(p) for p in p2]
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    want = run_oracle(oracle, c)
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    try:
        pc.set_borrow_inputs(True)
        pc.set_stats_ahead(True)
        torch.cuda.synchronize()
        for run in range(2):
            pc.first_pass = True  # (a new pyramid run on the same context)
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            assert_same_grid(pc.complete(), want, f"stats ahead, run {run}")
    finally:
        pc.close()


def test_error_reporting(gpu_device):
    from cybervision_amd._lib import CvhipError

    a, b, _ = synth.make_pair(128, 128)
    pc = correlation.PointCorrelations(gpu_device, (128, 128), (128, 128), synth.F_HORIZONTAL)
    try:
        with pytest.raises(CvhipError) as ei:  # 1/3 is not a power of two
            pc.correlate_images(a, b, 1.0 / 3.0)
        assert ei.value.code == -3 and "2^-k" in str(ei.value)
        with pytest.raises(CvhipError) as ei:  # wrong level dims for the scale
            pc.correlate_images(a, b, 0.5)
        assert ei.value.code == -3
        pc.first_pass = False
        with pytest.raises(CvhipError) as ei:  # refinement without a previous level
            pc.correlate_images(a, b, 1.0)
        assert ei.value.code == -1
        with pytest.raises(CvhipError):
            pc.cross_check_filter(1.0, correlation.CorrelationDirection.Forward)
    finally:
        pc.close()
    with pytest.raises(CvhipError):
        correlation.PointCorrelations(gpu_device, (8, 8), (128, 128), synth.F_HORIZONTAL)


def test_config2_1024_sem_matches_oracle_and_properties(gpu_device, oracle):
    """BASELINE config 2 (1024^2 SEM-style pair, 5 levels) at full size: bit-exact against the oracle run on all host
    cores (a couple of seconds on the GPU box), both directions; plus the size-independent properties: determinism,
    known-disparity recovery, cross-check consistency, border emptiness."""
    import os

    a, b, d = synth.make_pair(1024, 1024, sem_style=True)
    steps = synth.optimal_scale_steps(1024, 1024)
    c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=0, steps=steps)
    (fxy, fc), (rxy, rc) = run_gpu(gpu_device, c, both=True)
    (fxy2, fc2) = run_gpu(gpu_device, c)
    assert (fxy == fxy2).all() and (bits(fc) == bits(fc2)).all(), "not deterministic"
    p1, p2 = cases.pyramids(c)
    oc = oracle.Corr((1024, 1024), (1024, 1024), c["F"], 0, os.cpu_count())
    try:
        for i in range(steps + 1):
            k = steps - i
            oc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        assert_same_grid((fxy, fc), oc.get(0), "1024 SEM forward")
        assert_same_grid((rxy, rc), oc.get(1), "1024 SEM reverse")
    finally:
        oc.close()
    valid = fxy[..., 0] >= 0
    assert valid.mean() > 0.6
    assert not valid[:5].any() and not valid[-5:].any() and not valid[:, :5].any() and not valid[:, -5:].any()
    ys, xs = np.nonzero(valid)
    x2, y2 = fxy[..., 0][valid], fxy[..., 1][valid]
    assert ((np.abs(x2 + d[y2, x2] - xs)) <= 1).mean() > 0.95
    assert (fc[valid] >= np.float32(0.6)).all() and (fc[valid] <= np.float32(1.0001)).all()
    # every surviving forward match has a reverse match pointing back within +-4 (mod.rs:588-624)
    rng = np.random.default_rng(1)
    for i in rng.choice(len(xs), size=500, replace=False):
        x, y, mx, my = xs[i], ys[i], x2[i], y2[i]
        win = rxy[max(my - 4, 0):my + 5, max(mx - 4, 0):mx + 5].reshape(-1, 2)
        win = win[win[:, 0] >= 0]
        assert ((np.abs(win[:, 0] - x) <= 4) & (np.abs(win[:, 1] - y) <= 4)).any()


@pytest.mark.parametrize("name", ["tilt3_200x150", "vert_200x260", "h256", "tilt60_150x200"])
def test_row_sharded_equals_unsharded(gpu_device, oracle, name):
    """Two row-sharded contexts (shard 0/2 and 1/2) on one GPU, bands exchanged between their
    level grids after every search pass exactly as the RCCL all-gather would: the result must be
    bit-identical to the unsharded run (and so to the oracle).  The cases cover the stepped, transposed and
    lean box-filter instantiations and the candidate filter, each on a row range that does not start at 0."""
    import torch

    from cybervision_amd import sharding

    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    F, R = correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse
    ctxs = [correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"]) for _ in range(2)]
    try:
        for r, pc in enumerate(ctxs):
            pc.set_row_shard(r, 2)

        def exchange(direction):
            grids = [pc.level_grid(direction) for pc in ctxs]
            gpu_device.synchronize()
            for r, g in enumerate(grids):
                assert (g["row0"], g["row1"]) == sharding.shard_rows(g["lh"], r, 2)
                sharding.copy_rows(grids[1 - r], g, g["row0"], g["row1"])  # both planes (match words, scores)
            torch.cuda.synchronize()

        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            s = 1.0 / float(1 << k)
            for pc in ctxs:
                pc.correlate_images_step(p1[k], p2[k], s, F)
            exchange(F)
            for pc in ctxs:
                pc.correlate_images_step(p2[k], p1[k], s, R)
            exchange(R)
            for pc in ctxs:
                pc.cross_check_filter(s, F)
                pc.cross_check_filter(s, R)
                pc.first_pass = False
        want = run_oracle(oracle, c)
        for r, pc in enumerate(ctxs):
            assert_same_grid(pc.complete(F), want, f"shard context {r}")
    finally:
        for pc in ctxs:
            pc.close()


def test_sharded_level_call_with_gather_hook(gpu_device, oracle):
    """cvhip_correlate_level on a 1-of-1... n-shard context drives the gather hook itself; with a
    single participating rank emulated by den=1 semantics the hook must not be needed, and with
    den=2 but no hook the call must fail loudly on a level large enough to be sharded."""
    from cybervision_amd._lib import CvhipError

    a, b, _ = synth.make_pair(256, 256)
    steps = synth.optimal_scale_steps(256, 256)
    p1, p2 = synth.box_pyramid(a, steps), synth.box_pyramid(b, steps)
    pc = correlation.PointCorrelations(gpu_device, (256, 256), (256, 256), synth.F_HORIZONTAL)
    try:
        pc.set_row_shard(0, 2)                       # no hook
        pc.correlate_images(p1[2], p2[2], 0.25)      # 64 rows: below the sharding threshold, computed whole
        pc.correlate_images(p1[1], p2[1], 0.5)       # 128 rows / 2 = 64 per shard: sharded -> needs the hook
    except CvhipError as e:
        assert e.code == -1 and "all-gather hook" in str(e)
    else:
        raise AssertionError("sharded level without a gather hook must fail")
    finally:
        pc.close()


def test_progress_callback_and_wide_sharding(gpu_device, oracle):
    """(1) The progress hook (ProgressListener::report_status, gpu/mod.rs:241-249) is invoked synchronously with
    values in [0, 1] that follow the reference's mapping (forward pass in the lower half, reverse in the upper), for
    the per-pass calls and for the fused level call, and it does not change the result.  (2) Row sharding with more
    shards than any test so far (den = 5, ragged bands) through the host hook on a private-stream device: the
    library fences both sides of the hook itself."""
    from cybervision_amd import sharding

    c = cases.make_case("sem320x200")
    p1, p2 = cases.pyramids(c)
    want = run_oracle(oracle, c)
    F, R = correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse
    for fused in (True, False):
        seen = []
        pc = correlation.PointCorrelations(gpu_device, (320, 200), (320, 200), c["F"])
        try:
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), progress=seen.append, fused=fused)
            assert_same_grid(pc.complete(F), want, f"progress fused={fused}")
        finally:
            pc.close()
        assert len(seen) >= 3 * (c["steps"] + 1) and all(0.0 <= v <= 1.0 for v in seen)
        assert any(v < 0.5 for v in seen) and any(v > 0.5 for v in seen) and max(seen) > 0.99
    # den = 5 through the fused level call and the host hook, on this fixture's PRIVATE-stream device (the library
    # fences both sides of the hook): rank r of 5 searches only its band; the hook supplies the other four bands from
    # the level grids of an unsharded run captured after each search pass (what the other ranks would have sent)
    import torch

    ba, bb_, _ = synth.make_pair(640, 400, seed=13)
    big = dict(img1=ba, img2=bb_, F=synth.f_tilt(0.5), projection=0, steps=synth.optimal_scale_steps(640, 400))
    q1, q2 = cases.pyramids(big)
    captured = {}
    ref = correlation.PointCorrelations(gpu_device, (640, 400), (640, 400), big["F"])
    try:
        for i in range(big["steps"] + 1):
            k = big["steps"] - i
            s = 1.0 / float(1 << k)
            for d, (a, b) in ((F, (q1[k], q2[k])), (R, (q2[k], q1[k]))):
                ref.correlate_images_step(a, b, s, d)
                g = ref.level_grid(d)
                gpu_device.synchronize()
                captured[(k, int(d))] = sharding.alias_bytes(g["cells"], g["lh"] * g["lw"] * 4, device=True).clone()
                if d == F:  # direction 2 of the hook: the forward score plane (gathered at scale 1 only)
                    captured[(k, 2)] = sharding.alias_bytes(g["scores"], g["lh"] * g["lw"] * 4, device=True).clone()
            ref.cross_check_filter(s, F)
            ref.cross_check_filter(s, R)
            ref.first_pass = False
    finally:
        ref.close()
    want_big = run_oracle(oracle, big)
    for rank in (0, 3, 4):
        pc = correlation.PointCorrelations(gpu_device, (640, 400), (640, 400), big["F"])
        calls = []
        try:
            level = {"k": None}

            def hook(cells_ptr, shard_bytes, n_shards, direction):
                assert n_shards == 5
                full = captured[(level["k"], direction)]
                mine_lo, mine_hi = rank * shard_bytes, min((rank + 1) * shard_bytes, full.numel())
                dst = sharding.alias_bytes(cells_ptr, full.numel(), device=True)
                keep = dst[mine_lo:mine_hi].clone()          # this rank's own band stays what IT computed
                dst.copy_(full)
                dst[mine_lo:mine_hi].copy_(keep)
                torch.cuda.synchronize()
                calls.append((level["k"], direction))

            pc.set_row_shard(rank, 5, hook)
            for i in range(big["steps"] + 1):
                level["k"] = big["steps"] - i
                pc.correlate_images(q1[level["k"]], q2[level["k"]], 1.0 / float(1 << level["k"]))
            assert_same_grid(pc.complete(F), want_big, f"rank {rank} of 5")
        finally:
            pc.close()
        # only the full-resolution level has >= 64 rows per shard (400 // 5 = 80): its two match planes, and - scale 1 - the
        # forward score plane
        assert calls == [(0, 0), (0, 2), (0, 1)]


@pytest.mark.parametrize("case,mode", [("persp_240x180", "gather"), ("tilt3_200x150", "band")])
def test_two_rank_sharded_level_calls(case, mode):
    """End-to-end N = 2: two processes (sharing this box's one GPU) row-shard every search pass,
    cvhip_correlate_level drives the all-gather hook itself (gloo staged through the host here; RCCL
    on a real multi-GPU node), and both ranks must reproduce the golden grid bit for bit."""
    import os
    import socket
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(root / "tests" / "_shard_gpu_worker.py"), case, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, out


@pytest.mark.parametrize("tilt", [3.0, -2.0, 45.0, 87.0, 90.0])
def test_box_filter_variants_equal_exact_kernel_1024(gpu_device, tilt):
    """The stepped-line and transposed instantiations of the box filter (slightly tilted and near-vertical
    epipolar lines, pairs displaced along those lines) against the plain exact kernel at a size the oracle
    would need minutes for: same match coordinates and score bits, both directions."""
    a, b, _ = synth.make_pair(1024, 1024, seed=31, tilt_deg=tilt)
    steps = synth.optimal_scale_steps(1024, 1024)
    c = dict(img1=a, img2=b, F=synth.f_tilt(tilt), projection=0, steps=steps)
    cnt = {}
    (fxy, fc), (rxy, rc) = run_gpu(gpu_device, c, both=True, counters=cnt)
    (fxy1, fc1), (rxy1, rc1) = run_gpu(gpu_device, c, both=True, version=1)
    assert (fxy == fxy1).all() and (rxy == rxy1).all()
    vf, vr = fxy1[..., 0] >= 0, rxy1[..., 0] >= 0
    assert (bits(fc)[vf] == bits(fc1)[vf]).all() and (bits(rc)[vr] == bits(rc1)[vr]).all()
    assert vf.mean() > 0.7 and cnt["candidates"] > 100_000_000
    # the instantiations that do not count candidates (the timed ones) differ where a line runs within an ulp below a row 2^k -
    # at exactly 45 degrees every candidate of such a pixel: they walk six planes for it (box_body.inc, "loose" lanes)
    (fxy2, fc2), (rxy2, rc2) = run_gpu(gpu_device, c, both=True)
    assert (fxy2 == fxy1).all() and (rxy2 == rxy1).all()
    assert (bits(fc2)[vf] == bits(fc1)[vf]).all() and (bits(rc2)[vr] == bits(rc1)[vr]).all()


def test_box_filter_perspective_parameter_set_equals_exact_kernel_1024(gpu_device):
    """Nine stripes, threshold 0.5 (the perspective parameter set, mod.rs:127-133) on a rectified pair: the box
    filter then runs its second plane group (planes 5..8); against the plain exact kernel, both directions."""
    a, b, _ = synth.make_pair(1024, 1024, seed=41, sem_style=True)
    steps = synth.optimal_scale_steps(1024, 1024)
    c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=1, steps=steps)
    (fxy, fc), (rxy, rc) = run_gpu(gpu_device, c, both=True)
    (fxy1, fc1), (rxy1, rc1) = run_gpu(gpu_device, c, both=True, version=1)
    assert (fxy == fxy1).all() and (rxy == rxy1).all()
    vf, vr = fxy1[..., 0] >= 0, rxy1[..., 0] >= 0
    assert (bits(fc)[vf] == bits(fc1)[vf]).all() and (bits(rc)[vr] == bits(rc1)[vr]).all()
    assert vf.mean() > 0.7


def test_full_size_4096_filters_equal_exact_kernel():
    """BASELINE's 4096^2 pair: the filter + exact-re-evaluation searches (v3 box filter, v2) must reproduce, bit for bit,
    the plain kernel that sends every one of the 3.2e9 candidates through the reference's serial f32
    chain (v1, itself bit-exact against the oracle on every small case) — match coordinates and scores,
    both directions.  Plus the size-independent properties: known disparity recovered, empty border."""
    from cybervision_amd import correlation as corr_mod

    dev = corr_mod.create_gpu_context()
    try:
        a, b, d = synth.make_pair(4096, 4096)
        steps = synth.optimal_scale_steps(4096, 4096)
        c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=0, steps=steps)
        cnt = {}
        (fxy, fc), (rxy, rc) = run_gpu(dev, c, both=True, version=3, counters=cnt)
        (fxy2, fc2), (rxy2, rc2) = run_gpu(dev, c, both=True, version=2)
        (fxy1, fc1), (rxy1, rc1) = run_gpu(dev, c, both=True, version=1)
    finally:
        dev.close()
    assert (fxy == fxy1).all() and (rxy == rxy1).all(), "v3 match coordinates differ from the exact kernel"
    assert (fxy2 == fxy1).all() and (rxy2 == rxy1).all(), "v2 match coordinates differ from the exact kernel"
    vf, vr = fxy1[..., 0] >= 0, rxy1[..., 0] >= 0
    assert (bits(fc)[vf] == bits(fc1)[vf]).all() and (bits(rc)[vr] == bits(rc1)[vr]).all(), "v3 scores differ"
    assert (bits(fc2)[vf] == bits(fc1)[vf]).all() and (bits(rc2)[vr] == bits(rc1)[vr]).all(), "v2 scores differ"
    assert cnt["candidates"] > 3_000_000_000 and cnt["exact_evals"] < 0.02 * cnt["candidates"]
    assert vf.mean() > 0.8
    assert not vf[:5].any() and not vf[-5:].any() and not vf[:, :5].any() and not vf[:, -5:].any()
    ys, xs = np.nonzero(vf)
    x2, y2 = fxy[..., 0][vf], fxy[..., 1][vf]
    assert (np.abs(x2 + d[y2, x2] - xs) <= 1).mean() > 0.97


def test_config3_4096_matches_oracle_digest(gpu_device):
    """BASELINE config 3 (the 4096^2 headline pair, 7 levels) against the ORACLE at full size: SHA-256 of the forward
    and reverse match planes and score planes, generated by tests/tools/gen_digest_4096.py from oracle/cvref_corr.c
    (its result is independent of the thread count and of the machine) and committed under tests/golden/."""
    import hashlib
    import json
    from pathlib import Path

    want = json.loads((Path(__file__).parent / "golden" / "corr_4096_digest.json").read_text())
    size = want["size"]
    a, b, _ = synth.make_pair(size, size)
    c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=0, steps=synth.optimal_scale_steps(size, size))
    cnt = {}
    fwd, rev = run_gpu(gpu_device, c, both=True, counters=cnt)
    assert cnt["candidates"] == want["candidates"]
    for name, (xy, corr) in (("forward", fwd), ("reverse", rev)):
        valid = xy[..., 0] >= 0
        assert int(valid.sum()) == want[name]["matches"], name
        assert hashlib.sha256(np.ascontiguousarray(xy, dtype=np.int32).tobytes()).hexdigest() == want[name]["xy_sha256"], name
        score_bits = np.where(valid, corr.view(np.uint32), np.uint32(0))
        assert hashlib.sha256(np.ascontiguousarray(score_bits).tobytes()).hexdigest() == want[name]["score_sha256"], name
    # the default mode - no counters, i.e. the <COUNT = false> instantiations the bench times; scores of the observable pass
    # only - gives the same forward grid: positions everywhere, score bits wherever there is a match
    got = run_gpu(gpu_device, c)
    assert (got[0] == fwd[0]).all()
    valid = fwd[0][..., 0] >= 0
    assert (got[1].view(np.uint32)[valid] == fwd[1].view(np.uint32)[valid]).all()


@pytest.mark.parametrize("fixture", ["corr_tilt10_4096_digest.json", "corr_tilt45_4096_digest.json", "corr_tilt60_4096_digest.json",
                                     "corr_perspective_2048_digest.json"])
def test_full_size_geometries_match_oracle_digest(gpu_device, fixture):
    """The stepped (10, 45 degrees), the transposed stepped (60 degrees) and the perspective instantiations of the box
    filter at FULL size against the oracle, by digest like config 3 above: forward and reverse match planes, score planes,
    match counts and the candidate count - generated by tests/tools/gen_digest_4096.py <size> <case> from
    oracle/cvref_corr.c (4096^2: a minute on 8 cores).  The regenerated input pair is checked first."""
    import hashlib
    import json
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).parent / "tools"))
    import gen_digest_4096

    want = json.loads((Path(__file__).parent / "golden" / fixture).read_text())
    size = want["size"]
    a, b, F, projection, _ = gen_digest_4096.case_inputs(want["case"], size)
    assert gen_digest_4096.inputs_digest(a, b) == want["inputs_sha256"], "the synthetic pair is not the one the digest was made from"
    c = dict(img1=a, img2=b, F=F, projection=projection, steps=synth.optimal_scale_steps(size, size))
    cnt = {}
    fwd, rev = run_gpu(gpu_device, c, both=True, counters=cnt)
    assert cnt["candidates"] == want["candidates"]
    for name, (xy, corr) in (("forward", fwd), ("reverse", rev)):
        valid = xy[..., 0] >= 0
        assert int(valid.sum()) == want[name]["matches"], name
        assert hashlib.sha256(np.ascontiguousarray(xy, dtype=np.int32).tobytes()).hexdigest() == want[name]["xy_sha256"], name
        score_bits = np.where(valid, corr.view(np.uint32), np.uint32(0))
        assert hashlib.sha256(np.ascontiguousarray(score_bits).tobytes()).hexdigest() == want[name]["score_sha256"], name
    # the default mode (no counters: the launches the bench times; scores of the observable pass only) gives the same forward grid
    got = run_gpu(gpu_device, c)
    assert (got[0] == fwd[0]).all()
    valid = fwd[0][..., 0] >= 0
    assert (got[1].view(np.uint32)[valid] == fwd[1].view(np.uint32)[valid]).all()


@pytest.mark.parametrize("name", ["tilt3_200x150", "h256", "flat"])
def test_triangulate_affine_matches_oracle(gpu_device, oracle, name):
    """Dense consumer (triangulation.rs:268-330) straight from the device grid: same tracks, same order,
    same f64 bits as the oracle applied to the completed grid."""
    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"])
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        pts, m2 = pc.triangulate_affine()
        xy, _ = pc.complete()
    finally:
        pc.close()
    want_pts, want_m2 = oracle.triangulate_affine(xy)
    assert pts.shape == want_pts.shape and (pts.view(np.uint64) == want_pts.view(np.uint64)).all()
    assert (m2 == want_m2).all()
    if name != "flat":
        assert len(pts) > 1000 and (pts[:, 2] >= 0).all()


def _tracks_for(xy, seed, n=4000):
    """Existing tracks for extend_tracks: points near dense matches, far from any, duplicates, image corners, and
    tracks without a point in image 1."""
    h, w = xy.shape[:2]
    rng = np.random.default_rng(seed)
    p = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
    p[::17] = -1                                   # track.get(image1_index) is None
    p[5:40] = p[4]                                 # several tracks at one point: all take the same match
    p[40:44] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]]
    return p


@pytest.mark.parametrize("name,max_dim2", [("persp_240x180", 240), ("tilt3_200x150", 2300), ("h256", 4096), ("flat", 128)])
def test_extend_tracks_matches_oracle(gpu_device, oracle, name, max_dim2):
    """Triangulation::extend_tracks (triangulation.rs:1330-1419) straight from the device grid: the same image-2 point
    for every existing track (nearest match, first minimum), the same new tracks in the same order as the oracle
    applied to the completed grid - for the reference's three radius regimes (3, 6, 12 px)."""
    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    pc = correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"], correlation.ProjectionMode(c["projection"]))
    try:
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        xy, _ = pc.complete()
        tracks = _tracks_for(xy, seed=len(name))
        tp2, n1, n2 = pc.extend_tracks(tracks, max_dim2)
        e_tp2, e_n1, e_n2 = pc.extend_tracks(np.zeros((0, 2), dtype=np.int32), max_dim2)  # no tracks yet: the first pair
    finally:
        pc.close()
    w_tp2, w_n1, w_n2 = oracle.extend_tracks(xy, tracks, max_dim2)
    assert (tp2 == w_tp2).all()
    assert n1.shape == w_n1.shape and (n1 == w_n1).all() and (n2 == w_n2).all()
    valid = xy[..., 0] >= 0
    assert len(e_tp2) == 0 and len(e_n1) == valid.sum() and (e_n2 == xy[valid]).all()
    if name != "flat":
        assert (tp2[:, 0] >= 0).sum() > 1000 and len(n1) < valid.sum()
        assert (tp2[tracks[:, 0] < 0] == -1).all()


def test_device_box_pyramid_equals_host(gpu_device):
    """cvhip_downsample_box reproduces synth.box_pyramid byte for byte (odd sizes included)."""
    import torch

    rng = np.random.default_rng(3)
    for (h, w) in [(257, 301), (128, 128), (65, 96)]:
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        want = synth.box_pyramid(img, 3)
        torch_stream_dev = correlation.create_gpu_context(stream=torch.cuda.current_stream().cuda_stream)
        try:
            got = correlation.box_pyramid_device(torch_stream_dev, torch.from_numpy(img).cuda(), 3)
            torch.cuda.synchronize()
        finally:
            torch_stream_dev.close()
        for a, b in zip(got, want):
            assert a.shape == b.shape and (a.cpu().numpy() == b).all()
        # host pointers work too
        import ctypes as C

        from cybervision_amd import _lib
        dst = np.zeros((h // 2, w // 2), dtype=np.uint8)
        _lib.check(_lib.lib().cvhip_downsample_box(gpu_device.handle, C.c_void_p(img.ctypes.data), w, h,
                                                   C.c_void_p(dst.ctypes.data)), "cvhip_downsample_box")
        assert (dst == want[1]).all()


@pytest.mark.parametrize("name,den", [("h256", 2), ("sem320x200", 3), ("tilt3_200x150", 2)])
def test_independent_band_mode_equals_unsharded(gpu_device, oracle, name, den):
    """cvhip_ctx_set_row_band: every shard context runs the whole pyramid on its band + halo with NO
    exchange between levels; stitching the bands of the final forward grids must give the oracle's grid
    bit for bit (this is what the single final gather does on a multi-GPU node)."""
    import torch

    from cybervision_amd import sharding

    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    F = correlation.CorrelationDirection.Forward
    want = run_oracle(oracle, c)
    ctxs = [correlation.PointCorrelations(gpu_device, (w1, h1), (w2, h2), c["F"]) for _ in range(den)]
    try:
        for r, pc in enumerate(ctxs):
            assert pc.set_row_band(r, den), "geometry should be row-local"
            for i in range(c["steps"] + 1):
                k = c["steps"] - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        gpu_device.synchronize()
        grids = [pc.level_grid(F) for pc in ctxs]
        for r in range(1, den):  # the final gather: band r of context r -> context 0
            g = grids[r]
            r0, r1 = sharding.shard_rows(g["lh"], r, den)
            sharding.copy_rows(grids[0], g, r0, r1)
        torch.cuda.synchronize()
        assert_same_grid(ctxs[0].complete(F), want, f"{name} stitched from {den} independent bands")
    finally:
        for pc in ctxs:
            pc.close()


def test_band_mode_rejects_non_row_local_geometry(gpu_device):
    for Fm, proj in [(cases.perspective_f(240, 180), correlation.ProjectionMode.Perspective),
                     (synth.f_tilt(60.0), correlation.ProjectionMode.Affine)]:
        pc = correlation.PointCorrelations(gpu_device, (240, 180), (240, 180), Fm, proj)
        try:
            assert pc.set_row_band(0, 2) is False
        finally:
            pc.close()


def test_independent_band_mode_eight_bands_1024(gpu_device):
    """The N = 8 plan on a 1024^2 pair (5 levels, bands of 128 rows at full resolution, the coarse levels
    covered almost entirely by halos): eight independently computed bands stitched together must equal
    the unsharded run bit for bit."""
    import torch

    from cybervision_amd import sharding

    a, b, _ = synth.make_pair(1024, 1024, sem_style=True)
    steps = synth.optimal_scale_steps(1024, 1024)
    c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=0, steps=steps)
    want = run_gpu(gpu_device, c)
    p1, p2 = cases.pyramids(c)
    F = correlation.CorrelationDirection.Forward
    den = 8
    main = correlation.PointCorrelations(gpu_device, (1024, 1024), (1024, 1024), c["F"])
    try:
        for r in range(den):
            pc = main if r == 0 else correlation.PointCorrelations(gpu_device, (1024, 1024), (1024, 1024), c["F"])
            try:
                assert pc.set_row_band(r, den)
                for i in range(steps + 1):
                    k = steps - i
                    pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
                gpu_device.synchronize()
                if r:
                    g, g0 = pc.level_grid(F), main.level_grid(F)
                    r0, r1 = sharding.shard_rows(g["lh"], r, den)
                    sharding.copy_rows(g0, g, r0, r1)
                    torch.cuda.synchronize()
            finally:
                if r:
                    pc.close()
        assert_same_grid(main.complete(F), want, "1024^2 stitched from 8 independent bands")
    finally:
        main.close()


def _digest_of(xy, corr):
    import hashlib

    valid = xy[..., 0] >= 0
    score_bits = np.where(valid, corr.view(np.uint32), np.uint32(0))
    return (int(valid.sum()), hashlib.sha256(np.ascontiguousarray(xy, dtype=np.int32).tobytes()).hexdigest(),
            hashlib.sha256(np.ascontiguousarray(score_bits).tobytes()).hexdigest())


@pytest.mark.parametrize("den,ahead", [(2, False), (4, False), (8, False), (8, True)])
def test_config4_band_plan_4096_matches_oracle_digest(gpu_device, den, ahead):
    """BASELINE config 4 as far as one GPU allows: the N-rank independent-band plan (cvhip_ctx_set_row_band(r, N)) on the
    4096^2 headline pair.  The N band contexts run one after the other on this GPU - each the whole 7-level pyramid on
    its band + halo, no exchange between levels - their forward bands are stitched into context 0 (what the single
    RCCL gather does on an N-GPU node) and complete()'s grid must hash to the ORACLE's full-size digest."""
    import json
    import torch

    from cybervision_amd import sharding

    want = json.loads((Path(__file__).parent / "golden" / "corr_4096_digest.json").read_text())
    size = want["size"]
    a, b, _ = synth.make_pair(size, size)
    steps = synth.optimal_scale_steps(size, size)
    c = dict(img1=a, img2=b, F=synth.F_HORIZONTAL, projection=0, steps=steps)
    p1, p2 = cases.pyramids(c)
    if ahead:  # as bench.py runs it: level images resident and borrowed, their statistics on the side stream
        def resident(p):
            buf = torch.zeros(p.size + 64, dtype=torch.uint8, device="cuda")
            buf[:p.size].copy_(torch.from_numpy(p).reshape(-1))
            return buf[:p.size].view(p.shape[0], p.shape[1])

        p1, p2 = [resident(p) for p in p1], [resident(p) for p in p2]
        torch.cuda.synchronize()
    Fd = correlation.CorrelationDirection.Forward
    main = correlation.PointCorrelations(gpu_device, (size, size), (size, size), c["F"])
    try:
        for r in range(den):
            pc = main if r == 0 else correlation.PointCorrelations(gpu_device, (size, size), (size, size), c["F"])
            try:
                pc.set_borrow_inputs(ahead)
                pc.set_stats_ahead(ahead)
                assert pc.set_row_band(r, den), "the headline geometry is row-local"
                for i in range(steps + 1):
                    k = steps - i
                    pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
                gpu_device.synchronize()
                if r:
                    g, g0 = pc.level_grid(Fd), main.level_grid(Fd)
                    assert (g["lw"], g["lh"]) == (size, size)
                    r0, r1 = sharding.shard_rows(g["lh"], r, den)
                    assert r1 - r0 == size // den
                    sharding.copy_rows(g0, g, r0, r1)
                    torch.cuda.synchronize()
            finally:
                if r:
                    pc.close()
        xy, corr = main.complete(Fd)
    finally:
        main.close()
    n, xy_sha, score_sha = _digest_of(xy, corr)
    assert n == want["forward"]["matches"]
    assert xy_sha == want["forward"]["xy_sha256"], f"4096^2 stitched from {den} bands: match plane differs from the oracle"
    assert score_sha == want["forward"]["score_sha256"], f"4096^2 stitched from {den} bands: scores differ from the oracle"


def test_library_rccl_path_world1(oracle):
    """The collectives inside the library (cvhip_rccl_*), at the world size one GPU allows: id -> communicator on a
    device handle that owns a PRIVATE stream -> cvhip_ctx_set_row_shard_rccl drives a whole pyramid (the all-gather hook
    fires after every search pass of a sharded level) -> in-place ncclAllGather of the level grid -> the band gather,
    which exchanges the chunk with itself through the same grouped ncclSend / ncclRecv the N-rank gather uses.  The grid
    must still be the oracle's afterwards.  Lifetime: the device handle is destroyed BEFORE the communicator; the
    library defers the handle's release to cvhip_rccl_destroy."""
    from cybervision_amd import sharding

    c = cases.make_case("sem320x200")
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    want = run_oracle(oracle, c)
    Fd = correlation.CorrelationDirection.Forward
    dev = correlation.create_gpu_context()  # private stream
    comm = sharding.RcclCommunicator(dev, sharding.RcclCommunicator.unique_id(), 0, 1)
    pc = correlation.PointCorrelations(dev, (w1, h1), (w2, h2), c["F"])
    try:
        pc.set_row_shard_rccl(comm)
        for i in range(c["steps"] + 1):
            k = c["steps"] - i
            pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        g = pc.level_grid(Fd)
        assert g["rows_per_shard"] == g["lh"] == h1
        comm.allgather(g["cells"], g["rows_per_shard"] * g["lw"] * 4)   # ncclAllGather, in place, on the handle's stream
        comm.gather(g["scores"], g["rows_per_shard"] * g["lw"] * 4, 0)  # grouped ncclSend + ncclRecv (self exchange)
        pc.gather_bands_rccl(comm, 0)                                   # the same through the context
        pc.gather_bands_rccl(comm, -1)                                  # "to every rank" = all-gather
        dev.synchronize()
        assert_same_grid(pc.complete(Fd), want, "library RCCL path, world 1")
    finally:
        pc.close()
        dev.close()    # deferred: the communicator still references the handle
        comm.close()   # releases both


def test_library_rccl_path_world2():
    """The library's RCCL path at world size 2, one process per GPU (tests/_rccl_gpu_worker.py): runs wherever the box has two
    or more GPUs - the first multi-GPU node this suite meets - and is skipped on the one-GPU boxes of this pool.  The in-place
    ncclAllGather after every sharded pass must leave the golden grid on every rank; the grouped ncclSend / ncclRecv gather of
    independent-band mode must leave it on rank 0."""
    import os
    import socket
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(root / "tests" / "_rccl_gpu_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, out


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tilt", [(8192, 8192, 0.0), (9001, 6003, 7.0)])
def test_sizes_beyond_the_benchmark_equal_exact_kernel(gpu_device, w, h, tilt):
    """Twice the benchmark's side (8 levels; the search range's integer sums at their stated limit of 8192) and a ragged
    9001 x 6003 pair with tilted lines (no dimension a multiple of any tile): the default search against the plain exact
    kernel, match grids and score bits of both directions, everything resident on the device."""
    import torch

    a, b, _ = synth.make_pair_torch(w, h, tilt_deg=tilt, device="cuda")
    steps = synth.optimal_scale_steps(w, h)
    d1, d2 = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
    F = synth.f_tilt(tilt) if tilt else synth.F_HORIZONTAL
    torch.cuda.synchronize()  # (the device handle submits to a stream of its own)

    def run(version):
        pc = correlation.PointCorrelations(gpu_device, (w, h), (w, h), F, correlation.ProjectionMode.Affine)
        try:
            pc.set_exact_scores(True)
            if version is not None:
                pc.set_search_version(version)
            for i in range(steps + 1):
                k = steps - i
                pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            out = []
            for d in (correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse):
                xy = torch.empty((h, w, 2), dtype=torch.int32, device="cuda")
                corr = torch.empty((h, w), dtype=torch.float32, device="cuda")
                pc.complete(d, out_xy=xy, out_corr=corr)
                out.append((xy, corr))
            gpu_device.synchronize()
            return out
        finally:
            pc.close()

    got, want = run(None), run(1)
    for (xy, corr), (xy1, corr1) in zip(got, want):
        valid = xy1[..., 0] >= 0
        assert 0.8 < float(valid.float().mean()) < 0.95
        assert torch.equal(xy, xy1)
        assert torch.equal(corr.view(torch.int32)[valid], corr1.view(torch.int32)[valid])
