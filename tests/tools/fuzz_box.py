#!/usr/bin/env python3
"""Randomised parity sweep of the box filter (stepped / transposed / lean instantiations, forced and
host-selected) against the CPU oracle on small pairs: random sizes (incl. differing image dims), tilts around
both axes, consistent and inconsistent displacement directions, SEM-style noise.  Prints every mismatch.
usage: fuzz_box.py [--perspective] [cases] [seed] [maxdim];  tests/test_fuzz_gpu.py runs a seeded slice of it under the suite."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, synth  # noqa: E402
from oracle import cvref  # noqa: E402


def run(n_cases=40, seed=1, maxdim=420, perspective=False, dev=None, versions=(3, 4, 5, 6), log=print):
    """n_cases random pairs x `versions` against the oracle; returns the number of mismatching (case, version) runs."""
    rng = np.random.default_rng(seed)
    own = dev is None
    if own:
        dev = correlation.create_gpu_context()
    bad = 0
    for it in range(n_cases):
        w, h = int(rng.integers(90, maxdim)), int(rng.integers(90, maxdim))
        tilt = float(rng.choice([0.0, 0.3, -0.7, 1.5, 3.0, -4.0, 86.0, 89.2, 90.0, 91.0, -88.0, 12.0, 45.0]))
        consistent = bool(rng.integers(0, 2))
        pair_seed = int(rng.integers(1, 1000))
        a, b, _ = synth.make_pair(w, h, seed=pair_seed, sem_style=bool(rng.integers(0, 2)), tilt_deg=tilt if consistent else 0.0)
        if rng.integers(0, 3) == 0:  # second image larger (the reverse grid has its own dims)
            b = np.ascontiguousarray(np.pad(b, ((0, int(rng.integers(1, 20))), (0, int(rng.integers(1, 20)))), mode="edge"))
        proj = int(rng.integers(0, 2)) if abs(tilt) < 5 else 0
        F = synth.f_tilt(tilt) if tilt != 0.0 else synth.F_HORIZONTAL
        if perspective:
            # a true perspective F (per-pixel epipolar lines): mostly sideways (or, every third case, vertical) camera
            # translation with small random rotations and a little motion along the optical axis
            proj = 1
            size = max(w, h)
            K = np.array([[0.9 * size, 0.0, w / 2.0], [0.0, 0.9 * size, h / 2.0], [0.0, 0.0, 1.0]])
            ang = rng.uniform(-0.004, 0.004, size=3) * float(rng.choice([0.2, 1.0, 3.0]))
            R = synth._rot(*ang)
            t = np.array([1.0, rng.uniform(-0.06, 0.06), rng.uniform(-0.03, 0.03)])
            if it % 3 == 2:
                t = t[[1, 0, 2]]
            F = synth.sfm_true_f(K, (np.eye(3), np.zeros(3)), (R, t * 0.05))
        steps = synth.optimal_scale_steps(a.shape[1], a.shape[0])
        p1, p2 = synth.box_pyramid(a, steps), synth.box_pyramid(b, steps)
        want = cvref.correlate_dense(p1, p2, F, proj)
        # (5: the rectified affine launches on the matrix pipe, 6: on two image columns per lane; everything else of such a run is version 3's)
        for version in versions:
            # a FRESH context per run, as the reference makes one per pair (the upload ring / pool hand-over are part of the sweep)
            pc = correlation.PointCorrelations(dev, (a.shape[1], a.shape[0]), (b.shape[1], b.shape[0]), F,
                                               correlation.ProjectionMode(proj))
            pc.set_search_version(version)
            # every other case goes through the reference's four calls per level, executed as one level
            # (cvhip_ctx_set_fuse_level_calls; host images: the upload ring)
            four_calls = bool((it + version) & 1)
            if four_calls:
                pc.set_fuse_level_calls(True)
            # result bands (cvhip_ctx_set_result_bands; taken where the geometry and the height allow) and packed cells
            bands = int(rng.integers(0, 5))   # (0: the library's choice by size)
            pc.set_result_bands(bands)
            for i in range(steps + 1):
                k = steps - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=not four_calls)
            if (it + version) & 2:
                cells, corr = pc.complete_packed()
                got = (pc.unpack_cells(cells), corr)
            else:
                got = pc.complete()
            pc.close()
            ok = (got[0] == want[0]).all()
            v = want[0][..., 0] >= 0
            ok = ok and (got[1].view(np.uint32)[v] == want[1].view(np.uint32)[v]).all()
            if not ok:
                bad += 1
                log(f"MISMATCH it={it} version={version} four_calls={four_calls} bands={bands} {w}x{h} b={b.shape} tilt={tilt} consistent={consistent} seed={pair_seed} "
                    f"proj={proj} diff_cells={(got[0] != want[0]).any(axis=-1).sum()}")
        if it % 10 == 9:
            log(f"{it + 1} cases, {bad} mismatches")
    if own:
        dev.close()
    return bad


if __name__ == "__main__":
    PERSPECTIVE = "--perspective" in sys.argv
    argv = [a for a in sys.argv if a != "--perspective"]
    N = int(argv[1]) if len(argv) > 1 else 40
    bad = run(N, int(argv[2]) if len(argv) > 2 else 1, int(argv[3]) if len(argv) > 3 else 420, PERSPECTIVE,
              log=lambda m: print(m, flush=True))
    print(f"done: {N} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
