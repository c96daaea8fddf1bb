#!/usr/bin/env python3
"""Randomised check of the perspective RANSAC's schedules: batches of rounds scored as their generators
finish (the host polls), in order behind their events (cvhip_ransac_set_in_order), with and without the reference's
listener - the same matrix and the same inlier mask for every match set, seed and run, for both 7-point pencils (even cases:
the reference's thin-SVD rows, odd cases: the null space).   usage: fuzz_ransac.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402,F401

import cases  # noqa: E402
from cybervision_amd import correlation, fundamentalmatrix  # noqa: E402


class Listener:
    def __init__(self):
        self.matches = []

    def report_status(self, p):
        pass

    def report_matches(self, c):
        self.matches.append(c)


def run(n_cases=20, seed=1, dev=None, log=print):
    """n_cases random match sets through the polled, the in-order and the listener-attached schedule; returns the number of
    runs that disagree."""
    rng = np.random.default_rng(seed)
    own = dev is None
    if own:
        dev = correlation.create_gpu_context()
    bad = 0
    try:
        for it in range(n_cases):
            n = int(rng.integers(600, 30000))
            frac = float(rng.uniform(0.1, 0.6))
            size = int(rng.choice([1024, 2048, 4096]))
            m, truth, _, _ = cases.perspective_matches(n=n, outlier_frac=frac, seed=int(rng.integers(1, 10000)), size=size)
            rseed = int(rng.integers(0, 1 << 30))
            fmx = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Perspective, float(size))
            fundamentalmatrix.set_pencil(dev, it & 1)
            try:
                F0, _, mask0 = fmx.find_ransac(dev, m, seed=rseed, progress_listener=Listener())
            except Exception as ex:  # no model: both schedules must agree on that too
                F0, mask0 = None, str(ex)
            for rep in range(2):
                fundamentalmatrix.set_in_order(dev, rep == 1)
                try:
                    F1, _, mask1 = fmx.find_ransac(dev, m, seed=rseed)
                except Exception as ex:
                    F1, mask1 = None, str(ex)
                fundamentalmatrix.set_in_order(dev, False)
                same = (F0 is None and F1 is None and mask0 == mask1) or (
                    F0 is not None and F1 is not None and np.array_equal(F0, F1) and np.array_equal(mask0, mask1))
                if not same:
                    bad += 1
                    log(f"MISMATCH case {it} rep {rep}: n={n} outliers={frac:.2f} size={size} seed={rseed}")
            inl = int(np.asarray(mask0).sum()) if F0 is not None else -1
            log(f"case {it}: pencil={it & 1} n={n} outliers={frac:.2f} size={size} inliers={inl} of {int(truth.sum())} true")
    finally:
        fundamentalmatrix.set_pencil(dev, fundamentalmatrix.PENCIL_THIN_SVD)
        if own:
            dev.close()
    return bad


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    bad = run(N, int(sys.argv[2]) if len(sys.argv) > 2 else 1, log=lambda m: print(m, flush=True))
    print(f"done: {N} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
