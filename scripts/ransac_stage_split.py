#!/usr/bin/env python3
"""Where config 5's RANSAC stage spends its time for one pair: the 20 device rounds (cvhip_ransac_perspective), the host
LM refit on the winner's inliers (cvhip_optimize_perspective_f) and the inlier re-selection (cvhip_fits_model)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, fundamentalmatrix as fm, orb, pointmatching, synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
views, K, poses = synth.make_sfm_views(size)
dev = correlation.create_gpu_context()
kp = [orb.extract_points_multiscale(dev, synth.box_pyramid(v, orb.optimal_scale_steps(size, size))) for v in views[:2]]
m, _ = pointmatching.match_points(dev, kp[0][0], kp[0][1], kp[1][0], kp[1][1], 48)
for rep in range(3):
    dev.synchronize()
    t0 = time.perf_counter()
    F, mask = fm.find_ransac_perspective_device(dev, m, float(size), seed=7, refit=False)
    t1 = time.perf_counter()
    Fo = fm.optimize_perspective_f(F, m[mask])
    t2 = time.perf_counter()
    Fd = fm.optimize_perspective_f_device(dev, F, m[mask])
    t2b = time.perf_counter()
    assert (Fo is None) == (Fd is None) and (Fo is None or (Fo == Fd).all())
    mask2 = fm.inlier_mask(dev, Fo if Fo is not None else F, m, fm.RANSAC_T_PERSPECTIVE * size)
    t3 = time.perf_counter()
    whole = fm.FundamentalMatrix(fm.ProjectionMode.Perspective, float(size)).find_ransac(dev, m, seed=7)
    t4 = time.perf_counter()
    print(f"rounds {1e3 * (t1 - t0):.1f} ms, refit on {int(mask.sum())} inliers: host {1e3 * (t2 - t1):.1f} ms, "
          f"device {1e3 * (t2b - t2):.2f} ms; re-selection {1e3 * (t3 - t2b):.1f} ms; cvhip_find_ransac (all of it) {1e3 * (t4 - t3):.1f} ms", flush=True)
