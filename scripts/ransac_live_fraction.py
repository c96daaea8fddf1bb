#!/usr/bin/env python3
"""Fraction of the perspective generator's hypothesis slots (3 per sample) that survive validate_f's per-hypothesis
checks, on the config-5 scene's matches (pair 0-1) - what the scoring kernel's compaction removes."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, fundamentalmatrix, orb, pointmatching, synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
views, K, poses = synth.make_sfm_views(size)
dev = correlation.create_gpu_context()
kp = [orb.extract_points_multiscale(dev, synth.box_pyramid(v, orb.optimal_scale_steps(size, size))) for v in views[:2]]
m, _ = pointmatching.match_points(dev, kp[0][0], kp[0][1], kp[1][0], kp[1][1], 48)
rng = np.random.default_rng(1)
mm = m.astype(np.int64)
idx = []
while len(idx) < 20000:
    cand = rng.integers(0, min(len(m), 5000), size=(40000, 7))
    pts = mm[cand]
    d = np.abs(pts[:, :, None, :] - pts[:, None, :, :])
    close = (d < 10).any(axis=-1)
    close[:, np.arange(7), np.arange(7)] = False
    idx.extend(cand[~close.any(axis=(1, 2))].tolist())
idx = np.array(idx[:20000], dtype=np.uint32)
out = fundamentalmatrix.perspective_models_device(dev, m, idx, 0.01 * size)
live = np.isfinite(out[:, :, 0, 0])
print(f"{len(m)} matches, {len(idx)} samples: live slots {live.mean():.3f} of 3 per sample; samples with >= 1 live {live.any(axis=1).mean():.3f}")
