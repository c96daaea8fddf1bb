#!/usr/bin/env python3
"""Wall time of cvhip_orb_extract per pyramid level of a config-5 view (host and device-resident inputs)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cybervision_amd import correlation, orb, synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
views, K, poses = synth.make_sfm_views(size)
dev = correlation.create_gpu_context()
pyr = synth.box_pyramid(views[0], orb.optimal_scale_steps(size, size))
for resident in (False, True):
    levels = [torch.from_numpy(l).cuda() for l in pyr] if resident else pyr
    for lvl in levels:
        orb.extract_points(dev, lvl)
    for lvl in levels:
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            xy, desc = orb.extract_points(dev, lvl)
        dev.synchronize()
        print(f"resident={resident} {lvl.shape[1]}x{lvl.shape[0]}: {1e3 * (time.perf_counter() - t0) / 5:.3f} ms, {len(xy)} keypoints", flush=True)
