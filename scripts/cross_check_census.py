#!/usr/bin/env python3
"""How the cells of the last level fare in the cross-check: fraction of matches whose counterpart points exactly back
(the centre probe), within +-1, within the +-4 window, or nowhere (removed) - the work profile of cross_check_kernel."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, synth  # noqa: E402
from cybervision_amd.correlation import CorrelationDirection as D  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img1, img2, _ = synth.make_pair(size, size)
steps = synth.optimal_scale_steps(size, size)
p1, p2 = synth.box_pyramid(img1, steps), synth.box_pyramid(img2, steps)
dev = correlation.create_gpu_context()
pc = correlation.PointCorrelations(dev, (size, size), (size, size), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
for i in range(steps + 1):
    k = steps - i
    if k > 0:
        pc.correlate_images(p1[k], p2[k], 1.0 / (1 << k))
    else:  # last level: the two search passes only
        pc.correlate_images_step(p1[0], p2[0], 1.0, D.Forward)
        pc.correlate_images_step(p2[0], p1[0], 1.0, D.Reverse)
fw, _ = pc.complete(D.Forward)
rv, _ = pc.complete(D.Reverse)
ys, xs = np.nonzero(fw[..., 0] >= 0)
mx, my = fw[ys, xs, 0], fw[ys, xs, 1]
back = rv[my, mx]  # the counterpart's own match
has = back[:, 0] >= 0
dx, dy = np.abs(back[:, 0] - xs), np.abs(back[:, 1] - ys)
exact = has & (dx == 0) & (dy == 0)
near = has & (dx <= 4) & (dy <= 4)
print(f"{len(xs)} forward matches of {size * size} cells: centre probe points back within +-4: {near.mean():.3f} "
      f"(exactly: {exact.mean():.3f}); centre probe empty or elsewhere: {1 - near.mean():.3f}")
# of those, how many are still supported by SOME cell of the +-4 window (the full scan's yield)
idx = np.nonzero(~near)[0]
sup = 0
for i in idx[:20000]:
    x0, x1, y0, y1 = max(mx[i] - 4, 0), min(mx[i] + 5, size), max(my[i] - 4, 0), min(my[i] + 5, size)
    w = rv[y0:y1, x0:x1]
    ok = (w[..., 0] >= 0) & (np.abs(w[..., 0] - xs[i]) <= 4) & (np.abs(w[..., 1] - ys[i]) <= 4)
    sup += bool(ok.any())
n = min(len(idx), 20000)
print(f"of the rest ({len(idx)}), the window scan still finds support for {sup / max(n, 1):.3f}")
