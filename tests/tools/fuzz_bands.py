#!/usr/bin/env python3
"""Randomised check of the independent-band multi-GPU mode (cvhip_ctx_set_row_band) on ONE GPU: for random
sizes, band counts and (row-local) tilts, `den` contexts each compute their band + halo with no exchange, the
bands of the final forward grids are stitched as the single all-gather would, and the result must equal the
CPU oracle bit for bit.  Geometry the library declines (set_row_band -> False) is skipped and counted."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cybervision_amd import correlation, sharding, synth  # noqa: E402
from oracle import cvref  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = correlation.create_gpu_context()
Fdir = correlation.CorrelationDirection.Forward
bad = skipped = 0
for it in range(N):
    w, h = int(rng.integers(130, 700)), int(rng.integers(130, 700))
    den = int(rng.choice([2, 3, 4, 5, 8]))
    tilt = float(rng.choice([0.0, 0.0, 0.4, -1.0, 2.5]))
    a, b, _ = synth.make_pair(w, h, seed=int(rng.integers(1, 999)), sem_style=bool(rng.integers(0, 2)), tilt_deg=tilt)
    F = synth.f_tilt(tilt) if tilt else synth.F_HORIZONTAL
    steps = synth.optimal_scale_steps(w, h)
    p1, p2 = synth.box_pyramid(a, steps), synth.box_pyramid(b, steps)
    ctxs = [correlation.PointCorrelations(dev, (w, h), (w, h), F) for _ in range(den)]
    try:
        if not all(pc.set_row_band(r, den) for r, pc in enumerate(ctxs)):
            skipped += 1
            continue
        want = cvref.correlate_dense(p1, p2, F, 0)
        for pc in ctxs:
            for i in range(steps + 1):
                k = steps - i
                pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
        dev.synchronize()
        grids = [pc.level_grid(Fdir) for pc in ctxs]
        for r in range(1, den):
            g = grids[r]
            r0, r1 = sharding.shard_rows(g["lh"], r, den)
            sharding.copy_rows(grids[0], g, r0, r1)
        torch.cuda.synchronize()
        got = ctxs[0].complete(Fdir)
        v = want[0][..., 0] >= 0
        ok = (got[0] == want[0]).all() and (got[1].view(np.uint32)[v] == want[1].view(np.uint32)[v]).all()
        if not ok:
            bad += 1
            print(f"MISMATCH it={it} {w}x{h} den={den} tilt={tilt} diff_cells={(got[0] != want[0]).any(axis=-1).sum()}")
    finally:
        for pc in ctxs:
            pc.close()
    if it % 10 == 9:
        print(f"{it + 1} cases, {bad} mismatches, {skipped} skipped", flush=True)
print(f"done: {N} cases, {bad} mismatches, {skipped} skipped")
sys.exit(1 if bad else 0)
