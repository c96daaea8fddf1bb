#!/usr/bin/env python3
"""Beyond the benchmark's size: pairs of 8192 x 8192 and 9001 x 6003 (ragged: no dimension a multiple of any tile), rectified and
tilted, the default search against the plain exact kernel (search version 1) - match grids and score bits of both directions.
usage: big_size_check.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cybervision_amd import correlation, synth  # noqa: E402


def run(dev, d1, d2, F, steps, version, mode=correlation.ProjectionMode.Affine):
    H, W = d1[0].shape
    pc = correlation.PointCorrelations(dev, (W, H), (W, H), F, mode)
    pc.set_exact_scores(True)
    if version is not None:
        pc.set_search_version(version)
    t0 = time.perf_counter()
    for i in range(steps + 1):
        k = steps - i
        pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
    out = []
    for d in (correlation.CorrelationDirection.Forward, correlation.CorrelationDirection.Reverse):
        xy = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
        corr = torch.empty((H, W), dtype=torch.float32, device="cuda")
        pc.complete(d, out_xy=xy, out_corr=corr)
        out.append((xy, corr))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pc.close()
    return out, dt


dev = correlation.create_gpu_context()
bad = 0
# --sfm=SIZE: two of config 5's perspective views at that size (per-pixel lines, nine stripes) instead of the affine pairs
for a in [a for a in sys.argv[1:] if a.startswith("--sfm=")]:
    sys.argv.remove(a)
    size = int(a.split("=")[1])
    views, K, poses = synth.make_sfm_views(size)
    steps = synth.optimal_scale_steps(size, size)
    d1, d2 = ([torch.from_numpy(l).cuda() for l in synth.box_pyramid(v, steps)] for v in views[:2])
    F = synth.sfm_true_f(K, poses[0], poses[1])
    torch.cuda.synchronize()
    got, t_box = run(dev, d1, d2, F, steps, None, correlation.ProjectionMode.Perspective)
    want, t_v1 = run(dev, d1, d2, F, steps, 1, correlation.ProjectionMode.Perspective)
    ok = True
    for (xy, corr), (xy1, corr1) in zip(got, want):
        valid = xy1[..., 0] >= 0
        ok = ok and bool(torch.equal(xy, xy1)) and bool(torch.equal(corr.view(torch.int32)[valid], corr1.view(torch.int32)[valid]))
    print(f"perspective views {size} x {size}: {steps + 1} levels, matched {float((got[0][0][..., 0] >= 0).float().mean()):.3f}, default {t_box * 1e3:.1f} ms, exact kernel {t_v1 * 1e3:.0f} ms: {'EQUAL' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del got, want, d1, d2
    torch.cuda.empty_cache()
    if len(sys.argv) == 1:
        dev.close()
        sys.exit(1 if bad else 0)
SIZES = [tuple(float(v) if i == 2 else int(v) for i, v in enumerate(a.split("x"))) for a in sys.argv[1:]] or [(8192, 8192, 0.0), (9001, 6003, 0.0), (9001, 6003, 7.0), (6003, 9001, 80.0)]
for (W, H, tilt) in SIZES:
    a, b, _ = synth.make_pair_torch(W, H, tilt_deg=tilt, device="cuda")
    steps = synth.optimal_scale_steps(W, H)
    d1, d2 = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
    F = synth.f_tilt(tilt) if tilt else synth.F_HORIZONTAL
    torch.cuda.synchronize()  # (the device handle submits to a stream of its own)
    got, t_box = run(dev, d1, d2, F, steps, None)
    want, t_v1 = run(dev, d1, d2, F, steps, 1)
    ok = True
    for (xy, corr), (xy1, corr1) in zip(got, want):
        valid = xy1[..., 0] >= 0
        ok = ok and bool(torch.equal(xy, xy1)) and bool(torch.equal(corr.view(torch.int32)[valid], corr1.view(torch.int32)[valid]))
    matched = float((got[0][0][..., 0] >= 0).float().mean())
    print(f"{W} x {H}, tilt {tilt}: {steps + 1} levels, matched {matched:.3f}, default {t_box * 1e3:.1f} ms, exact kernel {t_v1 * 1e3:.0f} ms: {'EQUAL' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del got, want, d1, d2, a, b
    torch.cuda.empty_cache()
dev.close()
sys.exit(1 if bad else 0)
