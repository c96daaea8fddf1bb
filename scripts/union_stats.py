#!/usr/bin/env python3
"""How much longer than a pixel's own displacement range is the union over a row segment of the box kernel's width?
Runs the benchmark pair down to level 1, reads the forward grid, re-derives (approximately: float statistics, no clamps)
the level-0 search ranges of estimate_search_range and prints union / own for several segment widths.
usage: union_stats.py [tilt]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402

from cybervision_amd import correlation, synth  # noqa: E402

W = 4096
TILT = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
a, b, _ = synth.make_pair_torch(W, W, tilt_deg=TILT, device="cuda")
steps = synth.optimal_scale_steps(W, W)
d1, d2 = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
torch.cuda.synchronize()
dev = correlation.create_gpu_context()
pc = correlation.PointCorrelations(dev, (W, W), (W, W), synth.f_tilt(TILT) if TILT else synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
for i in range(steps):  # down to k = 1
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / (1 << k))
g = pc.level_grid(correlation.CorrelationDirection.Forward)
torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")
n = g["lw"] * g["lh"]
host = np.empty(n, dtype=np.uint32)
assert hip.hipDeviceSynchronize() == 0
assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(g["cells"]), C.c_size_t(n * 4), 2) == 0
cells = torch.from_numpy(host.astype(np.int64)).cuda().reshape(g["lh"], g["lw"])
valid = (cells != 0xFFFFFFFF).double()
X = (cells & 0xFFFF).double() * valid   # corridor position of a rectified pair: x of the match
k10 = torch.ones(1, 1, 10, 10, dtype=torch.float64, device="cuda")
def box(t):
    return Fn.conv2d(Fn.pad(t[None, None], (5, 4, 5, 4)), k10)[0, 0]
cnt, s1, s2 = box(valid), box(X), box(X * X)
ok = cnt > 0
mean = s1 / cnt.clamp(min=1)
var = (s2 / cnt.clamp(min=1) - mean * mean).clamp(min=0)
center = torch.round(2 * mean)
ln = torch.round(2.5 + 2 * var.sqrt())
xs = torch.arange(g["lw"], device="cuda", dtype=torch.float64)[None, :] * 2
lo, hi = center - ln - xs, center + ln - xs            # displacement interval of the level-0 pixels of this block
own = (hi - lo)[ok].mean().item()
print(f"tilt {TILT}: level-1 grid {g['lw']}x{g['lh']}, matched {valid.mean().item():.3f}; own range {own:.2f} steps on average")
for wpx in (16, 24, 32, 42, 53, 64, 106):
    wc = max(wpx // 2, 1)
    ncol = g["lw"] // wc
    L = torch.where(ok, lo, torch.full_like(lo, 1e9))[:, : ncol * wc].reshape(g["lh"], ncol, wc).amin(-1)
    H = torch.where(ok, hi, torch.full_like(hi, -1e9))[:, : ncol * wc].reshape(g["lh"], ncol, wc).amax(-1)
    has = H > L
    u = (H - L)[has]
    print(f"  segment {wpx:3d} px: union {u.mean().item():6.2f} steps = {u.mean().item() / own:.2f} x own; 90th pct {u.quantile(0.9).item():.0f}, over 61 steps: {(u > 61).double().mean().item():.4f}")
pc.close()
dev.close()
