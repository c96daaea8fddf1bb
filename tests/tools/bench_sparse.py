#!/usr/bin/env python3
"""Secondary benchmarks: ORB extraction, keypoint matching and RANSAC hypothesis
scoring through the C ABI, with the CPU oracle timed beside them.  Prints one JSON line each."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, fundamentalmatrix, orb, pointmatching, synth  # noqa: E402
from oracle import cvref  # noqa: E402  (baseline only)


def timeit(fn, reps=5):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t) / reps, out


dev = correlation.create_gpu_context()
quick = "--quick" in sys.argv

# ---- ORB: textured image + blocks (seed 77), 2048^2 and 4096^2 --------------------------------------
for side in ((1024,) if quick else (2048, 4096)):
    a, b, _ = synth.make_pair(side, side, seed=9)
    img1, img2 = synth.add_blocks(a, count=4000, seed=77), synth.add_blocks(b, count=4000, seed=77)
    dimg = torch.from_numpy(img1).cuda()
    tg, (xy, desc) = timeit(lambda: orb.extract_points(dev, dimg))
    tc = time.perf_counter()
    oxy, odesc = cvref.orb_extract(img1)
    tc = time.perf_counter() - tc
    assert (xy == oxy).all() and (desc == odesc).all()
    print(json.dumps({"bench": "orb_extract", "size": side, "keypoints": int(len(xy)), "gpu_ms": round(tg * 1e3, 3),
                      "mpx_per_s": round(side * side / 1e6 / tg, 1), "cpu_oracle_ms_1thread": round(tc * 1e3, 1),
                      "parity": "bit-exact"}))
    if side == (1024 if quick else 2048):
        k1 = (xy, desc)
        k2 = orb.extract_points(dev, img2)

# ---- matcher: the two keypoint sets above -------------------------------------------------------------
tg, (m, d) = timeit(lambda: pointmatching.match_points(dev, k1[0], k1[1], k2[0], k2[1], 48))
tc = time.perf_counter()
om, od = cvref.match_points(k1[0], k1[1], k2[0], k2[1], 48)
tc = time.perf_counter() - tc
assert (m == om).all() and (d == od).all()
print(json.dumps({"bench": "match_points", "n1": int(len(k1[0])), "n2": int(len(k2[0])), "matches": int(len(m)),
                  "gpu_ms": round(tg * 1e3, 3), "pairs_per_s": round(len(k1[0]) * len(k2[0]) / tg / 1e9, 2),
                  "unit": "G descriptor pairs/s", "cpu_oracle_ms_1thread": round(tc * 1e3, 1), "parity": "bit-exact"}))

# ---- RANSAC scoring: N = 20 000 matches (70 % inliers), H = 50 000 hypotheses ------------------------
rng = np.random.default_rng(42)
N, H = (5000, 5000) if quick else (20000, 50000)
x1 = rng.integers(0, 2048, size=N)
y1 = rng.integers(0, 2048, size=N)
x2 = np.clip(x1 + rng.integers(-60, 60, size=N), 0, None)
y2 = y1.copy()
outl = rng.random(N) < 0.3
y2[outl] = rng.integers(0, 2048, size=int(outl.sum()))
matches = np.stack([x1, y1, x2, y2], axis=1).astype(np.uint32)
F = np.repeat(synth.F_HORIZONTAL[None], H, axis=0).copy()
F += rng.normal(size=F.shape) * (10.0 ** rng.uniform(-6, -1, size=(H, 1, 1)))
dF = torch.from_numpy(F.reshape(H, 9)).cuda()
dm = torch.from_numpy(matches.astype(np.int64)).to(torch.int32).cuda()
dcnt = torch.empty(H, dtype=torch.int32, device="cuda")
derr = torch.empty(H, dtype=torch.float64, device="cuda")
import ctypes as C  # noqa: E402

from cybervision_amd import _lib  # noqa: E402


def score_dev():
    _lib.check(_lib.lib().cvhip_ransac_score(dev.handle, C.c_void_p(dF.data_ptr()), H, C.c_void_p(dm.data_ptr()), N,
                                             0.1, C.c_void_p(dcnt.data_ptr()), C.c_void_p(derr.data_ptr())), "score")
    dev.synchronize()


tg, _ = timeit(score_dev)
sub = min(H, 500)
tc = time.perf_counter()
oc, oe = cvref.ransac_score(F[:sub], matches, 0.1)
tc = (time.perf_counter() - tc) * H / sub
gc, ge = fundamentalmatrix.ransac_score(dev, F[:sub], matches, 0.1)
assert (gc == oc).all() and (ge.view(np.uint64) == oe.view(np.uint64)).all()
print(json.dumps({"bench": "ransac_score", "matches": N, "hypotheses": H, "gpu_ms": round(tg * 1e3, 3),
                  "evals_per_s": round(N * H / tg / 1e9, 2), "unit": "G (hypothesis, match) errors/s",
                  "cpu_oracle_ms_1thread_extrapolated": round(tc * 1e3, 1), "parity": "bit-exact (counts and sums)"}))
dev.close()
