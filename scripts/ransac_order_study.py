#!/usr/bin/env python3
"""How far does a RANSAC round's hypothesis have to walk down the match list before it is abandoned, under different
orders of the list?  Config 5's first pair: models of 20 000 random 7-samples (device generator), their full inlier
matrix (torch, f64), the best model's count B as the bound; a hypothesis is abandoned at the first group of 128 where its
misses exceed N - B.  Prints the mean fraction of the list walked per order."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from cybervision_amd import correlation, fundamentalmatrix as fm, reconstruction, synth  # noqa: E402

size = 2048
views, K, poses = synth.make_sfm_views(size)
lsteps = synth.optimal_scale_steps(size, size)
dev = correlation.create_gpu_context(ordinal=0, stream=torch.cuda.current_stream().cuda_stream)
pyr = [[torch.from_numpy(l).cuda() for l in synth.box_pyramid(v, lsteps)] for v in views]
rec = reconstruction.ImageReconstruction(dev, fm.ProjectionMode.Perspective)
kp = rec.extract_keypoints_set(pyr)
PAIR = (0, 1) if len(sys.argv) < 2 else tuple(int(c) for c in sys.argv[1].split(","))
matches = np.asarray(rec.match_keypoints(kp[PAIR[0]], kp[PAIR[1]]), dtype=np.uint32).reshape(-1, 4)
N = len(matches)
t = fm.RANSAC_T_PERSPECTIVE * size
rng = np.random.default_rng(7)
idx = np.stack([rng.choice(N, 7, replace=False) for _ in range(20000)]).astype(np.uint32)
F = fm.perspective_models_device(dev, matches, idx, t).reshape(-1, 9)
F = F[np.isfinite(F).all(axis=1)]
print(f"N = {N}, live hypotheses {len(F)} of {3 * len(idx)} slots")
Ft = torch.from_numpy(F).cuda()
m = torch.from_numpy(matches.astype(np.float64)).cuda()
x1, y1, x2, y2 = m[:, 0], m[:, 1], m[:, 2], m[:, 3]
inl = torch.empty((len(F), N), dtype=torch.bool, device="cuda")
for a in range(0, len(F), 1024):
    f = Ft[a:a + 1024]
    c = [f[:, i:i + 1] for i in range(9)]
    r0 = x2 * c[0] + y2 * c[3] + c[6]
    r1 = x2 * c[1] + y2 * c[4] + c[7]
    r2 = x2 * c[2] + y2 * c[5] + c[8]
    n = r0 * x1 + r1 * y1 + r2
    a0 = c[0] * x1 + c[1] * y1 + c[2]
    a1 = c[3] * x1 + c[4] * y1 + c[5]
    den = a0 * a0 + a1 * a1 + r0 * r0 + r1 * r1
    err = n * n / den
    inl[a:a + 1024] = torch.isfinite(err) & ~(err > t)
    if a == 0:
        err0 = err
counts = inl.sum(dim=1)
best = int(torch.argmax(counts))
B = int(counts[best])
q = torch.quantile(counts.double(), torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], dtype=torch.float64, device="cuda"))
print(f"best count {B} ({B / N:.3f} of N); count quantiles 10/25/50/75/90/99 %: {[int(v) for v in q]}")
popularity = inl.double().mean(dim=0)
best_in = inl[best]
f = Ft[best]
berr = ((x2 * f[0] + y2 * f[3] + f[6]) * x1 + (x2 * f[1] + y2 * f[4] + f[7]) * y1 + (x2 * f[2] + y2 * f[5] + f[8])) ** 2
ar = torch.arange(N, device="cuda")


def walked(order, label):
    miss = (~inl[:, order]).to(torch.int32)
    groups = (N + 127) // 128
    pad = groups * 128 - N
    if pad:
        miss = torch.nn.functional.pad(miss, (0, pad))
    per_group = miss.view(len(F), groups, 128).sum(dim=2).cumsum(dim=1)
    dead = per_group > (N - B)
    first = torch.where(dead.any(dim=1), dead.to(torch.int32).argmax(dim=1) + 1, torch.full((len(F),), groups, device="cuda"))
    frac = first.double() / groups
    # two hypotheses per workgroup walk as far as the longer of the two
    pairs = frac[: len(F) // 2 * 2].view(-1, 2)
    print(f"{label:58s} mean walked {float(frac.mean()):.3f}   per pair (max of two) {float(pairs.max(dim=1).values.mean()):.3f}   "
          f"never abandoned {float((first == groups).double().mean()):.3f}")


walked(ar, "the matcher's order")
walked(torch.argsort(best_in.to(torch.int32), stable=True), "best's outliers first")
key = torch.where(best_in, 1.0 + popularity, popularity * 0.0)
walked(torch.argsort(key, stable=True), "best's outliers, then its inliers by popularity (asc)")
walked(torch.argsort(popularity, stable=True), "everything by popularity (asc)")
key = torch.where(best_in, 1.0 - berr / berr.max(), torch.full_like(berr, -1.0))
walked(torch.argsort(key, stable=True), "best's outliers, then its inliers by its own error (desc)")
# popularity estimated from a sample of 256 hypotheses only
pop256 = inl[:256].double().mean(dim=0)
walked(torch.argsort(pop256, stable=True), "everything by popularity among 256 hypotheses (asc)")
top = torch.argsort(counts, descending=True)[:64]
poptop = inl[top].double().mean(dim=0)
walked(torch.argsort(poptop, stable=True), "everything by popularity among the 64 best (asc)")
dev.close()
