#!/usr/bin/env python3
"""The RANSAC scoring chain ALONE (no generators beside it): four rounds of caller-given hypotheses through the device loops'
rounds (cvhip_ransac_rounds_pick), with the counting screen's head on the vector pipe (default) and on the matrix pipe
(cvhip_ransac_set_count_mfma).  5 % of the hypotheses are the true F lightly perturbed, the others are far off - roughly the mix
config 5's generators produce.   usage: count_kernel_bench.py [hypotheses per round] [matches]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

import cases  # noqa: E402
from cybervision_amd import correlation, fundamentalmatrix  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 29000
m, truth, _, F_true = cases.perspective_matches(n=N, outlier_frac=0.35, seed=9)
rng = np.random.default_rng(3)
rounds = 4
F = np.empty((rounds * H, 9))
good = rng.random(rounds * H) < 0.05
F[:] = F_true.reshape(9) * (1.0 + rng.normal(0.0, 0.3, size=(rounds * H, 9)))
F[good] = F_true.reshape(9) * (1.0 + rng.normal(0.0, 2e-5, size=(int(good.sum()), 9)))
t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048
dev = correlation.create_gpu_context()
res = {}
for mode in (False, True, False, True):
    fundamentalmatrix.set_count_mfma(dev, mode)
    fundamentalmatrix.ransac_rounds_pick(dev, F, rounds, m, t, 8)  # warm
    dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = fundamentalmatrix.ransac_rounds_pick(dev, F, rounds, m, t, 8)
    dev.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    res.setdefault(mode, []).append(ms)
    print(f"count_mfma={int(mode)}: {ms:7.3f} ms per {rounds} rounds of {H} hypotheses x {N} matches; best count {out[2]} (index {out[0]})", flush=True)
dev.close()
