#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/ — our restatement of the reference's
--mode=cpu path; the reference itself ships no fixtures and cannot be built here).

Each file holds the inputs (both images, F, projection) and the expected forward/reverse grids,
so the GPU tests can check against committed data without running the oracle.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import cases  # noqa: E402
from oracle import cvref  # noqa: E402

out = ROOT / "tests" / "golden"
out.mkdir(parents=True, exist_ok=True)
for name in cases.GOLDEN_CASES:
    c = cases.make_case(name)
    p1, p2 = cases.pyramids(c)
    h1, w1 = c["img1"].shape
    h2, w2 = c["img2"].shape
    oc = cvref.Corr((w1, h1), (w2, h2), c["F"], c["projection"], 8)
    for i in range(c["steps"] + 1):
        k = c["steps"] - i
        oc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
    fxy, fcorr = oc.get(0)
    rxy, rcorr = oc.get(1)
    np.savez_compressed(out / f"corr_{name}.npz", img1=c["img1"], img2=c["img2"], F=c["F"],
                        projection=c["projection"], steps=c["steps"], fwd_xy=fxy.astype(np.int16),
                        fwd_corr=fcorr, rev_xy=rxy.astype(np.int16), rev_corr=rcorr,
                        candidates=oc.candidates)
    print(name, "valid fwd", int((fxy[..., 0] >= 0).sum()), "cand", oc.candidates)
    oc.close()
