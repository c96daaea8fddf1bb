#!/usr/bin/env python3
"""Generates tests/golden/corr_4096_digest.json: SHA-256 digests of the ORACLE's result for BASELINE config 3 (the
4096x4096 headline pair, F = horizontal epipolar lines, affine parameter set, 7 box-pyramid levels) - forward and
reverse match planes and score planes.  The oracle's result does not depend on its thread count, so the digest made
here (any machine) pins the GPU result at full size without carrying 300 MB of fixtures.

    python tests/tools/gen_digest_4096.py [size] [case]     (about a minute on 8 cores for 4096)

case: rectified (default) | tilt<degrees> (synth.make_pair(tilt_deg=...), F = synth.f_tilt: the stepped box instantiations
at full size) | perspective (two views of synth.make_sfm_views, the true F, perspective parameter set) ->
tests/golden/corr_<case>_<size>_digest.json.
"""
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from cybervision_amd import synth  # noqa: E402


def digests(fwd, rev):
    out = {}
    for name, (xy, corr) in (("forward", fwd), ("reverse", rev)):
        valid = xy[..., 0] >= 0
        score_bits = np.where(valid, corr.view(np.uint32), np.uint32(0))     # None cells carry no score
        out[name] = {"xy_sha256": hashlib.sha256(np.ascontiguousarray(xy, dtype=np.int32).tobytes()).hexdigest(),
                     "score_sha256": hashlib.sha256(np.ascontiguousarray(score_bits).tobytes()).hexdigest(),
                     "matches": int(valid.sum())}
    return out


def inputs_digest(a, b):
    """SHA-256 of the two input images: the test checks that it regenerated the same pair before it looks at results."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes() + np.ascontiguousarray(b).tobytes()).hexdigest()


def case_inputs(case: str, size: int):
    """-> (img1, img2, F, projection, description): the inputs of a digest case (tests/test_corr_gpu.py builds the same)."""
    if case == "rectified":
        a, b, _ = synth.make_pair(size, size)
        return a, b, synth.F_HORIZONTAL, 0, {"pair": "synth.make_pair(size, size), seed 1234", "F": "synth.F_HORIZONTAL"}
    if case.startswith("tilt"):
        deg = float(case[4:])
        a, b, _ = synth.make_pair(size, size, tilt_deg=deg)
        return a, b, synth.f_tilt(deg), 0, {"pair": f"synth.make_pair(size, size, tilt_deg={deg}), seed 1234", "F": f"synth.f_tilt({deg})"}
    if case == "perspective":
        views, K, poses = synth.make_sfm_views(size)
        return views[0], views[1], synth.sfm_true_f(K, poses[0], poses[1]), 1, {
            "pair": "views 0, 1 of synth.make_sfm_views(size)", "F": "synth.sfm_true_f(K, poses[0], poses[1])"}
    raise SystemExit(f"unknown case {case}")


def main():
    from oracle import cvref

    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    case = sys.argv[2] if len(sys.argv) > 2 else "rectified"   # rectified | tilt<degrees> | perspective
    a, b, F, projection, desc = case_inputs(case, size)
    steps = synth.optimal_scale_steps(size, size)
    p1, p2 = synth.box_pyramid(a, steps), synth.box_pyramid(b, steps)
    t0 = time.time()
    c = cvref.Corr((size, size), (size, size), F, projection, os.cpu_count())
    for i in range(steps + 1):
        k = steps - i
        c.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
    out = {"size": size, "case": case, **desc, "inputs_sha256": inputs_digest(a, b), "projection": projection,
           "pyramid": "synth.box_pyramid", "levels": steps + 1, "candidates": c.candidates,
           "generator": "tests/tools/gen_digest_4096.py (oracle/cvref_corr.c)", **digests(c.get(0), c.get(1))}
    c.close()
    out["oracle_seconds"] = round(time.time() - t0, 1)
    name = "corr_4096_digest.json" if (size == 4096 and case == "rectified") else f"corr_{case}_{size}_digest.json"
    (ROOT / "tests" / "golden" / name).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
