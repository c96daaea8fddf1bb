#!/usr/bin/env python3
"""Run the dense correlation of the benchmark pair once and print the search kernel's device
counters (candidates, exact evaluations, multi-contender pixels, whole-corridor pixels) per level."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
TILT = 0.0
for a in sys.argv:
    if a.startswith("--tilt="):
        TILT = float(a.split("=")[1])
img1, img2, _ = synth.make_pair_torch(W, W, tilt_deg=TILT, device="cuda")  # displacement along the epipolar direction of F
steps = synth.optimal_scale_steps(W, W)
d1, d2 = synth.box_pyramid_torch(img1, steps), synth.box_pyramid_torch(img2, steps)
torch.cuda.synchronize()  # the device handle below submits to a stream of its own
dev = correlation.create_gpu_context()
F = synth.F_HORIZONTAL
for a in sys.argv:
    if a.startswith("--tilt="):
        F = synth.f_tilt(float(a.split("=")[1]))
proj = correlation.ProjectionMode.Perspective if "--perspective" in sys.argv else correlation.ProjectionMode.Affine
pc = correlation.PointCorrelations(dev, (W, W), (W, W), F, proj)
COUNT = "--count" in sys.argv
for a in sys.argv:
    if a.startswith("--version="):
        pc.set_search_version(int(a.split("=")[1]))
pc.set_profiling(True, COUNT)
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / (1 << k))
    c = pc.get_counters()
    kt = pc.get_kernel_times()
    n = d1[k].numel() * 2
    times = " ".join(f"{name}={v['ms']:.3f}" for name, v in kt.items() if v["launches"])
    print(f"level k={k} {d1[k].shape[1]}x{d1[k].shape[0]}: {times} ms, cand/px {c['candidates'] / n:.1f}, "
          f"exact/px {c['exact_evals'] / n:.2f}, multi {c['multi_contender_pixels'] / n:.4f}, whole {c['whole_corridor_pixels'] / n:.5f}")
pc.close()
dev.close()
