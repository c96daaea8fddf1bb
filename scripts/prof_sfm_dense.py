#!/usr/bin/env python3
"""Per-level kernel times and search counters of ONE perspective dense correlation of the config-5 scene
(synth.make_sfm_views, pair (0, 1), the planted F).  usage: prof_sfm_dense.py [size] [--count]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

from cybervision_amd import correlation, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2048
views, K, poses = synth.make_sfm_views(W)
F = synth.sfm_true_f(K, poses[0], poses[1])
steps = synth.optimal_scale_steps(W, W)
d1 = [torch.from_numpy(p).cuda() for p in synth.box_pyramid(views[0], steps)]
d2 = [torch.from_numpy(p).cuda() for p in synth.box_pyramid(views[1], steps)]
dev = correlation.create_gpu_context()
pc = correlation.PointCorrelations(dev, (W, W), (W, W), F, correlation.ProjectionMode.Perspective)
for a in sys.argv:
    if a.startswith("--version="):
        pc.set_search_version(int(a.split("=")[1]))
pc.set_profiling(True, "--count" in sys.argv)
tot = 0.0
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / (1 << k))
    c = pc.get_counters()
    kt = pc.get_kernel_times()
    n = d1[k].numel() * 2
    tot += sum(v["ms"] for v in kt.values())
    times = " ".join(f"{name}={v['ms']:.3f}" for name, v in kt.items() if v["launches"])
    print(f"level k={k} {d1[k].shape[1]}x{d1[k].shape[0]}: {times} ms, cand/px {c['candidates'] / n:.1f}, "
          f"exact/px {c['exact_evals'] / n:.2f}, multi {c['multi_contender_pixels'] / n:.4f}, whole {c['whole_corridor_pixels'] / n:.5f}")
print("sum of kernel times", round(tot, 3), "ms")
pc.close()
dev.close()
