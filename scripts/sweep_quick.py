#!/usr/bin/env python3
"""Level-0 and whole-step times of the 4096^2 workload for a list of tilts (no extras): quick A/B of the box kernels."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cybervision_amd import correlation, synth  # noqa: E402

W = 4096
tilts = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,3,10,30,60,90").split(",")]
VERSION = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--version=")), None)  # cvhip_ctx_set_search_version
proj = correlation.ProjectionMode.Affine
steps = synth.optimal_scale_steps(W, W)
stream = torch.cuda.current_stream()
dev = correlation.create_gpu_context(ordinal=0, stream=stream.cuda_stream)
out_xy = torch.empty((W, W, 2), dtype=torch.int32, device="cuda")
out_corr = torch.empty((W, W), dtype=torch.float32, device="cuda")
for tilt in tilts:
    a, b, _ = synth.make_pair_torch(W, W, tilt_deg=tilt, device="cuda")
    d1, d2 = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
    F = synth.f_tilt(tilt) if tilt != 0.0 else synth.F_HORIZONTAL
    pc = correlation.PointCorrelations(dev, (W, W), (W, W), F, proj)
    if VERSION is not None:
        pc.set_search_version(VERSION)

    def step(ev=None):
        pc.first_pass = True
        for j in range(steps + 1):
            k = steps - j
            if ev is not None and k == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            if ev is not None and k == 0:
                e1.record(stream)
                ev.append((e0, e1))
        pc.complete(out_xy=out_xy, out_corr=out_corr)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    ev = []
    step(ev)
    torch.cuda.synchronize()
    print(f"tilt {tilt:5.1f}: step {ms:7.3f} ms  level0 {ev[0][0].elapsed_time(ev[0][1]):7.3f} ms  matched {float((out_xy[..., 0] >= 0).float().mean()):.4f}", flush=True)
    pc.close()
dev.close()
