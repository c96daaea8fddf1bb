#!/usr/bin/env python3
"""How long does the HOST take to submit one dense-correlation step (all levels + complete), compared with the
step's device time?  usage: cpu_submit_time.py [size]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cybervision_amd import correlation, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a, b, _ = synth.make_pair(W, W)
steps = synth.optimal_scale_steps(W, W)


def resident(p):
    buf = torch.zeros(p.size + 64, dtype=torch.uint8, device="cuda")
    buf[:p.size].copy_(torch.from_numpy(p).reshape(-1))
    return buf[:p.size].view(p.shape[0], p.shape[1])


d1 = [resident(p) for p in synth.box_pyramid(a, steps)]
d2 = [resident(p) for p in synth.box_pyramid(b, steps)]
dev = correlation.create_gpu_context(stream=torch.cuda.current_stream().cuda_stream)
pc = correlation.PointCorrelations(dev, (W, W), (W, W), synth.F_HORIZONTAL)
pc.set_borrow_inputs(True)
out_xy = torch.empty((W, W, 2), dtype=torch.int32, device="cuda")
out_corr = torch.empty((W, W), dtype=torch.float32, device="cuda")


def step():
    pc.first_pass = True
    for i in range(steps + 1):
        k = steps - i
        pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
    pc.complete(out_xy=out_xy, out_corr=out_corr)


for _ in range(5):
    step()
torch.cuda.synchronize()
sub, tot = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sub.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print(f"size {W}: host submission {sorted(sub)[10]:.3f} ms, step until idle {sorted(tot)[10]:.3f} ms")
