"""Result bands (cvhip_ctx_set_result_bands) on the 4096^2 pair: per-pair time with the forward grid landing in host memory, for
1 .. 16 bands - (a) resident pyramid, level call, page-locked destination; (b) the reference's four calls per level on
pageable host level images, pageable destination; (c) resident pyramid, grid left in HBM (what banding costs the search).
and (d) the same with the 8-byte cells of cvhip_complete_packed.  python scripts/result_bands_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from cybervision_amd import correlation, synth

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--bands", default="1,2,4,6,8,12,16")
ap.add_argument("--modes", default="pinned,four_call_host,device,pinned_packed,four_call_host_packed")
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--size", type=int, default=4096)
args = ap.parse_args()
W = H = args.size
steps = synth.optimal_scale_steps(W, H)
a, b, _ = synth.make_pair_torch(W, H, device="cuda")
pa, pb = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
def resident(p):
    buf = torch.zeros(p.numel() + 64, dtype=torch.uint8, device="cuda")
    buf[:p.numel()].copy_(p.reshape(-1))
    return buf[:p.numel()].view(p.shape[0], p.shape[1])
d1, d2 = [resident(p) for p in pa], [resident(p) for p in pb]
h1, h2 = [p.cpu().numpy() for p in d1], [p.cpu().numpy() for p in d2]
dxy = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
dco = torch.empty((H, W), dtype=torch.float32, device="cuda")
pxy, pco = torch.empty((H, W, 2), dtype=torch.int32).pin_memory(), torch.empty((H, W), dtype=torch.float32).pin_memory()
hxy, hco = np.empty((H, W, 2), dtype=np.int32), np.empty((H, W), dtype=np.float32)
pce, hce = torch.empty((H, W), dtype=torch.int32).pin_memory(), np.empty((H, W), dtype=np.uint32)
stream = torch.cuda.current_stream()
dev = correlation.create_gpu_context(ordinal=0, stream=stream.cuda_stream)
want = None
for bands in [int(v) for v in args.bands.split(",")]:
    row = {}
    for mode in args.modes.split(","):
        pc = correlation.PointCorrelations(dev, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
        pc.set_result_bands(bands)
        if not mode.startswith("four_call_host"):
            pc.set_borrow_inputs(True)
            pc.set_stats_ahead(True)
        else:
            pc.set_fuse_level_calls(True)
        src = (h1, h2) if mode.startswith("four_call_host") else (d1, d2)
        def pair():
            pc.first_pass = True
            for i in range(steps + 1):
                k = steps - i
                pc.correlate_images(src[0][k], src[1][k], 1.0 / float(1 << k), fused=not mode.startswith("four_call_host"))
            if mode == "pinned":
                pc.complete(out_xy=pxy, out_corr=pco)
            elif mode == "device":
                pc.complete(out_xy=dxy, out_corr=dco)
            elif mode == "pinned_packed":
                pc.complete_packed(out_cells=pce, out_corr=pco)
            elif mode == "four_call_host_packed":
                pc.complete_packed(out_cells=hce, out_corr=hco)
            else:
                pc.complete(out_xy=hxy, out_corr=hco)
        pair(); pair()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = args.pairs
        for _ in range(n):
            pair()
        torch.cuda.synchronize()
        row[mode] = (time.perf_counter() - t) * 1e3 / n
        live = pc.result_bands()
        got = {"pinned": lambda: pxy.numpy(), "device": lambda: dxy.cpu().numpy(), "four_call_host": lambda: hxy,
               "pinned_packed": lambda: correlation.PointCorrelations.unpack_cells(pce.numpy().view(np.uint32)),
               "four_call_host_packed": lambda: correlation.PointCorrelations.unpack_cells(hce)}[mode]()
        if want is None:
            want = got.copy()
        same = bool((got == want).all())
        pc.close()
        row[mode + "_same"] = same
    print(f"bands {bands:2d} (live {live}): " + ", ".join(f"{m} {row[m]:.3f} ms" for m in args.modes.split(",")) +
          f"; same {all(v for k_, v in row.items() if k_.endswith('_same'))}", flush=True)
