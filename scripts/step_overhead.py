"""What does the stepped instantiation of the box kernel cost where nothing steps?  The rectified 4096^2 pair with search version 3
(lean instantiation) and 4 (the stepped one forced), whole step and the full-resolution level: python scripts/step_overhead.py [tilt]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from cybervision_amd import correlation, synth

tilt = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
W = H = 4096
steps = synth.optimal_scale_steps(W, H)
a, b, _ = synth.make_pair_torch(W, H, tilt_deg=tilt, device="cuda")
pa, pb = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
def resident(p):
    buf = torch.zeros(p.numel() + 64, dtype=torch.uint8, device="cuda")
    buf[:p.numel()].copy_(p.reshape(-1))
    return buf[:p.numel()].view(p.shape[0], p.shape[1])
d1, d2 = [resident(p) for p in pa], [resident(p) for p in pb]
out_xy = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
out_corr = torch.empty((H, W), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
dev = correlation.create_gpu_context(ordinal=0, stream=stream.cuda_stream)
F = synth.F_HORIZONTAL if tilt == 0.0 else synth.f_tilt(tilt)
ref = None
for version in (3, 4):
    pc = correlation.PointCorrelations(dev, (W, H), (W, H), F, correlation.ProjectionMode.Affine)
    pc.set_borrow_inputs(True)
    pc.set_search_version(version)
    ev = []
    def step(timed=False):
        pc.first_pass = True
        for i in range(steps + 1):
            k = steps - i
            if timed and k == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            if timed and k == 0:
                e1.record(stream)
                ev.append((e0, e1))
        pc.complete(out_xy=out_xy, out_corr=out_corr)
    step(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    step(True)
    torch.cuda.synchronize()
    l0 = ev[0][0].elapsed_time(ev[0][1])
    same = True if ref is None else bool(torch.equal(ref, out_xy))
    if ref is None:
        ref = out_xy.clone()
    print(f"tilt {tilt:g}, search version {version}: {ms:.3f} ms per step, level 0 {l0:.3f} ms, same grid as version 3: {same}")
    pc.close()
dev.close()
