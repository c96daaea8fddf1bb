"""Does replaying one pyramid step as a captured HIP graph beat the stream launches?  (probe, 4096^2 headline workload)
python scripts/graph_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from cybervision_amd import correlation, synth

W = H = 4096
steps = synth.optimal_scale_steps(W, H)
a, b, _ = synth.make_pair_torch(W, H, device="cuda")
pa, pb = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
def resident(p):
    buf = torch.zeros(p.numel() + 64, dtype=torch.uint8, device="cuda")
    buf[:p.numel()].copy_(p.reshape(-1))
    return buf[:p.numel()].view(p.shape[0], p.shape[1])
d1, d2 = [resident(p) for p in pa], [resident(p) for p in pb]
out_xy = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
out_corr = torch.empty((H, W), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
def run(stream, stats_ahead, n=20, graph=False):
    with torch.cuda.stream(stream):
        dev = correlation.create_gpu_context(ordinal=0, stream=stream.cuda_stream)
        pc = correlation.PointCorrelations(dev, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
        pc.set_borrow_inputs(True)
        pc.set_stats_ahead(stats_ahead)
        def step():
            pc.first_pass = True
            for i in range(steps + 1):
                k = steps - i
                pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            pc.complete(out_xy=out_xy, out_corr=out_corr)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = None
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                step()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if g is not None:
                g.replay()
            else:
                step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        ref = out_xy.clone()
        pc.close()
        dev.close()
        return ms, ref
ms0, ref0 = run(side, True)
print(f"stream launches, statistics ahead: {ms0:.3f} ms per step")
ms1, ref1 = run(side, False)
print(f"stream launches, in line:          {ms1:.3f} ms per step")
try:
    ms2, ref2 = run(side, False, graph=True)
    print(f"graph replay, in line:             {ms2:.3f} ms per step, same grid: {bool(torch.equal(ref1, ref2))}")
except Exception as e:
    print("graph capture (in line) failed:", repr(e)[:300])
try:
    ms3, ref3 = run(side, True, graph=True)
    print(f"graph replay, statistics ahead:    {ms3:.3f} ms per step, same grid: {bool(torch.equal(ref0, ref3))}")
except Exception as e:
    print("graph capture (stats ahead) failed:", repr(e)[:300])
