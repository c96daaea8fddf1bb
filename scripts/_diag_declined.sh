# declined workgroups of the box kernels at levels 1 and 0 and why (ablation build, CVHIP_DEBUG = 32 | reason << 16: counter 3 =
# declined workgroups, counter 0 bits 36.. = those declined for that reason: 0 mixed lines, 1 too wide, 2 too tall, 3 walk too
# large for the candidates it serves)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CVHIP_EXTRA_FLAGS=-DCVHIP_ABLATIONS python3 -m cybervision_amd.build --force > /dev/null 2>&1 || exit 1
for T in ${1:-10 45 60}; do for R in 0 1 2 3; do
  CVHIP_DEBUG=$((32 + 128 + R * 65536)) python3 - $T $R <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from cybervision_amd import correlation, synth
T = float(sys.argv[1]); W = 4096
img1, img2, _ = synth.make_pair_torch(W, W, tilt_deg=T, device="cuda")
steps = synth.optimal_scale_steps(W, W)
d1, d2 = synth.box_pyramid_torch(img1, steps), synth.box_pyramid_torch(img2, steps)
torch.cuda.synchronize()
dev = correlation.create_gpu_context()
pc = correlation.PointCorrelations(dev, (W, W), (W, W), synth.f_tilt(T), correlation.ProjectionMode.Affine)
pc.set_profiling(True, True)
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(d1[k], d2[k], 1.0 / (1 << k))
    c = pc.get_counters()
    if k <= 1:
        c0 = c["candidates"]
        tiles = 2 * ((d1[k].shape[1] + 52) // 53) * (d1[k].shape[0] // 4)
        print(f"tilt {T} k={k}: declined {c['whole_corridor_pixels']} of ~{tiles} workgroups; reason {sys.argv[2]}: {c0 >> 36}")
pc.close(); dev.close()
PY
done; done
