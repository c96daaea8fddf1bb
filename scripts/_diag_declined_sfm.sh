# declined workgroups of the box kernels on config 5's first pair (2048^2 perspective views, F from the true cameras), per reason
# (ablation build, as scripts/_diag_declined.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CVHIP_EXTRA_FLAGS=-DCVHIP_ABLATIONS python3 -m cybervision_amd.build --force > /dev/null 2>&1 || exit 1
for R in 0 1 2 3; do
  CVHIP_DEBUG=$((32 + 128 + R * 65536)) python3 - $R <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from cybervision_amd import correlation, synth
size = 2048
views, K, poses = synth.make_sfm_views(size)
steps = synth.optimal_scale_steps(size, size)
pyr = [[torch.from_numpy(l).cuda() for l in synth.box_pyramid(v, steps)] for v in views]
torch.cuda.synchronize()
F = synth.sfm_true_f(K, poses[0], poses[1])
dev = correlation.create_gpu_context()
pc = correlation.PointCorrelations(dev, (size, size), (size, size), F, correlation.ProjectionMode.Perspective)
pc.set_profiling(True, True)
for i in range(steps + 1):
    k = steps - i
    pc.correlate_images(pyr[0][k], pyr[1][k], 1.0 / (1 << k))
    c = pc.get_counters()
    kt = pc.get_kernel_times()
    if k <= 1:
        tiles = 2 * ((pyr[0][k].shape[1] + 52) // 53) * (pyr[0][k].shape[0] // 4)
        t = " ".join(f"{n}={v['ms']:.3f}" for n, v in kt.items() if v["launches"])
        print(f"k={k}: reason {sys.argv[1]}: {c['candidates'] >> 36} of ~{tiles} workgroups   ({t})")
pc.close(); dev.close()
PY
done
