"""How many roots reach each stage of validate_f's LM funnel (thin-SVD pencil) on config 5's matches:
CVHIP_LM_CENSUS=1 python scripts/lm_census.py [size]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
os.environ["CVHIP_LM_CENSUS"] = "1"
from cybervision_amd import correlation, fundamentalmatrix, reconstruction, synth
from test_orb_ransac_gpu import _samples

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
views, K, poses = synth.make_sfm_views(size)
steps = synth.optimal_scale_steps(size, size)
pyr = [synth.box_pyramid(v, steps) for v in views]
dev = correlation.create_gpu_context()
rec = reconstruction.ImageReconstruction(dev)
kp = [rec.extract_keypoints(p) for p in pyr[:2]]
from cybervision_amd import pointmatching
m = pointmatching.match_points(dev, kp[0][0], kp[0][1], kp[1][0], kp[1][1], 48)[0]
print("matches", len(m))
idx = _samples(m, 7, 50000, seed=1)
t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * size
for pipe in (True, False):
    fundamentalmatrix.set_lm_pipeline(dev, pipe)
    t0 = time.perf_counter()
    got = fundamentalmatrix.perspective_models_device(dev, m, idx, t)
    print("pipeline", pipe, "survivors", int(np.isfinite(got[:, :, 0, 0]).sum()), "of", 3 * len(idx), f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
dev.close()
