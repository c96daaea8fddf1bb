"""The C++ host layer (cybervision_amd/csrc/host/cvhip_host.hpp) on a real GPU: a g++-built program
drives ORB -> matcher -> RANSAC (host sampling + GPU scoring) -> dense correlation through the
reference-shaped C++ classes; results must equal the ctypes path bit for bit and respect the known
geometry."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from cybervision_amd import correlation, orb, pointmatching, synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cpp_host_pipeline(gpu_device, tmp_path):
    exe = tmp_path / "host_pipeline"
    lib_dir = ROOT / "cybervision_amd"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", str(exe), str(ROOT / "tests" / "cpp" / "host_pipeline.cpp"),
                           f"-L{lib_dir}", "-lcvhip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    w, h = 384, 320
    a, _, d = synth.make_pair(w, h, seed=21)
    img1 = synth.add_blocks(a, count=250, seed=3)
    # second view: img2(x, y) = img1(x + d(x, y), y) with the spatially varying integer disparity field, so the
    # only linear relation between matched coordinates is y2 == y1 (a pure shift would be a degenerate
    # configuration for the affine model)
    xs = np.clip(np.arange(w)[None, :] + d, 0, w - 1)
    img2 = np.ascontiguousarray(np.take_along_axis(img1, xs, axis=1))
    img1.tofile(tmp_path / "img1.raw")
    img2.tofile(tmp_path / "img2.raw")
    res = subprocess.run([str(exe), str(tmp_path / "img1.raw"), str(tmp_path / "img2.raw"), str(w), str(h), str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["keypoints1"] > 200 and info["matches"] > 100

    # ORB and matcher: identical to the ctypes path
    kp = np.fromfile(tmp_path / "kp1.bin", dtype=np.uint32).reshape(-1, 10)
    xy1, desc1 = orb.extract_points(gpu_device, img1)
    xy2, desc2 = orb.extract_points(gpu_device, img2)
    assert (kp[:, :2] == xy1).all() and (kp[:, 2:] == desc1).all()
    m = np.fromfile(tmp_path / "matches.bin", dtype=np.uint32).reshape(-1, 4)
    want_m, _ = pointmatching.match_points(gpu_device, xy1, desc1, xy2, desc2, pointmatching.THRESHOLD_AFFINE)
    assert (m == want_m).all()

    # RANSAC: most matches are inliers of a horizontal-line F; F ~ +-[[0,0,0],[0,0,1],[0,-1,0]] up to scale
    assert info["inliers"] > 0.8 * info["matches"]
    F = np.fromfile(tmp_path / "f.bin", dtype=np.float64).reshape(3, 3)
    assert abs(F[2, 2] - 1.0) < 1e-12 and np.abs(F[:2, :2]).max() == 0.0  # affine form, normalised by f[2][2]
    Fn = F / np.abs(F).max()
    assert abs(abs(Fn[1, 2]) - 1) < 1e-6 and abs(Fn[1, 2] + Fn[2, 1]) < 1e-3
    assert abs(Fn[0, 2]) < 1e-3 and abs(Fn[2, 0]) < 1e-3

    # dense: identical to the ctypes path on the same pyramids
    steps = synth.optimal_scale_steps(w, h)
    p1, p2 = synth.box_pyramid(img1, steps), synth.box_pyramid(img2, steps)
    want_xy, want_corr = correlation.correlate_dense(gpu_device, p1, p2, synth.F_HORIZONTAL)
    xy = np.fromfile(tmp_path / "dense_xy.bin", dtype=np.int32).reshape(h, w, 2)
    corr = np.fromfile(tmp_path / "dense_corr.bin", dtype=np.float32).reshape(h, w)
    valid = want_xy[..., 0] >= 0
    assert info["dense_valid"] == int(valid.sum()) and info["levels"] == steps + 1
    assert (xy == want_xy).all() and (corr.view(np.uint32)[valid] == want_corr.view(np.uint32)[valid]).all()
    ys, xs = np.nonzero(valid)
    x2, y2 = want_xy[..., 0][valid], want_xy[..., 1][valid]
    assert (np.abs(x2 + d[y2, x2] - xs) <= 1).mean() > 0.95  # the disparity field is recovered


def test_cpp_host_perspective_find_ransac(gpu_device, oracle, oracle_fm, tmp_path):
    """FundamentalMatrix::new(Perspective, max_dimension).find_ransac through the C++ host layer: the planted
    geometry is recovered, the returned inliers are exactly fits_model of the returned F (oracle arithmetic), and the
    same seed through the one-call C entry (cvhip_find_ransac, what the Python mirror uses) gives the same F."""
    import sys

    sys.path.insert(0, str(ROOT / "tests"))
    import cases
    from cybervision_amd import fundamentalmatrix

    exe = tmp_path / "host_perspective"
    lib_dir = ROOT / "cybervision_amd"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", str(exe), str(ROOT / "tests" / "cpp" / "host_perspective.cpp"),
                           f"-L{lib_dir}", "-lcvhip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    m, truth, _, _ = cases.perspective_matches(n=4000, outlier_frac=0.3, seed=21)
    m.tofile(tmp_path / "matches.bin")
    res = subprocess.run([str(exe), str(tmp_path / "matches.bin"), str(len(m)), "2048", "17", str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    F = np.fromfile(tmp_path / "f_persp.bin", dtype=np.float64).reshape(3, 3)
    inl = np.fromfile(tmp_path / "inliers_persp.bin", dtype=np.uint32).reshape(-1, 4)
    t = fundamentalmatrix.RANSAC_T_PERSPECTIVE * 2048.0
    mask = oracle_fm.fits_model(F, m, t)
    assert info["inliers"] == len(inl) == int(mask.sum()) and (inl == m[mask]).all()
    cnt, _ = oracle.ransac_score(F, m, t)
    assert cnt[0] == len(inl)
    assert F[2, 2] == 1.0 and (mask & truth).sum() > 0.97 * truth.sum() and (mask & ~truth).sum() < 0.1 * (~truth).sum()
    F2, inl2, _ = fundamentalmatrix.FundamentalMatrix(fundamentalmatrix.ProjectionMode.Perspective, 2048.0).find_ransac(gpu_device, m, seed=17)
    assert (F2 == F).all() and (inl2 == inl).all()
