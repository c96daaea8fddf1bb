"""Shared test cases for the dense-correlation path (inputs are generated, never read from
the reference tree)."""
import math

import numpy as np

from cybervision_amd import synth


def perspective_f(w, h):
    """A plausible rank-2 perspective fundamental matrix (camera translating mostly along x
    with a little rotation), F = K^-T [t]x R K^-1."""
    f = 1.2 * max(w, h)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    a, b = math.radians(2.0), math.radians(-1.5)
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    R = Ry @ Rx
    t = np.array([1.0, 0.05, 0.1])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    return F / F[2, 2] if abs(F[2, 2]) > 1e-12 else F / np.abs(F).max()


def make_case(name):
    """name -> dict(img1, img2, F, projection, steps). Deterministic."""
    if name == "h256":
        a, b, _ = synth.make_pair(256, 256)
        F, proj = synth.F_HORIZONTAL, 0
    elif name == "sem320x200":
        a, b, _ = synth.make_pair(320, 200, seed=99, sem_style=True)
        F, proj = synth.F_HORIZONTAL, 0
    elif name == "tilt3_200x150":
        a, b, _ = synth.make_pair(200, 150, seed=7)
        F, proj = synth.f_tilt(3.0), 0
    elif name == "tilt60_150x200":
        a, b, _ = synth.make_pair(150, 200, seed=8)
        F, proj = synth.f_tilt(60.0), 0
    elif name == "tilt05_400x300":
        # slightly unrectified pair: the box filter handles most workgroups, the ones where a line steps inside a
        # pixel's interval go to the candidate filter (both kernels contribute to one grid)
        a, b, _ = synth.make_pair(400, 300, seed=13)
        F, proj = synth.f_tilt(0.5), 0
    elif name == "vert_200x260":
        # vertical epipolar lines: tall displacement boxes, declined by the box filter
        a0, b0, _ = synth.make_pair(260, 200, seed=17)
        a, b = np.ascontiguousarray(a0.T), np.ascontiguousarray(b0.T)
        F, proj = synth.f_tilt(90.0), 0
    elif name == "ragged_dims":
        # the two images have different sizes (the reverse grid has its own dims)
        a, b0, _ = synth.make_pair(190, 170, seed=11)
        b = np.ascontiguousarray(np.pad(b0, ((0, 14), (0, 23)), mode="edge"))
        F, proj = synth.f_tilt(-2.0), 0
    elif name == "persp_240x180":
        a, b, _ = synth.make_pair(240, 180, seed=21)
        F, proj = perspective_f(240, 180), 1
    elif name == "flat":
        a = np.full((96, 128), 77, dtype=np.uint8)
        b = a.copy()
        F, proj = synth.F_HORIZONTAL, 0
    elif name == "tiny_single_level":
        a, b, _ = synth.make_pair(64, 48, seed=3)
        F, proj = synth.F_HORIZONTAL, 0
    else:
        raise KeyError(name)
    steps = synth.optimal_scale_steps(a.shape[1], a.shape[0])
    return dict(name=name, img1=a, img2=b, F=np.asarray(F, dtype=np.float64), projection=proj, steps=steps)


CASES = ["h256", "sem320x200", "tilt3_200x150", "tilt60_150x200", "tilt05_400x300", "vert_200x260", "ragged_dims",
         "persp_240x180", "flat", "tiny_single_level"]
GOLDEN_CASES = ["tilt3_200x150", "persp_240x180", "ragged_dims"]


def pyramids(case):
    return synth.box_pyramid(case["img1"], case["steps"]), synth.box_pyramid(case["img2"], case["steps"])


def perspective_matches(n=4000, outlier_frac=0.3, seed=5, size=2048):
    """Two pinhole views of random 3-D points (mostly sideways translation, so that the reference's rank test
    fundamentalmatrix.rs:356-362 accepts the true F), rounded to integer pixels, plus uniform outliers.
    -> (matches [n, 4] uint32, inlier truth [n] bool, exact coordinates [n, 4] float64, F_true [3, 3])."""
    rng = np.random.default_rng(seed)
    f = 0.88 * size
    K = np.array([[f, 0, size / 2.0], [0, f, size / 2.0], [0, 0, 1.0]])
    ax, ay = 0.0005, -0.001
    Ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    R = Ry @ Rx
    t = np.array([1.0, 0.03, 0.02])
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)], axis=1)
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    x2 = (K @ ((R @ X.T).T + t).T).T
    x2 = x2[:, :2] / x2[:, 2:]
    exact = np.concatenate([x1, x2], axis=1)
    out = rng.random(n) < outlier_frac
    m = np.round(exact)
    m[out, 2:] = rng.integers(0, size, size=(int(out.sum()), 2))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    return m.clip(0, size - 1).astype(np.uint32), ~out, exact, F / F[2, 2]
