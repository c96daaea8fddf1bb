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
