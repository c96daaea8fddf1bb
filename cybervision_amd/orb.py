"""orb::extract_points / optimal_scale_steps (src/orb.rs:50-84, 407-415) over the C ABI, and the
multi-scale driver of match_keypoints (src/reconstruction.rs:407-459)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

MAX_KEYPOINTS = 10_000            # orb.rs:41
KEYPOINT_SCALE_MIN_SIZE = 256     # orb.rs:32


def optimal_scale_steps(width: int, height: int) -> int:
    m = min(width, height)
    if m <= KEYPOINT_SCALE_MIN_SIZE:
        return 0
    return int(math.floor(math.log2(m / KEYPOINT_SCALE_MIN_SIZE)))


def extract_points(device, img, cap: int = MAX_KEYPOINTS, progress=None):
    """-> (xy[n, 2] uint32, desc[n, 8] uint32), in the reference's order.  progress(pos): the reference's
    `Option<&PL>` (orb.rs:43-53), called at its stage boundaries."""
    if hasattr(img, "data_ptr"):
        ptr, w, h = C.c_void_p(img.data_ptr()), int(img.shape[1]), int(img.shape[0])
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        ptr, w, h = C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0]
    xy = np.zeros((cap, 2), dtype=np.uint32)
    desc = np.zeros((cap, 8), dtype=np.uint32)
    n = C.c_uint32(0)
    cb = _lib.PROGRESS_FN(lambda _u, v: progress(v)) if progress else _lib.NULL_PROGRESS
    _lib.check(_lib.lib().cvhip_orb_extract(device.handle, ptr, w, h, cap, C.c_void_p(xy.ctypes.data),
                                            C.c_void_p(desc.ctypes.data), C.byref(n), cb, None), "cvhip_orb_extract")
    return xy[:n.value].copy(), desc[:n.value].copy()


def extract_points_multiscale(device, pyramid):
    """match_keypoints' per-image loop (reconstruction.rs:418-458): levels coarse to fine,
    coordinates mapped back with ((x as f32 / scale) as usize), lists concatenated."""
    steps = len(pyramid) - 1
    xs, ds = [], []
    for i in range(steps + 1):
        k = steps - i
        scale = np.float32(1.0 / float(1 << k))
        xy, desc = extract_points(device, pyramid[k])
        mapped = np.floor(xy.astype(np.float32) / scale).astype(np.uint32)
        xs.append(mapped)
        ds.append(desc)
    return np.concatenate(xs, axis=0), np.concatenate(ds, axis=0)
