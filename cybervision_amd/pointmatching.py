"""KeypointMatching (src/pointmatching.rs:29-77) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

THRESHOLD_AFFINE = 32        # pointmatching.rs:8
THRESHOLD_PERSPECTIVE = 48   # pointmatching.rs:9


def match_points(device, xy1, desc1, xy2, desc2, threshold: int):
    """-> (matches[n, 4] uint32 = (x1, y1, x2, y2), dist[n] uint32), stably sorted by distance."""
    xy1 = np.ascontiguousarray(xy1, dtype=np.uint32)
    xy2 = np.ascontiguousarray(xy2, dtype=np.uint32)
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint32)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint32)
    n1, n2 = xy1.shape[0], xy2.shape[0]
    out = np.zeros((max(n1, 1), 4), dtype=np.uint32)
    dist = np.zeros(max(n1, 1), dtype=np.uint32)
    n = C.c_uint32(0)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(_lib.lib().cvhip_match_points(device.handle, p(xy1), p(desc1), n1, p(xy2), p(desc2), n2, threshold,
                                             p(out), p(dist), C.byref(n)), "cvhip_match_points")
    return out[:n.value].copy(), dist[:n.value].copy()


class KeypointMatching:
    def __init__(self, device, points1, points2, perspective: bool = False):
        thr = THRESHOLD_PERSPECTIVE if perspective else THRESHOLD_AFFINE
        self.matches, self.distances = match_points(device, points1[0], points1[1], points2[0], points2[1], thr)
