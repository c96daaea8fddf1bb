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


def set_orientation_guard(device, guard: float):
    """Test hook (cvhip_orb_set_orientation_guard): 1e-9 default, larger = more images through the host libm path,
    0 = the device orientation off.  Results do not depend on it."""
    _lib.check(_lib.lib().cvhip_orb_set_orientation_guard(device.handle, float(guard)), "cvhip_orb_set_orientation_guard")


def extract_points_batch(device, images, cap: int = MAX_KEYPOINTS, progress=None):
    """extract_points for several independent images in ONE call (cvhip_orb_extract_batch): the three host round
    trips of an extraction are paid once for the batch.  -> [(xy, desc)] in the order of `images`."""
    n = len(images)
    if n == 0:
        return []
    keep, ptrs, ws, hs = [], (C.c_void_p * n)(), (C.c_uint32 * n)(), (C.c_uint32 * n)()
    for i, img in enumerate(images):
        if hasattr(img, "data_ptr"):
            ptrs[i], ws[i], hs[i] = img.data_ptr(), int(img.shape[1]), int(img.shape[0])
            keep.append(img)
        else:
            a = np.ascontiguousarray(img, dtype=np.uint8)
            ptrs[i], ws[i], hs[i] = a.ctypes.data, a.shape[1], a.shape[0]
            keep.append(a)
    # (output buffers are kept on the device object: fresh pages would be faulted in one by one under the library's copies)
    cache = device.__dict__.setdefault("_orb_out", {})
    if (n, cap) not in cache:
        cache.clear()
        cache[(n, cap)] = ([np.empty((cap, 2), dtype=np.uint32) for _ in range(n)], [np.empty((cap, 8), dtype=np.uint32) for _ in range(n)])
    xys, descs = cache[(n, cap)]
    pxy = (C.c_void_p * n)(*[a.ctypes.data for a in xys])
    pdesc = (C.c_void_p * n)(*[a.ctypes.data for a in descs])
    counts = (C.c_uint32 * n)()
    cb = _lib.PROGRESS_FN(lambda _u, v: progress(v)) if progress else _lib.NULL_PROGRESS
    _lib.check(_lib.lib().cvhip_orb_extract_batch(device.handle, n, ptrs, ws, hs, cap, pxy, pdesc, counts, cb, None),
               "cvhip_orb_extract_batch")
    del keep
    return [(xys[i][:counts[i]].copy(), descs[i][:counts[i]].copy()) for i in range(n)]


def extract_points_multiscale(device, pyramid, batched: bool = True):
    """match_keypoints' per-image loop (reconstruction.rs:418-458): levels coarse to fine,
    coordinates mapped back with ((x as f32 / scale) as usize), lists concatenated.  batched: the levels go out in one
    cvhip_orb_extract_batch call (same results as level-by-level calls)."""
    steps = len(pyramid) - 1
    order = [steps - i for i in range(steps + 1)]
    results = extract_points_batch(device, [pyramid[k] for k in order]) if batched else [extract_points(device, pyramid[k]) for k in order]
    xs, ds = [], []
    for k, (xy, desc) in zip(order, results):
        scale = np.float32(1.0 / float(1 << k))
        mapped = np.floor(xy.astype(np.float32) / scale).astype(np.uint32)
        xs.append(mapped)
        ds.append(desc)
    return np.concatenate(xs, axis=0), np.concatenate(ds, axis=0)


def extract_points_multiscale_set(device, pyramids):
    """extract_points_multiscale for several images at once: ALL levels of ALL images in one batch (the sparse stage of
    reconstruct() needs the keypoints of every image before the first pair is matched, reconstruction.rs:261-277)."""
    plan = []
    for p in pyramids:
        steps = len(p) - 1
        plan.append([steps - i for i in range(steps + 1)])
    flat = [pyramids[i][k] for i, order in enumerate(plan) for k in order]
    results = extract_points_batch(device, flat)
    out, pos = [], 0
    for order in plan:
        xs, ds = [], []
        for k in order:
            xy, desc = results[pos]
            pos += 1
            scale = np.float32(1.0 / float(1 << k))
            xs.append(np.floor(xy.astype(np.float32) / scale).astype(np.uint32))
            ds.append(desc)
        out.append((np.concatenate(xs, axis=0), np.concatenate(ds, axis=0)))
    return out
