"""Deterministic, integer-only synthetic inputs for the dense-correlation path.

The reference ships no images, so every platform must be able to produce
byte-identical inputs: all arithmetic here is integer (uint32 hashing, 8.8 fixed-point
bilinear value noise), no libm.  Layout follows the reference's `Grid<u8>`: row-major,
``index = width*y + x`` (src/data.rs:22-64).

``make_pair``       img1 = T(x, y), img2(x, y) = T(x + d(x, y), y) with an integer disparity
                    field d, so surviving matches have a known answer.
``box_pyramid``     2x2 integer box-filter pyramid, level k has dims floor(W / 2^k) — the
                    documented stand-in for the `image` crate's Lanczos3 resize, which is
                    upstream of the accelerated boundary (src/reconstruction.rs:146-152).
``level_schedule``  the reference's level loop (src/reconstruction.rs:565-568).
"""
from __future__ import annotations

import math

import numpy as np

_U32 = np.uint64(0xFFFFFFFF)


def hash32(x, y, seed: int):
    """32-bit integer mix of (x, y, seed); x, y are int64 arrays (may be negative)."""
    x = np.asarray(x, dtype=np.int64).astype(np.uint64) & _U32
    y = np.asarray(y, dtype=np.int64).astype(np.uint64) & _U32
    h = (x * np.uint64(0x9E3779B1)) ^ (y * np.uint64(0x85EBCA77)) ^ np.uint64((seed * 0xC2B2AE3D) & 0xFFFFFFFF)
    h &= _U32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & _U32
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & _U32
    h ^= h >> np.uint64(15)
    return h


def _octave(xs, ys, cell: int, seed: int):
    """Bilinear value noise with lattice spacing `cell`, 8.8 fixed point, result 0..255."""
    ix = np.floor_divide(xs, cell)
    iy = np.floor_divide(ys, cell)
    fx = ((xs - ix * cell) * 256) // cell
    fy = ((ys - iy * cell) * 256) // cell
    l00 = (hash32(ix, iy, seed) & np.uint64(0xFF)).astype(np.int64)
    l10 = (hash32(ix + 1, iy, seed) & np.uint64(0xFF)).astype(np.int64)
    l01 = (hash32(ix, iy + 1, seed) & np.uint64(0xFF)).astype(np.int64)
    l11 = (hash32(ix + 1, iy + 1, seed) & np.uint64(0xFF)).astype(np.int64)
    top = l00 * (256 - fx) + l10 * fx
    bot = l01 * (256 - fx) + l11 * fx
    return (top * (256 - fy) + bot * fy) >> 16


def texture(xs, ys, seed: int = 1234):
    """T(x, y): mean of four value-noise octaves (cells 4/16/64/256 px), 0..255, int64."""
    xs = np.asarray(xs, dtype=np.int64)
    ys = np.asarray(ys, dtype=np.int64)
    acc = np.zeros(np.broadcast(xs, ys).shape, dtype=np.int64)
    for k, cell in enumerate((4, 16, 64, 256)):
        acc = acc + _octave(xs, ys, cell, seed + 17 * k)
    return (acc + 2) >> 2


def _tri(v, period: int):
    """Integer triangle wave in [-256, 256] with the given period."""
    u = np.mod(v, period)
    t = (u * 1024) // period
    return np.where(t < 512, t - 256, 768 - t)


def disparity(width: int, height: int):
    """d(x, y) = (A * tri(x, P) * tri(y, Q)) >> 16, A = W/64, P = W/4, Q = H/4; |d| <= A."""
    a = max(width // 64, 1)
    p = max(width // 4, 4)
    q = max(height // 4, 4)
    xs = np.arange(width, dtype=np.int64)[None, :]
    ys = np.arange(height, dtype=np.int64)[:, None]
    return (a * _tri(xs, p) * _tri(ys, q)) >> 16


def make_pair(width: int, height: int | None = None, seed: int = 1234, sem_style: bool = False,
              tilt_deg: float = 0.0):
    """Return (img1, img2, d): two uint8 (H, W) images and the int64 disparity field.  The displacement is
    d along the direction (cos, sin) of tilt_deg - the epipolar direction of f_tilt(tilt_deg) - rounded to
    integer pixels (16.16 fixed point), so a tilted F sees a geometrically consistent pair."""
    height = width if height is None else height
    xs = np.arange(width, dtype=np.int64)[None, :]
    ys = np.arange(height, dtype=np.int64)[:, None]
    d = disparity(width, height)
    t1 = texture(xs, ys, seed)
    if tilt_deg == 0.0:
        t2 = texture(xs + d, ys + 0 * xs, seed)
    else:
        ci = int(round(math.cos(math.radians(tilt_deg)) * 65536.0))
        si = int(round(math.sin(math.radians(tilt_deg)) * 65536.0))
        t2 = texture(xs + ((d * ci + 32768) >> 16), ys + ((d * si + 32768) >> 16), seed)
    if sem_style:
        # linear shading ramp of +-16 grey levels and 0.2 % salt noise (independent per image), both integer
        ramp = ((xs * 32) // max(width, 1)) - 16
        t1 = t1 + ramp
        t2 = t2 + ramp
        for t, s in ((t1, seed + 1), (t2, seed + 2)):
            h = hash32(xs + 0 * ys, ys + 0 * xs, s)
            salt = (h % np.uint64(500)) == 0
            t[salt] = 255
    img1 = np.clip(t1, 0, 255).astype(np.uint8)
    img2 = np.clip(t2, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img1), np.ascontiguousarray(img2), d


# ---------------------------------------------------------------------------------------------
# The same generator in torch int64 arithmetic (any device): byte-identical to make_pair - every step is integer, and
# the low 32 bits of a wrapping 64-bit product are the low 32 bits of the exact one - but runs on the GPU in
# milliseconds where numpy needs half a minute for a 4096^2 pair (bench.py builds six such pairs).
# ---------------------------------------------------------------------------------------------
def _hash32_t(x, y, seed: int):
    import torch  # noqa: F401

    m = 0xFFFFFFFF
    h = ((x & m) * 0x9E3779B1) ^ ((y & m) * 0x85EBCA77) ^ ((seed * 0xC2B2AE3D) & m)
    h = h & m
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & m
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & m
    return h ^ (h >> 15)


def _octave_t(xs, ys, cell: int, seed: int):
    import torch

    ix = torch.div(xs, cell, rounding_mode="floor")
    iy = torch.div(ys, cell, rounding_mode="floor")
    fx = torch.div((xs - ix * cell) * 256, cell, rounding_mode="floor")
    fy = torch.div((ys - iy * cell) * 256, cell, rounding_mode="floor")
    l00 = _hash32_t(ix, iy, seed) & 0xFF
    l10 = _hash32_t(ix + 1, iy, seed) & 0xFF
    l01 = _hash32_t(ix, iy + 1, seed) & 0xFF
    l11 = _hash32_t(ix + 1, iy + 1, seed) & 0xFF
    top = l00 * (256 - fx) + l10 * fx
    bot = l01 * (256 - fx) + l11 * fx
    return (top * (256 - fy) + bot * fy) >> 16


def _texture_t(xs, ys, seed: int):
    acc = None
    for k, cell in enumerate((4, 16, 64, 256)):
        o = _octave_t(xs, ys, cell, seed + 17 * k)
        acc = o if acc is None else acc + o
    return (acc + 2) >> 2


def make_pair_torch(width: int, height: int | None = None, seed: int = 1234, tilt_deg: float = 0.0, device="cuda"):
    """make_pair(width, height, seed, tilt_deg=tilt_deg) as torch uint8 tensors on `device` (same bytes; tested)."""
    import torch

    height = width if height is None else height
    xs = torch.arange(width, dtype=torch.int64, device=device)[None, :].expand(height, width)
    ys = torch.arange(height, dtype=torch.int64, device=device)[:, None].expand(height, width)

    def tri(v, period):
        t = torch.div(torch.remainder(v, period) * 1024, period, rounding_mode="floor")
        return torch.where(t < 512, t - 256, 768 - t)

    a, p, q = max(width // 64, 1), max(width // 4, 4), max(height // 4, 4)
    d = (a * tri(xs, p) * tri(ys, q)) >> 16
    t1 = _texture_t(xs, ys, seed)
    if tilt_deg == 0.0:
        t2 = _texture_t(xs + d, ys, seed)
    else:
        ci = int(round(math.cos(math.radians(tilt_deg)) * 65536.0))
        si = int(round(math.sin(math.radians(tilt_deg)) * 65536.0))
        t2 = _texture_t(xs + ((d * ci + 32768) >> 16), ys + ((d * si + 32768) >> 16), seed)
    return t1.clamp(0, 255).to(torch.uint8).contiguous(), t2.clamp(0, 255).to(torch.uint8).contiguous(), d


def box_pyramid_torch(img, steps: int):
    """box_pyramid on a torch uint8 tensor (same bytes)."""
    import torch

    out = [img.contiguous()]
    for _ in range(steps):
        v = out[-1]
        h2, w2 = v.shape[0] // 2, v.shape[1] // 2
        v = v[: h2 * 2, : w2 * 2].to(torch.int32)
        s4 = v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2]
        out.append(((s4 + 2) >> 2).to(torch.uint8).contiguous())
    return out


def add_blocks(img: np.ndarray, count: int = 400, seed: int = 77):
    """Overlay random axis-aligned rectangles (contrast +-40) so FAST has corners to find."""
    out = img.astype(np.int64)
    h, w = img.shape
    idx = np.arange(count, dtype=np.int64)
    x0 = (hash32(idx, 0 * idx, seed) % np.uint64(max(w - 8, 1))).astype(np.int64)
    y0 = (hash32(idx, 0 * idx + 1, seed) % np.uint64(max(h - 8, 1))).astype(np.int64)
    bw = 8 + (hash32(idx, 0 * idx + 2, seed) % np.uint64(max(w // 16, 1))).astype(np.int64)
    bh = 8 + (hash32(idx, 0 * idx + 3, seed) % np.uint64(max(h // 16, 1))).astype(np.int64)
    sg = np.where((hash32(idx, 0 * idx + 4, seed) & np.uint64(1)) == 0, 40, -40)
    for i in range(count):
        out[y0[i]:y0[i] + bh[i], x0[i]:x0[i] + bw[i]] += sg[i]
    return np.clip(out, 0, 255).astype(np.uint8)


def box_downsample(img: np.ndarray) -> np.ndarray:
    """One 2x2 box-filter step with rounding; dims floor(w/2) x floor(h/2)."""
    h, w = img.shape
    h2, w2 = h // 2, w // 2
    v = img[: h2 * 2, : w2 * 2].astype(np.uint16)
    s = v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2]
    return np.ascontiguousarray(((s + 2) >> 2).astype(np.uint8))


def box_pyramid(img: np.ndarray, steps: int):
    """[level 0 (full res), level 1 (1/2), ..., level `steps` (1/2^steps)]."""
    out = [np.ascontiguousarray(img)]
    for _ in range(steps):
        out.append(box_downsample(out[-1]))
    return out


def optimal_scale_steps(width: int, height: int, min_size: int = 64) -> int:
    """PointCorrelations::optimal_scale_steps (src/correlation/mod.rs:542-550)."""
    m = min(width, height)
    if m <= min_size:
        return 0
    return int(math.floor(math.log2(m / min_size)))


def level_schedule(width: int, height: int, min_size: int = 64):
    """[(k, scale)] coarse to fine, scale = 1 / (1 << k) (src/reconstruction.rs:565-566)."""
    steps = optimal_scale_steps(width, height, min_size)
    return [(steps - i, 1.0 / float(1 << (steps - i))) for i in range(steps + 1)]


F_HORIZONTAL = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])


def f_tilt(theta_deg: float) -> np.ndarray:
    """Affine fundamental matrix whose epipolar lines are tilted by theta."""
    t = math.radians(theta_deg)
    s, c = math.sin(t), math.cos(t)
    return np.array([[0.0, 0.0, -s], [0.0, 0.0, c], [s, -c, 0.0]])


# ---------------------------------------------------------------------------------------------
# Three-view perspective scene (BASELINE config 5): one textured surface seen by three pinhole cameras.
# The surface is a depth map Z0 over view 0's image plane (the "canvas"); view i is the canvas image
# resampled through the exact projective map of camera i, so every pair of views has a true fundamental
# matrix F_ij = K^-T [t_ij]x R_ij K^-1.  Only + - * / and floor on float64 (IEEE-exact, no libm per pixel),
# so every platform renders identical bytes.
# ---------------------------------------------------------------------------------------------
def _rot(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def sfm_cameras(size: int):
    """K and the three poses (R_i, t_i), X_cam_i = R_i X + t_i.  Mostly sideways translation with very small
    rotations: the reference's rank test on the F22-normalised matrix (fundamentalmatrix.rs:362-366, 418-423)
    accepts such geometry, which is also what its stereo use case looks like."""
    f = 0.9 * size
    K = np.array([[f, 0.0, size / 2.0], [0.0, f, size / 2.0], [0.0, 0.0, 1.0]])
    poses = [(np.eye(3), np.zeros(3)),
             (_rot(0.0005, -0.0010, 0.0004), np.array([-0.060, 0.002, 0.001])),
             (_rot(-0.0004, 0.0008, -0.0003), np.array([0.050, -0.0015, -0.0012]))]
    return K, poses


def sfm_true_f(K, pose_i, pose_j):
    """F with x_j^T F x_i = 0, normalised by F[2][2] like the reference's models."""
    Ri, ti = pose_i
    Rj, tj = pose_j
    R = Rj @ Ri.T
    t = tj - R @ ti
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    return F / F[2, 2]


def _sfm_depth(x, y, size: int):
    """Z0 over the canvas: a smooth bump, 1.0 at the border down to 0.85 in the middle (polynomial only)."""
    u, v = x / size, y / size
    return 1.0 - 0.15 * (16.0 * u * (1.0 - u) * v * (1.0 - v))


def _bilinear_u8(img: np.ndarray, x, y):
    """Bilinear sample of a u8 image at float coordinates, 8.8 fixed-point weights, clamped at the border."""
    h, w = img.shape
    xf, yf = np.floor(x), np.floor(y)
    fx = np.floor((x - xf) * 256.0).astype(np.int64)
    fy = np.floor((y - yf) * 256.0).astype(np.int64)
    x0 = np.clip(xf.astype(np.int64), 0, w - 1)
    y0 = np.clip(yf.astype(np.int64), 0, h - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    y1 = np.clip(y0 + 1, 0, h - 1)
    v = img.astype(np.int64)
    top = v[y0, x0] * (256 - fx) + v[y0, x1] * fx
    bot = v[y1, x0] * (256 - fx) + v[y1, x1] * fx
    return ((top * (256 - fy) + bot * fy + 32768) >> 16).astype(np.uint8)


def make_sfm_views(size: int, seed: int = 1234, blocks: int | None = None):
    """-> (views [3 x (size, size) uint8], K, poses).  The canvas carries the value-noise texture plus a layer of
    rectangles (FAST corners); it is `size + 2*margin` wide so that all three views stay inside it."""
    margin = max(size // 8, 16)
    cw = size + 2 * margin
    xs = np.arange(cw, dtype=np.int64)[None, :]
    ys = np.arange(cw, dtype=np.int64)[:, None]
    canvas = np.clip(texture(xs, ys, seed), 0, 255).astype(np.uint8)
    canvas = add_blocks(canvas, count=blocks if blocks is not None else max(size * size // 2600, 40), seed=seed + 70)
    K, poses = sfm_cameras(size)
    Ki = np.linalg.inv(K)
    views = []
    gx, gy = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64))
    for R, t in poses:
        # invert p_i = proj(K (R Z0(p0) K^-1 p0 + t)) for p0 by fixed-point iteration on the displacement
        px, py = gx.copy(), gy.copy()
        for _ in range(12):
            z = _sfm_depth(px, py, size)
            X = [z * (Ki[r, 0] * px + Ki[r, 1] * py + Ki[r, 2]) for r in range(3)]
            Y = [R[r, 0] * X[0] + R[r, 1] * X[1] + R[r, 2] * X[2] + t[r] for r in range(3)]
            q = [K[r, 0] * Y[0] + K[r, 1] * Y[1] + K[r, 2] * Y[2] for r in range(3)]
            fx, fy = q[0] / q[2], q[1] / q[2]
            px, py = px + (gx - fx), py + (gy - fy)
        views.append(np.ascontiguousarray(_bilinear_u8(canvas, px + margin, py + margin)))
    return views, K, poses
