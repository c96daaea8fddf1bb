"""numpy restatement of image::imageops::resize(.., FilterType::Lanczos3) for Luma8 images, as SourceImage::resize
calls it (src/reconstruction.rs:146-162).  TEST INFRASTRUCTURE ONLY (see oracle/cvref.h).

The `image` crate (0.25.10, Cargo.lock:475-476) is a third-party dependency that is NOT vendored under
/root/reference; this follows its published algorithm (imageops/sample.rs: `resize` = `vertical_sample` into an f32
image, then `horizontal_sample`; `lanczos3_kernel`, `sinc`), in float32 like the crate.  Parity unpinned: no fixture
of the reference holds a resized image, and f32 `sin` differs between libm implementations - the tests therefore
compare the device with this module to a stated tolerance (one grey level on < 0.1 % of the pixels).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _sinc(t):
    a = t * F(np.pi)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(t == 0, F(1.0), np.sin(a, dtype=F) / a).astype(F)


def lanczos3_kernel(x):
    x = np.asarray(x, dtype=F)
    return np.where(np.abs(x) < F(3.0), _sinc(x) * _sinc(x / F(3.0)), F(0.0)).astype(F)


def _taps(in_size: int, out_size: int, o: int):
    """(left, normalised f32 weights) of output sample o."""
    ratio = F(in_size) / F(out_size)
    sratio = F(1.0) if ratio < F(1.0) else ratio
    src_support = F(3.0) * sratio
    centre = (F(o) + F(0.5)) * ratio
    left = int(np.floor(centre - src_support))
    left = min(max(left, 0), in_size - 1)
    right = int(np.ceil(centre + src_support))
    right = min(max(right, left + 1), in_size)
    centre = centre - F(0.5)
    w = lanczos3_kernel((np.arange(left, right).astype(F) - centre) / sratio)
    total = F(0.0)
    for v in w:            # `sum += w` in source order, f32
        total = F(total + v)
    return left, (w / total).astype(F)


def _sample_axis0(img_f32, out_size: int):
    """vertical_sample: [h, w] -> [out_size, w], f32 accumulation in tap order."""
    h, w = img_f32.shape
    out = np.zeros((out_size, w), dtype=F)
    for o in range(out_size):
        left, ws = _taps(h, out_size, o)
        t = np.zeros(w, dtype=F)
        for i, wi in enumerate(ws):
            t = (t + img_f32[left + i] * wi).astype(F)
        out[o] = t
    return out


def resize_lanczos3(img, nw: int, nh: int):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    if (nw, nh) == (w, h):
        return img.copy()
    tmp = _sample_axis0(img.astype(F), nh)                 # vertical pass, unclamped f32
    out = _sample_axis0(np.ascontiguousarray(tmp.T), nw).T  # horizontal pass
    out = np.clip(out, F(0.0), F(255.0))
    # FloatNearest -> f32::round (half away from zero); values are >= 0 here
    fl = np.floor(out)
    return (fl + ((out - fl) >= F(0.5))).astype(np.uint8)


def resize_scale(img, scale: float):
    """SourceImage::resize (reconstruction.rs:146-152): dims = (w as f32 * scale) as u32."""
    h, w = img.shape
    return resize_lanczos3(img, int(F(w) * F(scale)), int(F(h) * F(scale)))
