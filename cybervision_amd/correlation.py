"""Host-side front-end of the dense-correlation C ABI (include/cvhip.h).

Mirrors how the reference drives its GPU backend: ``create_gpu_context`` (correlation/mod.rs:145),
``PointCorrelations::{new, correlate_images, complete, optimal_scale_steps}``
(correlation/mod.rs:150-245, 542-550) and the level loop of ``correlate_dense``
(reconstruction.rs:554-588).  Everything here forwards to libcvhip.so — no compute in Python.

Images may be numpy uint8 arrays (host) or anything exposing ``data_ptr()`` (torch CUDA uint8
tensors); the library detects host vs device pointers itself.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib
from .synth import optimal_scale_steps  # noqa: F401  (PointCorrelations::optimal_scale_steps)


class ProjectionMode(IntEnum):  # correlation/mod.rs:43-47
    Affine = 0
    Perspective = 1


class HardwareMode(IntEnum):  # correlation/mod.rs:49-54
    Gpu = 0
    GpuLowPower = 1
    Cpu = 2


class CorrelationDirection(IntEnum):  # correlation/mod.rs:77-81
    Forward = 0
    Reverse = 1


def _ptr_shape(img):
    """(pointer, width, height, keepalive) of a 2-D uint8 image on host or device."""
    if hasattr(img, "data_ptr"):  # torch tensor
        assert img.dim() == 2 and img.element_size() == 1 and img.is_contiguous()
        return C.c_void_p(img.data_ptr()), int(img.shape[1]), int(img.shape[0]), img
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 2
    return C.c_void_p(a.ctypes.data), int(a.shape[1]), int(a.shape[0]), a


class GpuDevice:
    """create_gpu_context(HardwareMode) -> GpuDevice (correlation/mod.rs:145-147)."""

    def __init__(self, hardware_mode: HardwareMode = HardwareMode.Gpu, ordinal: int = -1, stream: int | None = None):
        """stream: optional raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) to
        submit all work to; default is a private stream."""
        if hardware_mode == HardwareMode.Cpu:
            raise ValueError("HardwareMode.Cpu has no GPU device (the CPU path is the reference's own)")
        self._h = C.c_void_p()
        self.hardware_mode, self.ordinal = hardware_mode, ordinal
        self._side_handles: list["GpuDevice"] = []  # further handles on the same GPU, each with a stream of its own (side_handle)
        low = int(hardware_mode == HardwareMode.GpuLowPower)
        if stream is None:
            rc = _lib.lib().cvhip_device_create(low, ordinal, C.byref(self._h))
        else:  # 0 is a valid value: HIP's default (null) stream, which is torch's default current stream
            rc = _lib.lib().cvhip_device_create_on_stream(low, ordinal, C.c_void_p(stream), C.byref(self._h))
        _lib.check(rc, "cvhip_device_create")

    @property
    def handle(self):
        return self._h

    def name(self) -> str:
        return _lib.lib().cvhip_device_name(self._h).decode()

    def synchronize(self):
        _lib.check(_lib.lib().cvhip_device_synchronize(self._h), "cvhip_device_synchronize")

    def side_handle(self, i: int) -> "GpuDevice":
        """The i-th further device handle on this handle's GPU, with a private stream, created on first use and closed with
        this one: independent jobs (the pairs of reconstruct_dense) run side by side on them."""
        while len(self._side_handles) <= i:
            self._side_handles.append(GpuDevice(self.hardware_mode, self.ordinal, None))
        return self._side_handles[i]

    def close(self):
        for d in getattr(self, "_side_handles", []):
            d.close()
        self._side_handles = []
        if getattr(self, "_h", None):
            _lib.lib().cvhip_device_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_gpu_context(hardware_mode: HardwareMode = HardwareMode.Gpu, ordinal: int = -1,
                       stream: int | None = None) -> GpuDevice:
    return GpuDevice(hardware_mode, ordinal, stream)


class PointCorrelations:
    """PointCorrelations with a GPU context (correlation/mod.rs:63-245, GPU branch)."""

    def __init__(self, device: GpuDevice, img1_dimensions, img2_dimensions, fundamental_matrix,
                 projection_mode: ProjectionMode = ProjectionMode.Affine):
        self.device = device
        self.w1, self.h1 = int(img1_dimensions[0]), int(img1_dimensions[1])
        self.w2, self.h2 = int(img2_dimensions[0]), int(img2_dimensions[1])
        F = np.ascontiguousarray(np.asarray(fundamental_matrix, dtype=np.float64).reshape(9))
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cvhip_ctx_create(device.handle, self.w1, self.h1, self.w2, self.h2,
                                               int(projection_mode), F.ctypes.data_as(C.POINTER(C.c_double)),
                                               C.byref(self._h)), "cvhip_ctx_create")
        self.first_pass = True
        self.selected_hardware = f"GPU {device.name()}"
        self.correlated_points = None

    def get_selected_hardware(self) -> str:
        return self.selected_hardware

    # -- the four GpuContext calls (correlation/gpu/mod.rs:172-362) ---------------------------
    def correlate_images_step(self, img1, img2, scale: float, direction: CorrelationDirection,
                              progress=None):
        p1, w1, h1, k1 = _ptr_shape(img1)
        p2, w2, h2, k2 = _ptr_shape(img2)
        cb = _lib.PROGRESS_FN(lambda _u, v: progress(v)) if progress else _lib.NULL_PROGRESS
        _lib.check(_lib.lib().cvhip_correlate_images(self._h, p1, w1, h1, p2, w2, h2, scale, int(self.first_pass),
                                                     int(direction), cb, None), "cvhip_correlate_images")
        del k1, k2

    def cross_check_filter(self, scale: float, direction: CorrelationDirection):
        _lib.check(_lib.lib().cvhip_cross_check_filter(self._h, scale, int(direction)), "cvhip_cross_check_filter")

    def correlate_images(self, img1, img2, scale: float, progress=None, fused: bool = True):
        """PointCorrelations::correlate_images (mod.rs:217-245).  fused=True issues the single
        cvhip_correlate_level call, fused=False the reference's four backend calls."""
        if fused:
            p1, w1, h1, k1 = _ptr_shape(img1)
            p2, w2, h2, k2 = _ptr_shape(img2)
            cb = _lib.PROGRESS_FN(lambda _u, v: progress(v)) if progress else _lib.NULL_PROGRESS
            _lib.check(_lib.lib().cvhip_correlate_level(self._h, p1, w1, h1, p2, w2, h2, scale,
                                                        int(self.first_pass), cb, None), "cvhip_correlate_level")
            del k1, k2
        else:
            self.correlate_images_step(img1, img2, scale, CorrelationDirection.Forward, progress)
            self.correlate_images_step(img2, img1, scale, CorrelationDirection.Reverse, progress)
            self.cross_check_filter(scale, CorrelationDirection.Forward)
            self.cross_check_filter(scale, CorrelationDirection.Reverse)
        self.first_pass = False

    def complete(self, direction: CorrelationDirection = CorrelationDirection.Forward, out_xy=None, out_corr=None):
        """complete() (mod.rs:208-215): returns (xy[h, w, 2] int32 with -1 = None, corr[h, w] f32)."""
        w, h = (self.w1, self.h1) if direction == CorrelationDirection.Forward else (self.w2, self.h2)
        if out_xy is None:
            out_xy = np.empty((h, w, 2), dtype=np.int32)
            out_corr = np.empty((h, w), dtype=np.float32)
        pxy = C.c_void_p(out_xy.data_ptr() if hasattr(out_xy, "data_ptr") else out_xy.ctypes.data)
        pc = None
        if out_corr is not None:
            pc = C.c_void_p(out_corr.data_ptr() if hasattr(out_corr, "data_ptr") else out_corr.ctypes.data)
        _lib.check(_lib.lib().cvhip_complete_dir(self._h, int(direction), pxy, pc), "cvhip_complete_dir")
        if direction == CorrelationDirection.Forward:
            self.correlated_points = (out_xy, out_corr)
        return out_xy, out_corr

    def complete_packed(self, direction: CorrelationDirection = CorrelationDirection.Forward, out_cells=None, out_corr=None):
        """cvhip_complete_packed: (cells[h, w] uint32 = y2 << 16 | x2 with 0xFFFFFFFF = None, corr[h, w] f32) - 8 bytes per
        cell over PCIe instead of 12; unpack_cells gives complete()'s xy."""
        w, h = (self.w1, self.h1) if direction == CorrelationDirection.Forward else (self.w2, self.h2)
        if out_cells is None:
            out_cells = np.empty((h, w), dtype=np.uint32)
            out_corr = np.empty((h, w), dtype=np.float32)
        pxy = C.c_void_p(out_cells.data_ptr() if hasattr(out_cells, "data_ptr") else out_cells.ctypes.data)
        pc = None
        if out_corr is not None:
            pc = C.c_void_p(out_corr.data_ptr() if hasattr(out_corr, "data_ptr") else out_corr.ctypes.data)
        _lib.check(_lib.lib().cvhip_complete_packed(self._h, int(direction), pxy, pc), "cvhip_complete_packed")
        return out_cells, out_corr

    @staticmethod
    def unpack_cells(cells):
        """[h, w] uint32 packed cells -> [h, w, 2] int32 (x, y), (-1, -1) = None."""
        cells = np.asarray(cells)
        xy = np.empty(cells.shape + (2,), dtype=np.int32)
        none = cells == 0xFFFFFFFF
        xy[..., 0] = np.where(none, -1, (cells & 0xFFFF).astype(np.int32))
        xy[..., 1] = np.where(none, -1, (cells >> 16).astype(np.int32))
        return xy

    def triangulate_affine(self):
        """AffineTriangulation::triangulate (triangulation.rs:268-330) straight from the device grid:
        -> (points3d[n, 3] float64 = (x, y, |p1 - p2|), p2[n, 2] uint32), one row per Some cell in scan order."""
        n = C.c_uint64(0)
        _lib.check(_lib.lib().cvhip_triangulate_affine(self._h, None, None, 0, C.byref(n)), "cvhip_triangulate_affine")
        pts = np.zeros((max(n.value, 1), 3), dtype=np.float64)
        p2 = np.zeros((max(n.value, 1), 2), dtype=np.uint32)
        if n.value:
            _lib.check(_lib.lib().cvhip_triangulate_affine(self._h, C.c_void_p(pts.ctypes.data), C.c_void_p(p2.ctypes.data),
                                                           n.value, C.byref(n)), "cvhip_triangulate_affine")
        return pts[:n.value].copy(), p2[:n.value].copy()

    def extend_tracks(self, track_p1, max_dimension2: int):
        """Triangulation::extend_tracks (triangulation.rs:1330-1419) straight from the device grid.  track_p1: [n, 2]
        int32, (-1, -1) = the track has no point in image 1.  -> (track_p2 [n, 2] int32 with (-1, -1) = nothing to add,
        new_p1 [m, 2] uint32, new_p2 [m, 2] uint32): the image-2 points for the existing tracks and the new tracks."""
        tp1 = np.ascontiguousarray(np.asarray(track_p1, dtype=np.int32).reshape(-1, 2))
        n = len(tp1)
        tp2 = np.full((max(n, 1), 2), -1, dtype=np.int32)
        m = C.c_uint64(0)
        p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        _lib.check(_lib.lib().cvhip_extend_tracks(self._h, p(tp1) if n else None, n, int(max_dimension2), p(tp2) if n else None,
                                                  None, None, 0, C.byref(m)), "cvhip_extend_tracks")
        n1 = np.zeros((max(m.value, 1), 2), dtype=np.uint32)
        n2 = np.zeros((max(m.value, 1), 2), dtype=np.uint32)
        if m.value:
            _lib.check(_lib.lib().cvhip_extend_tracks(self._h, p(tp1) if n else None, n, int(max_dimension2),
                                                      p(tp2) if n else None, p(n1), p(n2), m.value, C.byref(m)),
                       "cvhip_extend_tracks")
        return tp2[:n].copy(), n1[:m.value].copy(), n2[:m.value].copy()

    # -- measurement / sharding hooks -----------------------------------------------------------
    def set_profiling(self, time_kernels, count_candidates: bool):
        """time_kernels: 0/False off, 1/True every kernel class, 2 the search class only (include/cvhip.h)."""
        _lib.check(_lib.lib().cvhip_ctx_set_profiling(self._h, int(time_kernels), int(count_candidates)),
                   "cvhip_ctx_set_profiling")

    def get_profile(self, reset: bool = True):
        n, ms, cand = C.c_uint32(0), C.c_double(0.0), C.c_uint64(0)
        _lib.check(_lib.lib().cvhip_ctx_get_profile(self._h, C.byref(n), C.byref(ms), C.byref(cand), int(reset)),
                   "cvhip_ctx_get_profile")
        return {"launches": n.value, "search_ms": ms.value, "candidates": cand.value}

    KERNEL_CLASSES = ("window_stats", "search_range", "search", "exact", "cross_check", "expand", "search_filter")

    def get_kernel_times(self, reset: bool = True):
        ms = (C.c_double * 7)()
        n = (C.c_uint32 * 7)()
        _lib.check(_lib.lib().cvhip_ctx_get_kernel_times(self._h, ms, n, int(reset)), "cvhip_ctx_get_kernel_times")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(self.KERNEL_CLASSES)}

    def get_counters(self, reset: bool = True):
        arr = (C.c_uint64 * 4)()
        _lib.check(_lib.lib().cvhip_ctx_get_counters(self._h, arr, int(reset)), "cvhip_ctx_get_counters")
        return {"candidates": arr[0], "exact_evals": arr[1], "multi_contender_pixels": arr[2],
                "whole_corridor_pixels": arr[3]}

    def set_borrow_inputs(self, borrow: bool):
        """Device-resident level images are used in place (no copy): the caller guarantees 64 readable bytes after
        the last pixel of each and keeps them unchanged until the call's work has completed (include/cvhip.h)."""
        _lib.check(_lib.lib().cvhip_ctx_set_borrow_inputs(self._h, int(borrow)), "cvhip_ctx_set_borrow_inputs")

    def set_fuse_level_calls(self, enable: bool):
        """cvhip_ctx_set_fuse_level_calls: the four per-pass calls of a level arrive in the reference's order
        (correlate_images(fused=False) issues exactly that) and are executed as one level call (include/cvhip.h)."""
        _lib.check(_lib.lib().cvhip_ctx_set_fuse_level_calls(self._h, int(enable)), "cvhip_ctx_set_fuse_level_calls")

    def set_stats_ahead(self, ahead: bool):
        """cvhip_ctx_set_stats_ahead: the window statistics of borrowed level images run on a side stream, under the
        coarse levels' search (the images must be complete in memory when they are passed)."""
        _lib.check(_lib.lib().cvhip_ctx_set_stats_ahead(self._h, int(ahead)), "cvhip_ctx_set_stats_ahead")

    def set_search_version(self, version: int):
        _lib.check(_lib.lib().cvhip_ctx_set_search_version(self._h, version), "cvhip_ctx_set_search_version")

    def set_async_readback(self, enable: bool):
        """complete() into page-locked HOST arrays returns once the copies are enqueued; they are complete after
        device.synchronize() (include/cvhip.h)."""
        _lib.check(_lib.lib().cvhip_ctx_set_async_readback(self._h, int(enable)), "cvhip_ctx_set_async_readback")

    def set_result_bands(self, bands: int):
        """cvhip_ctx_set_result_bands: the last level in row bands, each copied out to a HOST destination under the search
        of the bands behind it (same result; include/cvhip.h)."""
        _lib.check(_lib.lib().cvhip_ctx_set_result_bands(self._h, bands), "cvhip_ctx_set_result_bands")

    def result_bands(self) -> int:
        """How many bands the grid now held went out in (1 = not banded)."""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().cvhip_ctx_get_result_bands(self._h, C.byref(n)), "cvhip_ctx_get_result_bands")
        return int(n.value)

    def set_exact_scores(self, all_passes: bool):
        """Scores of EVERY pass are the reference's bits (default: only the observable ones - the forward pass at
        scale 1; include/cvhip.h).  Positions are exact either way."""
        _lib.check(_lib.lib().cvhip_ctx_set_exact_scores(self._h, int(all_passes)), "cvhip_ctx_set_exact_scores")

    def set_range_mode(self, mode: int):
        """Test hook of the search-range kernel (include/cvhip.h): 0 default, 1 chain only, 2 / 3 mixed paths."""
        _lib.check(_lib.lib().cvhip_ctx_set_range_mode(self._h, mode), "cvhip_ctx_set_range_mode")

    def set_row_shard_rccl(self, comm):
        """Bands + one RCCL all-gather per sharded search pass, issued by the library itself on the device handle's
        stream (cvhip_ctx_set_row_shard_rccl); comm is a sharding.RcclCommunicator."""
        _lib.check(_lib.lib().cvhip_ctx_set_row_shard_rccl(self._h, comm.handle), "cvhip_ctx_set_row_shard_rccl")

    def gather_bands_rccl(self, comm, root: int = 0):
        """Independent-band mode's single gather (cvhip_ctx_gather_bands_rccl): forward bands -> root (-1: every rank)."""
        _lib.check(_lib.lib().cvhip_ctx_gather_bands_rccl(self._h, comm.handle, root), "cvhip_ctx_gather_bands_rccl")

    def set_row_shard(self, num: int, den: int, gather=None):
        """gather(cells_ptr: int, shard_bytes: int, n_shards: int, direction: int) -> None does the
        in-place all-gather of the level grid (see cybervision_amd.sharding).  Stream ordering (include/cvhip.h): with a
        device created on the caller's stream the hook must enqueue its collective on that stream; with a device that
        owns a private stream the library fences both sides of the hook itself (slower).  set_row_shard_rccl avoids
        the question: the library issues the collective on its own stream."""
        if gather is None:
            cb = _lib.NULL_ALLGATHER
        else:
            def _cb(_user, cells, shard_bytes, n_shards, direction):
                try:
                    gather(int(cells), int(shard_bytes), int(n_shards), int(direction))
                    return 0
                except Exception as exc:  # never let an exception cross the C frame
                    self._gather_error = exc
                    return 1
            cb = _lib.ALLGATHER_FN(_cb)
        self._gather_cb = cb  # keep the thunk alive as long as the context may call it
        _lib.check(_lib.lib().cvhip_ctx_set_row_shard(self._h, num, den, cb, None), "cvhip_ctx_set_row_shard")

    def set_row_band(self, num: int, den: int) -> bool:
        """Independent-band sharding (no collectives until the final gather).  Returns False when the
        geometry is not row-local and the all-gather mode (set_row_shard) has to be used instead."""
        rc = _lib.lib().cvhip_ctx_set_row_band(self._h, num, den)
        if rc == -3:
            return False
        _lib.check(rc, "cvhip_ctx_set_row_band")
        return True

    def level_grid(self, direction: CorrelationDirection):
        """-> cells / scores: device pointers of the level grid's two planes (u32 match words, f32 scores; 4 bytes per
        level pixel each), lw, lh, this shard's rows, rows_per_shard."""
        cells, scores = C.c_void_p(), C.c_void_p()
        lw, lh, r0, r1, rps = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(_lib.lib().cvhip_ctx_level_grid(self._h, int(direction), C.byref(cells), C.byref(scores), C.byref(lw), C.byref(lh),
                                                   C.byref(r0), C.byref(r1), C.byref(rps)), "cvhip_ctx_level_grid")
        return {"cells": cells.value, "scores": scores.value, "lw": lw.value, "lh": lh.value, "row0": r0.value, "row1": r1.value,
                "rows_per_shard": rps.value}

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cvhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def correlate_dense(device: GpuDevice, pyr1, pyr2, fundamental_matrix,
                    projection_mode: ProjectionMode = ProjectionMode.Affine, fused: bool = True):
    """The dense stage of ImageReconstruction::correlate_dense (reconstruction.rs:554-588) over
    prebuilt pyramids: pyr[k] is the 1/2^k image, levels run coarse to fine, then complete()."""
    steps = len(pyr1) - 1

    def dims(img):
        return (int(img.shape[1]), int(img.shape[0]))

    pc = PointCorrelations(device, dims(pyr1[0]), dims(pyr2[0]), fundamental_matrix, projection_mode)
    try:
        for i in range(steps + 1):
            k = steps - i
            pc.correlate_images(pyr1[k], pyr2[k], 1.0 / float(1 << k), fused=fused)
        return pc.complete()
    finally:
        pc.close()


def box_pyramid_device(device: GpuDevice, img, steps: int):
    """[level 0, ..., level `steps`] of 2x2 box-filter levels built on the device (cvhip_downsample_box);
    img is a torch CUDA uint8 tensor, the levels are torch tensors resident in HBM."""
    import torch

    out = [img.contiguous()]
    for _ in range(steps):
        src = out[-1]
        h, w = int(src.shape[0]), int(src.shape[1])
        dst = torch.empty((h // 2, w // 2), dtype=torch.uint8, device=src.device)
        _lib.check(_lib.lib().cvhip_downsample_box(device.handle, C.c_void_p(src.data_ptr()), w, h,
                                                   C.c_void_p(dst.data_ptr())), "cvhip_downsample_box")
        out.append(dst)
    return out


def resize_lanczos3(device: GpuDevice, img, scale: float):
    """SourceImage::resize (reconstruction.rs:146-162): the level image at `scale` with the `image` crate's Lanczos3
    (cvhip_resize_lanczos3; tolerance parity, see include/cvhip.h).  img: numpy uint8 array or torch CUDA uint8 tensor;
    the result lives where the input does.  A device result is written in the order of the DEVICE HANDLE's stream (no host
    synchronisation): with a handle on a stream of its own, `device.synchronize()` before another stream reads it."""
    p, w, h, keep = _ptr_shape(img)
    s = np.float32(scale)
    nw, nh = int(np.float32(w) * s), int(np.float32(h) * s)   # (w as f32 * scale) as u32
    if hasattr(img, "data_ptr"):
        import torch

        out = torch.empty((nh, nw), dtype=torch.uint8, device=img.device)
        po = C.c_void_p(out.data_ptr())
    else:
        out = np.empty((nh, nw), dtype=np.uint8)
        po = C.c_void_p(out.ctypes.data)
    _lib.check(_lib.lib().cvhip_resize_lanczos3(device.handle, p, w, h, po, nw, nh), "cvhip_resize_lanczos3")
    del keep
    return out


def lanczos_pyramid(device: GpuDevice, img, steps: int):
    """[level 0, ..., level `steps`]: every level resized from the FULL-RESOLUTION image, as the reference's level
    loops do (reconstruction.rs:421-422, 567-568), scale = 1 / (1 << k)."""
    pyr = [resize_lanczos3(device, img, 1.0 / float(1 << k)) for k in range(steps + 1)]
    if hasattr(img, "data_ptr"):
        device.synchronize()  # (one wait per pyramid: the levels are complete for whichever stream reads them next)
    return pyr
