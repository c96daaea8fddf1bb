"""The hot-path calls of ImageReconstruction (src/reconstruction.rs) in the order the reference makes them,
over the C ABI: match_keypoints (:400-500: per-level ORB on both images + KeypointMatching),
find_fundamental_matrix (:502-526), correlate_dense (:528-603) and the pair loops of reconstruct /
reconstruct_dense (:261-277, :680-730).  This is BASELINE config 5's workload ("3-image perspective SFM:
ORB + RANSAC F-matrix on GPU + pairwise dense correlation").  Everything between the calls that the
reference does on the CPU (image decode, Lanczos3 resize, triangulation, bundle adjustment) is outside
the boundary; images arrive here as prebuilt pyramids (pyr[k] = the 1/2^k level, host arrays or device
tensors).  No compute in Python - only the calls and their timing.
"""
from __future__ import annotations

import time

import numpy as np

from . import correlation, fundamentalmatrix, orb, pointmatching
from .fundamentalmatrix import ProjectionMode


class ImageReconstruction:
    def __init__(self, device, projection_mode: ProjectionMode = ProjectionMode.Perspective):
        self.device = device
        self.projection_mode = ProjectionMode(projection_mode)
        self.timings_ms: dict[str, float] = {}

    def _timed(self, key: str, fn):
        self.device.synchronize()
        t0 = time.perf_counter()
        out = fn()
        self.device.synchronize()
        self.timings_ms[key] = self.timings_ms.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    # reconstruction.rs:418-458 for ONE image: levels coarse to fine, coordinates mapped back, lists concatenated
    def extract_keypoints(self, pyramid):
        dims = (int(pyramid[0].shape[1]), int(pyramid[0].shape[0]))
        steps = orb.optimal_scale_steps(*dims)
        return self._timed("orb", lambda: orb.extract_points_multiscale(self.device, pyramid[:steps + 1]))

    # the same for every image of the set in ONE batched call (all levels of all images: cvhip_orb_extract_batch)
    def extract_keypoints_set(self, pyramids):
        trimmed = []
        for pyramid in pyramids:
            dims = (int(pyramid[0].shape[1]), int(pyramid[0].shape[0]))
            trimmed.append(pyramid[:orb.optimal_scale_steps(*dims) + 1])
        return self._timed("orb", lambda: orb.extract_points_multiscale_set(self.device, trimmed))

    # reconstruction.rs:400-500
    def match_keypoints(self, keypoints1, keypoints2):
        thr = (pointmatching.THRESHOLD_PERSPECTIVE if self.projection_mode == ProjectionMode.Perspective
               else pointmatching.THRESHOLD_AFFINE)
        m, _ = self._timed("match", lambda: pointmatching.match_points(self.device, keypoints1[0], keypoints1[1],
                                                                       keypoints2[0], keypoints2[1], thr))
        return m

    # reconstruction.rs:502-526
    def find_fundamental_matrix(self, img1_dimensions, img2_dimensions, point_matches, seed: int = 0, progress_listener=None):
        """progress_listener: the reference ALWAYS passes one (`Some(&pb)`, reconstruction.rs:510-518): an object with
        report_status(pos) / report_matches(count), both thunks of cvhip_find_ransac."""
        max_dimension = float(max(img1_dimensions[0], img1_dimensions[1], img2_dimensions[0], img2_dimensions[1]))
        fm = fundamentalmatrix.FundamentalMatrix(self.projection_mode, max_dimension)
        return self._timed("ransac", lambda: fm.find_ransac(self.device, point_matches, seed=seed, progress_listener=progress_listener))

    # reconstruction.rs:528-603 (up to complete(); triangulation is the next stage)
    def correlate_dense(self, pyr1, pyr2, f, out_xy=None, out_corr=None, borrow=False):
        """borrow: the pyramids are device tensors with 64 readable bytes behind every level and complete in memory
        (padded_pyramid below) - the library then uses them in place and computes their window statistics ahead of the
        levels' turn (cvhip_ctx_set_borrow_inputs / cvhip_ctx_set_stats_ahead)."""
        def dims(img):
            return (int(img.shape[1]), int(img.shape[0]))

        steps = correlation.optimal_scale_steps(*dims(pyr1[0]))
        mode = correlation.ProjectionMode(int(self.projection_mode))

        if out_xy is None and hasattr(pyr1[0], "data_ptr"):
            # device-resident pyramids: the grid stays in HBM too (the next stage - triangulation, track extension -
            # reads it there); a host copy is the caller's explicit choice (pass numpy outputs)
            import torch

            w, h = dims(pyr1[0])
            out_xy = torch.empty((h, w, 2), dtype=torch.int32, device=pyr1[0].device)
            out_corr = torch.empty((h, w), dtype=torch.float32, device=pyr1[0].device)

        def run():
            pc = correlation.PointCorrelations(self.device, dims(pyr1[0]), dims(pyr2[0]), f, mode)
            try:
                if borrow:
                    pc.set_borrow_inputs(True)
                    pc.set_stats_ahead(True)
                for i in range(steps + 1):
                    k = steps - i
                    pc.correlate_images(pyr1[k], pyr2[k], 1.0 / float(1 << k))
                return pc.complete(out_xy=out_xy, out_corr=out_corr)
            finally:
                pc.close()

        return self._timed("dense", run)


    def correlate_dense_set(self, jobs, borrow=False):
        """jobs: [(pyr1, pyr2, f)] - the pairs of reconstruct_dense's loop (reconstruction.rs:680-730), each a correlate_dense of
        its own -> [(xy, corr)].  The pairs are independent: with device-resident pyramids (borrow) every pair runs on a device
        handle - a stream - of its own, so that one pair's coarse levels (chains of small dependent launches that leave the
        chip idle) run under another pair's full-resolution levels.  Results are those of the calls one after the other."""
        if not borrow or len(jobs) < 2:
            return [self.correlate_dense(p1, p2, f, borrow=borrow) for p1, p2, f in jobs]
        import torch

        mode = correlation.ProjectionMode(int(self.projection_mode))
        devs = [self.device] + [self.device.side_handle(i) for i in range(len(jobs) - 1)]

        def dims(img):
            return (int(img.shape[1]), int(img.shape[0]))

        outs = []
        for p1, _, _ in jobs:
            w, h = dims(p1[0])
            outs.append((torch.empty((h, w, 2), dtype=torch.int32, device=p1[0].device),
                         torch.empty((h, w), dtype=torch.float32, device=p1[0].device)))
        torch.cuda.synchronize(jobs[0][0][0].device)  # (the outputs exist before a private stream writes them)

        def run():
            pcs = []
            try:
                for dev, (p1, p2, f) in zip(devs, jobs):
                    pc = correlation.PointCorrelations(dev, dims(p1[0]), dims(p2[0]), f, mode)
                    pcs.append(pc)
                    pc.set_borrow_inputs(True)
                    pc.set_stats_ahead(dev is self.device)  # (one statistics side stream: every further stream of the process
                    #                                          shares a hardware queue with one already in use)
                res = []
                for pc, (p1, p2, f), (oxy, ocorr) in zip(pcs, jobs, outs):
                    steps = correlation.optimal_scale_steps(*dims(p1[0]))
                    for i in range(steps + 1):
                        k = steps - i
                        pc.correlate_images(p1[k], p2[k], 1.0 / float(1 << k))
                    res.append(pc.complete(out_xy=oxy, out_corr=ocorr))
                for dev in devs:
                    dev.synchronize()
                return res
            finally:
                for pc in pcs:
                    pc.close()

        return self._timed("dense", run)


def padded_pyramid(pyr):
    """Device copies of a pyramid's levels with 64 readable bytes behind each (what cvhip_ctx_set_borrow_inputs asks
    for); host pyramids are returned as they are.  -> (pyramid, borrowable)."""
    if not hasattr(pyr[0], "data_ptr") or not pyr[0].is_cuda:
        return pyr, False
    import torch

    out = []
    for level in pyr:
        n = level.numel()
        buf = torch.zeros(n + 64, dtype=torch.uint8, device=level.device)
        buf[:n].copy_(level.reshape(-1))
        out.append(buf[:n].view(level.shape[0], level.shape[1]))
    torch.cuda.synchronize(pyr[0].device)  # complete in memory before any level is handed to the library
    return out, True


class ProgressBar:
    """Stand-in for the reference's indicatif bar (reconstruction.rs:840-863): both RANSAC thunks, nothing drawn."""

    def __init__(self):
        self.pos, self.matches, self.calls = 0.0, 0, 0

    def report_status(self, pos):
        self.pos = pos
        self.calls += 1

    def report_matches(self, count):
        self.matches = count
        self.calls += 1


def reconstruct_pairs(device, pyramids, projection_mode: ProjectionMode = ProjectionMode.Perspective, seed: int = 0,
                      dense: bool = True, borrow: bool = False, listener: bool = False, images=None, concurrent_pairs: bool = True):
    """The two pair loops of `reconstruct` (reconstruction.rs:261-277 sparse, :680-730 dense) over n images:
    for every i < j the sparse stage (ORB on both, matcher, RANSAC); then, for every pair that produced an F, the
    dense correlation.  The reference re-extracts an image's keypoints for every pair it takes part in; the result
    is a pure function of the image, so they are extracted once per image here.  concurrent_pairs (device-resident pyramids):
    the pairs' dense correlations - independent of each other - run side by side, each on a stream of its own
    (ImageReconstruction.correlate_dense_set); False: one after the other, as the reference's loop does.
    -> dict: keypoints [n], pairs {(i, j): {matches, f, inliers, (xy, corr)}}, timings_ms per stage."""
    rec = ImageReconstruction(device, projection_mode)
    if images is not None:
        # The reference's own pyramids: SourceImage::resize (Lanczos3, reconstruction.rs:146-162) per level, rebuilt by every
        # stage that needs them - match_keypoints for both images of a pair (:421-422; once per image here, like the
        # keypoints) and correlate_dense for both images of a pair (:567-568).  `pyramids` is ignored; the resizes run on the
        # device (cvhip_resize_lanczos3) and are timed as their own stage.
        pyramids = [rec._timed("resize", lambda im=im: correlation.lanczos_pyramid(
            device, im, orb.optimal_scale_steps(int(im.shape[1]), int(im.shape[0])))) for im in images]
    n = len(pyramids)
    keypoints = rec.extract_keypoints_set(pyramids)
    pairs = {}
    for i in range(n - 1):
        for j in range(i + 1, n):
            di = (int(pyramids[i][0].shape[1]), int(pyramids[i][0].shape[0]))
            dj = (int(pyramids[j][0].shape[1]), int(pyramids[j][0].shape[0]))
            matches = rec.match_keypoints(keypoints[i], keypoints[j])
            entry = {"matches": matches, "f": None, "inliers": None, "error": None}
            try:
                f, inliers, _ = rec.find_fundamental_matrix(di, dj, matches, seed=seed + 1000 * i + j,
                                                            progress_listener=ProgressBar() if listener else None)
                entry["f"], entry["inliers"] = f, inliers
            except Exception as exc:  # "Failed to match images" (reconstruction.rs:268-274): the pair is skipped
                entry["error"] = str(exc)
            pairs[(i, j)] = entry
    if dense and borrow and images is None and concurrent_pairs:
        todo = [(key, entry) for key, entry in pairs.items() if entry["f"] is not None]
        results = rec.correlate_dense_set([(pyramids[i], pyramids[j], entry["f"]) for (i, j), entry in todo], borrow=True)
        for (_, entry), (xy, corr) in zip(todo, results):
            entry["xy"], entry["corr"] = xy, corr
    elif dense:
        for (i, j), entry in pairs.items():
            if entry["f"] is None:
                continue
            pi, pj = pyramids[i], pyramids[j]
            if images is not None:
                dsteps = correlation.optimal_scale_steps(int(images[i].shape[1]), int(images[i].shape[0]))
                pi, pj = rec._timed("resize", lambda: (correlation.lanczos_pyramid(device, images[i], dsteps),
                                                       correlation.lanczos_pyramid(device, images[j], dsteps)))
            entry["xy"], entry["corr"] = rec.correlate_dense(pi, pj, entry["f"], borrow=borrow and images is None)
    return {"keypoints": keypoints, "pairs": pairs, "timings_ms": dict(rec.timings_ms)}


def match_count(xy) -> int:
    """Number of Some cells of a dense grid (host array or device tensor)."""
    if hasattr(xy, "data_ptr"):
        return int((xy[..., 0] >= 0).sum().item())
    return int((np.asarray(xy)[..., 0] >= 0).sum())
