/*
 * cvhip.h — C ABI of libcvhip.so, the MI355X (gfx950) backend for cybervision's hot path.
 *
 * This is the drop-in boundary: every entry point names the reference interface it replaces
 * (file:line into zlogic/cybervision v0.20.3, src/...).  Plain C types only; no HIP or torch
 * types cross the boundary.  INTEGRATION.md shows the Rust `extern "C"` block and the
 * `hip.rs` backend module a maintainer would add next to gpu/vulkan.rs and gpu/metal.rs.
 *
 * Conventions
 *  - Return value: 0 = CVHIP_OK, negative = error class; the message for the calling thread's
 *    last failure is cvhip_last_error() (maps to GpuError::Internal(&'static str),
 *    correlation/gpu/vulkan.rs:1204-1272).  Nothing throws or aborts across the ABI.
 *  - Ownership: the caller owns every host pointer and it only has to stay valid for the
 *    duration of the call; the library owns all device memory behind the opaque handles.
 *  - Image and output pointers may be HOST or DEVICE (HIP) pointers; the library detects
 *    which (hipPointerGetAttributes).  Device pointers must belong to the handle's GPU.
 *  - Grids are row-major, index = width*y + x (data.rs:22-64).  Option<Match> is encoded as
 *    int32 (x, y) with (-1, -1) = None plus a separate float score plane (NaN for None).
 *  - Threading: a handle is used by one thread at a time, as in the reference where every
 *    call goes through `&mut` (correlation/mod.rs:255).  Different handles are independent.
 *  - dir: 0 = CorrelationDirection::Forward, 1 = Reverse (correlation/mod.rs:77-81).  As in
 *    the reference, the caller swaps the images for Reverse (mod.rs:231-237) and the library
 *    transposes F (mod.rs:268-271).
 *  - projection: 0 = ProjectionMode::Affine, 1 = Perspective (correlation/mod.rs:43-47).
 */
#ifndef CVHIP_H
#define CVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVHIP_OK 0
#define CVHIP_ERR_INVALID (-1)     /* bad argument / call sequence */
#define CVHIP_ERR_DEVICE (-2)      /* HIP runtime error (message has hipGetErrorString) */
#define CVHIP_ERR_UNSUPPORTED (-3) /* valid in the reference, not supported here (documented) */
#define CVHIP_ERR_NOMEM (-4)
#define CVHIP_ERR_NO_MODEL (-5)    /* RANSAC: "Not enough matches" / "No reliable matches found" (RansacError) */

typedef struct cvhip_device cvhip_device;
typedef struct cvhip_ctx cvhip_ctx;

/* Progress hook: replaces ProgressListener::report_status (correlation/mod.rs:56-61).
 * Invoked synchronously on the calling thread between kernel submissions; never retained. */
typedef void (*cvhip_progress_fn)(void *user, float pos);
/* fundamentalmatrix.rs:41-47: the RANSAC listener's second method, ProgressListener::report_matches(matches_count) -
 * the largest inlier count seen so far.  Same rules as cvhip_progress_fn. */
typedef void (*cvhip_matches_fn)(void *user, uint64_t matches_count);

/* Message of the calling thread's last error ("" if none). Static lifetime per thread. */
const char *cvhip_last_error(void);
/* ABI version, bumped on incompatible change. */
uint32_t cvhip_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Device — replaces GpuDevice::new / create_gpu_context (correlation/mod.rs:145-147) and
 * the DeviceContext trait (correlation/gpu/mod.rs:64-81).
 * ---------------------------------------------------------------------------------------- */

/* low_power mirrors HardwareMode::GpuLowPower (mod.rs:50-54); accepted and ignored (no
 * dispatch segmenting is needed on MI355X).  ordinal < 0 = HIP's current device. */
int cvhip_device_create(int low_power, int ordinal, cvhip_device **out);
/* Same, but all work is submitted to the caller's HIP stream (a hipStream_t passed as void*,
 * e.g. torch.cuda.current_stream().cuda_stream) so it is ordered with the caller's own work
 * (collectives between passes when row-sharding).  NULL means HIP's default (null) stream.  The
 * stream is not owned by the library. */
int cvhip_device_create_on_stream(int low_power, int ordinal, void *hip_stream, cvhip_device **out);
void cvhip_device_destroy(cvhip_device *dev);
/* DeviceContext::get_device_name (gpu/mod.rs:70). Valid until cvhip_device_destroy. */
const char *cvhip_device_name(const cvhip_device *dev);
/* Block until everything submitted on the device's stream has finished. */
int cvhip_device_synchronize(cvhip_device *dev);

/* ------------------------------------------------------------------------------------------
 * Dense correlation context — replaces GpuContext (correlation/gpu/mod.rs:106-362) as used by
 * PointCorrelations (correlation/mod.rs:150-245).
 * ---------------------------------------------------------------------------------------- */

/* GpuContext::new (gpu/mod.rs:125-163; called mod.rs:159-165).  dims are the FULL-RES image
 * dimensions, F is row-major 3x3 (nalgebra Matrix3 (r,c) -> F[3*r+c]). */
int cvhip_ctx_create(cvhip_device *dev, uint32_t w1, uint32_t h1, uint32_t w2, uint32_t h2, int projection,
                     const double *F, cvhip_ctx **out);
void cvhip_ctx_destroy(cvhip_ctx *ctx);

/* GpuContext::correlate_images (gpu/mod.rs:218-362; called mod.rs:255-259): one search pass
 * over level images img1 (searched) and img2 (target) at `scale`, results replace this
 * direction's level grid.  Results follow the reference's --mode=cpu semantics
 * (mod.rs:247-540), NOT the GLSL shaders' (SURVEY.md §8a N1).
 * Supported schedule (the only one the reference issues, reconstruction.rs:565-568):
 * scale = 2^-k exactly, k strictly decreasing from call to call per direction, level dims =
 * floor(full * scale); anything else returns CVHIP_ERR_UNSUPPORTED. */
int cvhip_correlate_images(cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2,
                           uint32_t w2, uint32_t h2, float scale, int first_pass, int dir,
                           cvhip_progress_fn progress, void *user);

/* GpuContext::cross_check_filter (gpu/mod.rs:172-208; called mod.rs:557-560) with the CPU
 * semantics of mod.rs:552-624. */
int cvhip_cross_check_filter(cvhip_ctx *ctx, float scale, int dir);

/* One whole PointCorrelations::correlate_images (mod.rs:217-245) in a single call: forward
 * pass, reverse pass (images swapped, F transposed), cross-check forward, cross-check reverse.
 * Same results as the four calls above; uploads and window statistics are shared. */
int cvhip_correlate_level(cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2,
                          uint32_t w2, uint32_t h2, float scale, int first_pass, cvhip_progress_fn progress,
                          void *user);

/* Result bands for host destinations.  With bands > 1 the last level (scale 1) of a pair whose geometry is row-local
 * (the same test as cvhip_plan_bands: affine models, near-horizontal epipolar lines) is searched and cross-checked in
 * `bands` row bands, and cvhip_complete_dir(dir 0) into HOST memory expands and copies band b out on the copy stream
 * while the bands behind it are still being searched - the 201 MB grid of a 4096^2 pair then costs ~1.5 ms beyond the
 * search instead of ~4.5.  The result is the one of bands == 1, bit for bit (tests/test_corr_gpu.py).  In the fused
 * four-call mode the last level's launches wait for its two cross_check_filter calls.  Under cvhip_ctx_set_async_readback
 * the level is not banded (the whole transfer runs under the next pair's search there).  Default 1 (off); at most 16;
 * 0 = the library chooses by size (bands of at least half a megapixel, six at most - what a binding whose grid always goes
 * to the host should pass).
 * Replaces nothing in the reference (its complete() maps one buffer after the last submission, gpu/mod.rs:321-349). */
int cvhip_ctx_set_result_bands(cvhip_ctx *ctx, uint32_t bands);
/* How many bands the grid now held went out in (1: not banded - the geometry, the size or the call order ruled it out). */
int cvhip_ctx_get_result_bands(cvhip_ctx *ctx, uint32_t *live);

/* GpuContext::complete_process (gpu/mod.rs:210-216; called mod.rs:208-215): write the forward
 * full-resolution grid.  out_xy: 2*w1*h1 int32, out_corr: w1*h1 float (may be NULL).
 * Host destinations are complete on return (synchronises); DEVICE destinations are written in
 * stream order on the context's stream without a host synchronisation (cvhip_device_synchronize,
 * or any later work on that stream, orders after them).  The context can be destroyed or reused
 * for cvhip_complete_dir afterwards. */
int cvhip_complete(cvhip_ctx *ctx, int32_t *out_xy, float *out_corr);
/* Asynchronous readback for pipelines that correlate pair after pair (reconstruction.rs:680-730): with enable = 1,
 * cvhip_complete into HOST destinations returns once the copies are enqueued - on the handle's own copy stream, behind
 * the grid expansion, from one of two device staging sets - so that the 12 B/px transfer of this pair runs under the
 * search of the next.  The destinations must be page-locked (hipHostMalloc / hipHostRegister; a pageable destination
 * makes the copy synchronous again) and are complete after cvhip_device_synchronize; at most two readbacks are in
 * flight per device handle (a third waits, on the device, for the first).  Default 0: complete on return. */
int cvhip_ctx_set_async_readback(cvhip_ctx *ctx, int enable);
/* Same for either direction (dir 1 = correlated_points_reverse, mod.rs:65); test hook. */
int cvhip_complete_dir(cvhip_ctx *ctx, int dir, int32_t *out_xy, float *out_corr);
/* The same grid in 8 instead of 12 bytes per cell, for host destinations behind PCIe: out_cells[y * w + x] =
 * y2 << 16 | x2 of the match (level dimensions are at most 65535), 0xFFFFFFFF = None; out_corr as above (may be NULL).
 * The binding unpacks in the loop in which it builds its Grid<Option<Match>> anyway (gpu/mod.rs:321-349 reads its own
 * [i32; 2] + f32 buffers cell by cell).  Everything else - result bands, asynchronous readback, device destinations - as
 * cvhip_complete_dir. */
int cvhip_complete_packed(cvhip_ctx *ctx, int dir, uint32_t *out_cells, float *out_corr);

/* All-gather hook for row sharding: called by cvhip_correlate_level after each sharded search
 * pass, on the calling thread.  `cells` is the device pointer of the level grid's match plane (4 bytes per level pixel)
 * of direction `dir`; shard r (0 <= r < n_shards) owns bytes [r*shard_bytes, (r+1)*shard_bytes).  The hook
 * must all-gather in place (e.g. ncclAllGather / torch.distributed.all_gather_into_tensor on
 * the stream given to cvhip_device_create_on_stream) and return 0, or non-zero to abort.  dir = 0 / 1: the match plane
 * of that direction (after each sharded search pass); dir = 2: the forward SCORE plane, once, after the forward pass at
 * scale 1 (same geometry; the only scores cvhip_complete reports).
 * Stream ordering: with a device handle created by cvhip_device_create_on_stream the hook MUST enqueue its collective
 * on that stream (the search pass before it and the cross-checks after it are submitted there).  With a handle that
 * owns a private stream (cvhip_device_create) the library fences both sides of the hook itself (stream synchronise
 * before, device synchronise after) - correct, but slower.  cvhip_ctx_set_row_shard_rccl needs neither. */
typedef int (*cvhip_allgather_fn)(void *user, void *cells, uint64_t shard_bytes, uint32_t n_shards, int dir);

/* Dense consumer — replaces AffineTriangulation::triangulate + triangulate_point
 * (triangulation.rs:268-330; reached from Triangulation::triangulate, :181-201, right after
 * PointCorrelations::complete()).  Straight from the device-resident forward grid, one track per
 * Some cell in scan order: out_points3d gets (x, y, sqrt(dx^2 + dy^2)) as 3 doubles per track,
 * out_p2 (may be NULL) the matched point (2 u32).  At most `cap` tracks are written; *out_n is
 * the number of Some cells (call with cap = 0 to size the buffers).  Host or device pointers. */
int cvhip_triangulate_affine(cvhip_ctx *ctx, double *out_points3d, uint32_t *out_p2, uint64_t cap, uint64_t *out_n);

/* Dense consumer, perspective pipeline — replaces Triangulation::extend_tracks (triangulation.rs:1330-1419; called
 * right after a pair's dense correlation, :638, :697) on the device-resident forward grid.
 *  - track_p1: the existing tracks' points in image 1, 2 int32 per track, (-1,-1) where a track has none.
 *  - out_track_p2[i]: the image-2 point `track.add(image2_index, ..)` is called with for track i - the match of the
 *    nearest Some cell within [p - r, p + r) (squared distance, first minimum in row-major order), r =
 *    3 * max_dimension2 / 1000 (3 when max_dimension2 <= 1000; max_dimension2 = max(width, height) of image 2) - or
 *    (-1,-1).  The caller applies Track::add's "only if the track has no point for this image yet" (:370-375).
 *  - The merged points are then cleared from the remaining grid AT THEIR OWN coordinates (the reference indexes the
 *    image-1 grid with the image-2 point, :1391-1393; out of bounds it panics - here CVHIP_ERR_INVALID), and every
 *    remaining Some cell becomes a new track: out_new_p1 = the cell, out_new_p2 = its match, scan order, at most
 *    `cap` written; *out_n_new = their number (call with cap = 0 to size the buffers).
 * Integer only, bit-exact.  Host or device pointers.  Uses the context's per-pass scratch: call it between pairs. */
int cvhip_extend_tracks(cvhip_ctx *ctx, const int32_t *track_p1, uint64_t n_tracks, uint32_t max_dimension2,
                        int32_t *out_track_p2, uint32_t *out_new_p1, uint32_t *out_new_p2, uint64_t cap,
                        uint64_t *out_n_new);

/* Row sharding (multi-GPU): restrict the SEARCH passes of this context to shard `num` of `den`
 * equal row chunks of the searched level image: rows [num*rps, min((num+1)*rps, h_level)) with
 * rps = ceil(h_level / den).  Rows outside the band keep whatever the level grid holds until
 * the bands are all-gathered; the cross-checks then run on the whole grid on every rank, so
 * every rank ends with the complete, identical result (no reduction across ranks: N-GPU output
 * is bit-identical to 1-GPU output).  cvhip_correlate_level calls `gather` itself (levels with
 * fewer than 64 rows per shard are simply computed whole on every rank); with the per-pass
 * calls the host gathers between passes (cvhip_ctx_level_grid).  num = 0, den = 1 (default) =
 * all rows; den <= 64; gather may be NULL when only the per-pass calls are used. */
int cvhip_ctx_set_row_shard(cvhip_ctx *ctx, uint32_t num, uint32_t den, cvhip_allgather_fn gather, void *user);
/* Row sharding without collectives ("independent band" mode).  For row-local geometry — an affine F
 * whose epipolar lines stay within a few rows of the pixel's own row at every pyramid level — shard
 * `num` of `den` computes, at every level, only the rows its band of the FINAL forward grid depends
 * on: the band plus a halo (about 2*D + 20 rows per level, D = the corridor's row excursion) that is
 * recomputed redundantly instead of exchanged.  Uses the fact that the two cross-checks of a level
 * commute (any reverse match that supports a forward match is itself supported by it, mod.rs:588-624),
 * so no filtered grid has to be communicated between passes.  After the last level only rows
 * [num*rps, (num+1)*rps) of the forward level grid (cvhip_ctx_level_grid, dir 0) are meaningful; the
 * host gathers those bands once (the single RCCL gather of the north star) before cvhip_complete.
 * Returns CVHIP_ERR_UNSUPPORTED when the geometry is not row-local (perspective F, steep or
 * column-major lines): fall back to cvhip_ctx_set_row_shard + all-gather hook.  The context must be
 * driven with cvhip_correlate_level over the reference's schedule (scale = 2^-k, k = steps..0). */
int cvhip_ctx_set_row_band(cvhip_ctx *ctx, uint32_t num, uint32_t den);
/* ------------------------------------------------------------------------------------------
 * The collectives of row sharding on RCCL, inside the library (no torch, no host hook): one process per GPU, one
 * communicator per device handle, every collective enqueued on the handle's own stream - ordered with the kernels
 * around it by construction.  librccl.so.1 is opened on first use (CVHIP_ERR_UNSUPPORTED if it cannot be).
 * ---------------------------------------------------------------------------------------- */
typedef struct cvhip_rccl cvhip_rccl;
#define CVHIP_RCCL_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls this and hands the 128 bytes to every rank (launcher's store, MPI, a file). */
int cvhip_rccl_unique_id(uint8_t *id);
/* ncclCommInitRank on the handle's GPU; collective over all `world` ranks (<= 64). */
int cvhip_rccl_create(cvhip_device *dev, const uint8_t *id, uint32_t rank, uint32_t world, cvhip_rccl **out);
void cvhip_rccl_destroy(cvhip_rccl *comm);
/* In-place all-gather of `world` chunks of shard_bytes at `buf` (device memory; this rank's chunk already in place). */
int cvhip_rccl_allgather(cvhip_rccl *comm, void *buf, uint64_t shard_bytes);
/* In-place gather to `root` only: every other rank sends its chunk straight into the root's buffer (grouped
 * ncclSend/ncclRecv - on an 8-GPU xGMI node seven concurrent point-to-point transfers, not a ring). */
int cvhip_rccl_gather(cvhip_rccl *comm, void *buf, uint64_t shard_bytes, uint32_t root);
/* cvhip_ctx_set_row_shard(ctx, rank, world, <all-gather above>): bands + one all-gather per sharded search pass. */
int cvhip_ctx_set_row_shard_rccl(cvhip_ctx *ctx, cvhip_rccl *comm);
/* The single gather of independent-band mode (cvhip_ctx_set_row_band(ctx, rank, world)): after the last level, the
 * forward level grid's bands go to `root` (root < 0: to every rank) before cvhip_complete. */
int cvhip_ctx_gather_bands_rccl(cvhip_ctx *ctx, cvhip_rccl *comm, int root);

/* Device pointers + geometry of direction `dir`'s current level grid, for the host's collectives.  Two planes of one
 * 4-byte word per level pixel, row-major lw x lh: `cells` = the match plane, u32 x | y << 16 in LEVEL coordinates
 * (0xFFFFFFFF = None) - everything the search range and the cross-checks of other ranks read, i.e. all that has to
 * travel between passes; `scores` (may be NULL) = the f32 score plane, which only cvhip_complete reads (the single
 * final gather moves both).  Each plane always has room for den * rows_per_shard rows, so bands can be gathered in
 * equal chunks. */
int cvhip_ctx_level_grid(cvhip_ctx *ctx, int dir, void **cells, void **scores, uint32_t *lw, uint32_t *lh, uint32_t *row0,
                         uint32_t *row1, uint32_t *rows_per_shard);

/* Use device-resident level images where they are instead of copying them into the context's own padded
 * buffers first.  By enabling this the caller guarantees, for every DEVICE pointer it passes as a level image:
 * at least 64 readable bytes after the last pixel (the kernels read whole dwords at row ends), and that the
 * image stays unchanged until the work of the call has completed on the device's stream.  Under
 * cvhip_ctx_set_fuse_level_calls the work of a level's correlate calls is enqueued by LATER calls - both directions in
 * one launch at the reverse call, and with result bands (cvhip_ctx_set_result_bands != 1) the whole last level at its
 * second cvhip_cross_check_filter call - so borrowed images of a level must then stay unchanged until that level's
 * second cvhip_cross_check_filter call (or cvhip_complete) has completed on the stream.  Host images are still
 * copied.  Off by default. */
int cvhip_ctx_set_borrow_inputs(cvhip_ctx *ctx, int borrow);
/* The reference issues FOUR backend calls per pyramid level (PointCorrelations::correlate_images, correlation/mod.rs:
 * 217-245): correlate_images forward, correlate_images reverse with the SAME two images exchanged, cross_check_filter
 * forward, cross_check_filter reverse.  By enabling this the caller promises that order - the binding of
 * GpuContext::new does - and the library executes the four calls as cvhip_correlate_level would: the
 * forward call takes the images in (both images' window statistics, once per level), the reverse call - recognised by
 * the exchanged image pointers, dimensions, scale and first_pass - launches both search passes together, the second
 * cross-check call launches both filters.  Results are those of the independent calls, bit for bit; a call sequence
 * that departs from the order is executed call by call as without the promise (whatever was taken in runs first), and
 * every other entry point of the context first runs what is pending.  What the promise adds to the contract: between a
 * level's forward and reverse call the two images must not change (the reverse call does not read them again), and an
 * error of the forward pass is reported by the call that executes it.  Off by default. */
int cvhip_ctx_set_fuse_level_calls(cvhip_ctx *ctx, int enable);
/* Window statistics of a level (compute_image_point_data, mod.rs:632-694) on a stream of the library's own, ahead of
 * the level's turn: they depend on the level's images only, and the coarse levels' search is a chain of small
 * dependent launches that leaves the chip idle for ~0.4 ms of a 4096^2 pair - room for the 0.5 ms of full-chip work
 * the statistics of the two finest levels are.  Applies to cvhip_correlate_level calls with BORROWED device images
 * (cvhip_ctx_set_borrow_inputs) outside row-shard mode and outside per-kernel timing; the caller additionally
 * promises that a level image is complete in memory when it is passed (not merely ordered on the handle's stream:
 * the statistics kernel reads it on another stream, ordered behind the work enqueued before the pyramid run's FIRST
 * level only).  Same results bit for bit.  Off by default. */
int cvhip_ctx_set_stats_ahead(cvhip_ctx *ctx, int ahead);

/* Measurement hooks (bench.py).  time_kernels: 1 = every kernel launch is bracketed by HIP events on the
 * stream it is launched on, 2 = only the launches of the search class ([2] below), 0 = off.  Timing is
 * not free: with any timing event in flight the runtime profiles every dispatch of the step (~4 us per
 * kernel, 0.45 ms per 4096^2 step on this stack), however few events are recorded.  count_candidates:
 * evaluated candidates are counted on the device. */
int cvhip_ctx_set_profiling(cvhip_ctx *ctx, int time_kernels, int count_candidates);
/* Accumulated since the last reset: search-kernel launches, their summed duration (ms) and the
 * number of candidates that executed the 121-term sum (mod.rs:442-454).  Synchronises. */
int cvhip_ctx_get_profile(cvhip_ctx *ctx, uint32_t *launches, double *search_ms, uint64_t *candidates,
                          int reset);

/* Per-kernel-class device time since the last reset (needs time_kernels = 1), measured with HIP
 * events on the context's stream: [0] window statistics, [1] search-range estimation, [2] search
 * (version 3: the box filter; 1: the whole search), [3] fallback kernels (whole-corridor
 * exact re-evaluation; for version 3 also the candidate filter on the workgroups the box filter
 * declined), [4] cross-check, [5] grid expansion in complete(), [6] the candidate filter launched as a level's search
 * (the first pass, steep geometry, search version 2).  Synchronises. */
int cvhip_ctx_get_kernel_times(cvhip_ctx *ctx, double ms[7], uint32_t launches[7], int reset);
/* Device counters of the search kernel since the last reset (needs count_candidates = 1):
 * out[0] candidates that passed the reference's bounds/stdev tests (== candidates above),
 * out[1] exact 121-term f32 evaluations, out[2] pixels whose filter band held 2..4 contenders,
 * out[3] pixels that re-evaluated their whole corridor exactly (tile too large for LDS, or more
 * than 4 contenders).  Synchronises. */
int cvhip_ctx_get_counters(cvhip_ctx *ctx, uint64_t out[4], int reset);
/* Select the search kernel: 1 = every candidate through the exact serial f32 chain, 2 = exact-integer
 * filter per candidate + exact re-evaluation of the contenders, 3 (default) = the same filter evaluated
 * as displacement-plane box sums for whole row segments where the epipolar lines keep one major axis, with 2 for
 * every other geometry and as the per-workgroup fallback; 4 = 3 with the box kernel launched for every geometry
 * (testing); 5 = 3 with the rectified affine launches (exactly axis-parallel row-major lines, five stripes) as int8
 * matrix products on the matrix pipe (search4_mfma_kernel: the formulation the north star names; bit-exact, and
 * measured SLOWER than the box sums on MI355X - DESIGN.md section 4.7 - so it is not the default); 6 = 3 with the
 * rectified affine launches on TWO image columns per lane (search3_box2_kernel: one prefix sum and two permutes for two
 * pixels; bit-exact, 0.7-0.77 of the one-column group per pixel in isolation and SLOWER as a kernel - its 116-pixel waves walk
 * longer unions of displacement ranges - DESIGN.md section 4.2).  All give identical results. */
int cvhip_ctx_set_search_version(cvhip_ctx *ctx, int version);
/* Which passes write the reference's SCORES.  Match positions are the reference's in every pass, always.  Scores are
 * only observable for the forward grid of the last (full-resolution) level: the reference overwrites every cell a
 * coarser level wrote, None included (mod.rs:311-316), and complete() drops the reverse grid (mod.rs:208-215); nothing
 * else ever reads a score (estimate_search_range and cross_check_point look at positions only, mod.rs:468-540,
 * 588-624).  By default (all_passes = 0) the forward pass at scale 1 evaluates every recorded contender with the
 * reference's serial f32 chain, as before, and every other pass does so only where it decides something (several
 * contenders inside the filter's band, or one within the band of the threshold); a cell settled without it holds the
 * filter's estimate internally, its score plane is not written, and cvhip_complete_dir reports NaN scores for such a
 * grid.  all_passes = 1: the reference's bits in every cell of every pass - for tests that read coarser levels or the
 * reverse grid through cvhip_complete_dir. */
int cvhip_ctx_set_exact_scores(cvhip_ctx *ctx, int all_passes);
/* Test hook of the search-range kernel (estimate_search_range, mod.rs:468-540): 0 (default) = integer box sums with the
 * reference's f64 chain only where the rounding of `len` is open, 1 = the chain for every pixel, 2 / 3 = every third
 * block of a box-sum tile through the staged / the global-memory chain.  All give identical results. */
int cvhip_ctx_set_range_mode(cvhip_ctx *ctx, int mode);

/* ------------------------------------------------------------------------------------------
 * Pyramid level on the device.  The reference builds every level with
 * the `image` crate's Lanczos3 resize on the host (reconstruction.rs:146-162), upstream of the
 * boundary; its source is not available here, so this is the documented substitute used by the
 * benchmark: one 2x2 box-filter step with rounding, dst dims floor(w/2) x floor(h/2),
 * dst[y][x] = (s[2y][2x] + s[2y][2x+1] + s[2y+1][2x] + s[2y+1][2x+1] + 2) >> 2.  Keeps pyramids
 * resident in HBM for both the ORB and the dense stage.  Host or device pointers.
 * ---------------------------------------------------------------------------------------- */
int cvhip_downsample_box(cvhip_device *dev, const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst);

/* SourceImage::resize (reconstruction.rs:146-162) on the device: image::imageops::resize(.., FilterType::Lanczos3) of a
 * Luma8 image to nw x nh (the caller passes (w as f32 * scale) as u32, (h as f32 * scale) as u32, :149-150).  The
 * `image` crate (0.25.10) is not vendored with the reference; this is its published separable algorithm (vertical pass
 * to f32, horizontal pass, f32 weights normalised per output sample, clamp + round to nearest at the end; equal
 * dimensions are a copy).  Weights come from glibc's sinf on the host (what Rust's f32::sin resolves to on linux-gnu);
 * every other step is a single IEEE f32 operation in the crate's order, so the bytes equal the oracle's restatement
 * (oracle/cvref_resize.py, same sinf) exactly - tested with MAX_DIFF = 0.  Nothing pins either to the crate itself
 * ("parity unpinned").  Host or device pointers: with a host source or destination the call is complete on return; with
 * both on the device the two kernels are enqueued on the handle's stream and the call returns (stream order, like
 * cvhip_complete into device memory).  The resampling tables of every (source size, output size) met are kept in device
 * memory with the handle - a pipeline resizes equally sized images to the same scales pair after pair. */
int cvhip_resize_lanczos3(cvhip_device *dev, const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst, uint32_t nw,
                          uint32_t nh);

/* ------------------------------------------------------------------------------------------
 * ORB — replaces orb::extract_points (orb.rs:50-84).
 * out_xy: 2*cap u32 (x, y), out_desc: 8*cap u32, *out_n = keypoints written (<= cap).
 * Order = the reference's (Harris-descending stable, then BRIEF filter).  cap >= 10000 to
 * receive everything the reference returns (MAX_KEYPOINTS, orb.rs:41).
 * progress (may be NULL) replaces `Option<&PL>` (orb.rs:43-53): the reference reports from inside its parallel
 * loops, FAST rows 0 .. 0.20 (orb.rs:93-100), scores .. 0.25 (:112-118), non-maximum suppression .. 0.35 (:138-146),
 * Harris .. 0.70 (:60-66), BRIEF .. 1.0 (:358-363); here the same positions are reported on the calling thread as
 * the stages complete (0.25 and 1.0 after the stage's results have reached the host, the others at submission).
 * ---------------------------------------------------------------------------------------- */
int cvhip_orb_extract(cvhip_device *dev, const uint8_t *img, uint32_t w, uint32_t h, uint32_t cap,
                      uint32_t *out_xy, uint32_t *out_desc, uint32_t *out_n, cvhip_progress_fn progress, void *user);
/* The same for n_images independent images in one call - the pyramid levels of an image, as match_keypoints' per-image
 * loop extracts them (reconstruction.rs:418-458), or the images of a set.  An extraction has three host round trips
 * (corner counts; results - and, for an image with a keypoint inside the orientation guard band, patch moments out /
 * libm orientations in); the batch enqueues every image's work of a stage before it waits, so it pays them once.  Results per image are exactly cvhip_orb_extract's.  imgs / ws / hs / out_xy / out_desc: n_images entries
 * (each out_xy[i]: 2*cap u32, out_desc[i]: 8*cap u32); out_n: n_images counts. */
/* Test hook of the orientation step.  The reference computes atan2 / sin / cos with libm (orb.rs:337-341, 365-366); the
 * extraction uses the device's f64 functions wherever that provably gives the same rounded sample offsets - every
 * round(o_y cos - o_x sin) farther than `guard` from a half-integer - and redoes an image with the host's libm
 * otherwise.  Default 1e-9 (the functions agree to a few ulp: offsets move by < 1e-12).  A larger guard sends more
 * images down the host path, 0 switches the device orientation off (every image takes the host path).  Results are
 * identical for every setting. */
int cvhip_orb_set_orientation_guard(cvhip_device *dev, double guard);
int cvhip_orb_extract_batch(cvhip_device *dev, uint32_t n_images, const uint8_t *const *imgs, const uint32_t *ws,
                            const uint32_t *hs, uint32_t cap, uint32_t *const *out_xy, uint32_t *const *out_desc,
                            uint32_t *out_n, cvhip_progress_fn progress, void *user);

/* ------------------------------------------------------------------------------------------
 * Keypoint matcher — replaces KeypointMatching::match_points (pointmatching.rs:43-77).
 * xy: 2 u32 per keypoint, desc: 8 u32 per keypoint.  out_matches: 4*n1 u32 (x1,y1,x2,y2),
 * sorted by Hamming distance (stable); out_dist (n1, may be NULL).
 * ---------------------------------------------------------------------------------------- */
int cvhip_match_points(cvhip_device *dev, const uint32_t *xy1, const uint32_t *desc1, uint32_t n1,
                       const uint32_t *xy2, const uint32_t *desc2, uint32_t n2, uint32_t threshold,
                       uint32_t *out_matches, uint32_t *out_dist, uint32_t *out_n);

/* ------------------------------------------------------------------------------------------
 * RANSAC hypothesis scoring — replaces the all-matches fold of FundamentalMatrix::validate_f
 * (fundamentalmatrix.rs:210-216) with fits_model/reprojection_error (:452-471), for H
 * hypotheses at once.  F: 9*H doubles row-major, matches: 4*N u32 (x1,y1,x2,y2).
 * out_count[h] = inliers, out_err_sum[h] = sum of their errors in match order.
 * ---------------------------------------------------------------------------------------- */
int cvhip_ransac_score(cvhip_device *dev, const double *F, uint32_t H, const uint32_t *matches, uint32_t N,
                       double t, uint32_t *out_count, double *out_err_sum);

/* ------------------------------------------------------------------------------------------
 * Whole affine RANSAC on the device — replaces
 * FundamentalMatrix::new(Affine, _).find_ransac(matches) (fundamentalmatrix.rs:72-147, 155-286,
 * 231-239): sampling, the 4-point model fit and the sample checks run next to the scoring kernel;
 * the host only reads one early-exit word per 50 000-iteration round.  matches: 4*N u32
 * (x1,y1,x2,y2) sorted by descriptor distance as KeypointMatching returns them (the first 5000 are
 * sampled).  out_F: 9 doubles row-major; out_inlier_mask: N bytes (may be NULL).  The reference's RNG
 * is OS-seeded (not reproducible), so equality with it is statistical; `seed` makes this one
 * reproducible.  Returns CVHIP_ERR_NO_MODEL with the reference's RansacError messages.
 * ---------------------------------------------------------------------------------------- */
int cvhip_ransac_affine(cvhip_device *dev, const uint32_t *matches, uint32_t N, uint64_t seed, double *out_F,
                        uint32_t *out_inlier_count, uint8_t *out_inlier_mask);

/* The basis of the 7-point pencil in calculate_model_perspective (fundamentalmatrix.rs:309-322).
 * CVHIP_PENCIL_THIN_SVD (default) is the reference as written: `a.svd(false, true)` on the 7 x 9 system is nalgebra's
 * thin decomposition (v_t: 7 x 9, singular values descending), so `v_t.row(nrows - 2)` / `.row(nrows - 1)` are rows 5 and
 * 6 - the right singular vectors of the two SMALLEST of the seven singular values, not the null space.  Its hypotheses
 * do not fit their own sample until validate_f's optimize_perspective_f (:201-205) has run on EVERY root, which the
 * device does.  Sign convention of a singular vector (nalgebra's is not observable from its published interface): the
 * entry of largest magnitude is positive.  CVHIP_PENCIL_NULL_SPACE is the 7-point algorithm as published (the true
 * null space of A: rounds 2-3 of this build), kept as an option: better hypotheses, and ~3x cheaper - not the reference's. */
#define CVHIP_PENCIL_THIN_SVD 0
#define CVHIP_PENCIL_NULL_SPACE 1
int cvhip_ransac_set_pencil(cvhip_device *dev, int pencil);
/* Test hook: with the thin-SVD pencil validate_f's least_squares runs as two passes over the roots - up to the first
 * accepted step for all of them, then the accepting ones from their start - on persistent waves whose lanes take the next
 * root off the queue as they finish (enable = 2, default), as the same two passes with a root per thread (1), or as the
 * scalar loop itself in one kernel (0).  Same values, bit for bit. */
int cvhip_ransac_set_lm_pipeline(cvhip_device *dev, int enable);
/* The counting kernel's f32 screen over the head of the match list as [hypotheses x 12] x [12 x matches] products
 * (v_mfma_f32_16x16x4_f32; whoever is still alive behind the head continues in the vector kernel): enable = 1.  Default 0, the
 * vector kernel alone: in this form the matrix and the vector phase of every wave follow each other in lockstep
 * (SQ_VALU_MFMA_COEXEC_CYCLES = 0; within one wave the two pipes serialise), so it saves nothing - measured 47 against 45 ms for
 * config 5's RANSAC stage.  The counts are the f64 counts either way. */
int cvhip_ransac_set_count_mfma(cvhip_device *dev, int enable);
/* Test hook of the round scheduler: batches of rounds are normally scored as their generators finish (the host polls
 * their events, for at most 50 ms per decision); enable = 1 takes the branch that polling falls back to - the oldest
 * pending batch is enqueued behind its event, in order, no polling.  Same result (Ord's maximum does not depend on the
 * order; equal hypotheses are settled by their position in iteration order). */
int cvhip_ransac_set_in_order(cvhip_device *dev, int enable);

/* Perspective RANSAC on the device — FundamentalMatrix::new(Perspective, max_dimension).find_ransac
 * (fundamentalmatrix.rs:72-147, 155-229, 289-389): per sample the 7-point model (the pencil's basis as
 * cvhip_ransac_set_pencil says - default: rows 5 and 6 of the thin SVD, as the reference writes it -, the
 * determinant cubic, the reference's rank and sign-consistency checks, up to three roots),
 * every surviving root scored against ALL matches, best = most inliers then smallest mean error, early
 * exit above 50 000 inliers; t = 0.01 * max_dimension.  `rounds` = number of 50 000-sample rounds (0 or
 * more than 20 = the reference's 20).  validate_f's per-hypothesis optimize_perspective_f (:201-205: the LM
 * over the sample and the rank test on the re-parametrised matrix) runs on the device with the rest.  Not done
 * here: the final LM refit of optimize_result (:246-256): cvhip_optimize_perspective_f below, called by the
 * host layers.  out_F: the best hypothesis
 * (normalised by F[2][2]); mask/count: its inliers.  Statistical parity, as above. */
int cvhip_ransac_perspective(cvhip_device *dev, const uint32_t *matches, uint32_t N, double max_dimension,
                             uint64_t seed, uint32_t rounds, double *out_F, uint32_t *out_inlier_count,
                             uint8_t *out_inlier_mask);
/* Test hook of the generator above: the hypotheses of B caller-chosen samples (sample_idx: 7 match indices
 * each, host memory) exactly as they go on to the all-matches fold - i.e. after calculate_model_perspective
 * (:289-389) AND validate_f's finiteness test, optimize_perspective_f over the sample (LM + rank test on the
 * re-parametrised matrix, :201-205, :391-426) and sample-fit test (:206-209) -> out_F: B x 3 x 9 doubles, NaN
 * where a root does not exist or is rejected; t as in fits_model.  Compared with the oracle's numpy restatement
 * (oracle/cvref_fm.py) on identical samples. */
int cvhip_ransac_perspective_models(cvhip_device *dev, const uint32_t *matches, uint32_t N, const uint32_t *sample_idx,
                                    uint32_t B, double t, double *out_F);
/* Same for the affine generator (calculate_model_affine :260-286 + validate_f's checks): sample_idx holds 4
 * match indices per sample -> out_F: B x 9 doubles, NaN where the sample is rejected. */
int cvhip_ransac_affine_models(cvhip_device *dev, const uint32_t *matches, uint32_t N, const uint32_t *sample_idx,
                               uint32_t B, double t, double *out_F);
/* fits_model (fundamentalmatrix.rs:452-458) of one F for every match - the inlier filter of optimize_result
 * (:233-236, 248-254).  out_mask: N bytes, 1 = inlier.  Host or device pointers. */
int cvhip_fits_model(cvhip_device *dev, const double *F, const uint32_t *matches, uint32_t N, double t, uint8_t *out_mask);
/* Test hook: the scoring of ONE RANSAC round for caller-given hypotheses, as the device loops run it - inlier counts
 * from the counting kernel (one wave per hypothesis; pairs decided in packed f32 against rigorous error bounds, the
 * reference's f64 expression inside the guard band, so the counts are exact) and the reference's ordered error sum
 * for the hypotheses tied at the largest count (0 for all others: Ord, :623-649, never looks at theirs).  No pruning.
 * out_count must equal cvhip_ransac_score's counts; out_err_sum its sums where non-zero. */
int cvhip_ransac_round_score(cvhip_device *dev, const double *F, uint32_t H, const uint32_t *matches, uint32_t N, double t,
                             uint32_t *out_count, double *out_err_sum);
/* Test hook: the device loops' rounds on caller-given hypotheses: F is cut into `rounds` consecutive slices and every
 * slice is scored as cvhip_ransac_perspective / cvhip_find_ransac score a generated round - the counting kernel on its
 * own re-sorted copy of the match list, abandoning hypotheses against the best of the earlier slices and the running
 * maximum of the launch; the round's candidates; tie-break sums only where counts tie; the pick.  Out: the best
 * hypothesis after the last slice (its nine doubles, inlier count, mean error - NaN where no tie ever asked for it -
 * and its index in F, -1 if none reached min_count; the index needs F on the host).  Must be Ord's maximum
 * (fundamentalmatrix.rs:623-649) over cvhip_ransac_score's (count, err_sum / count), first of equals by index. */
int cvhip_ransac_rounds_pick(cvhip_device *dev, const double *F, uint32_t H, uint32_t rounds, const uint32_t *matches, uint32_t N,
                             double t, uint32_t min_count, double *out_F, uint32_t *out_count, double *out_mean_error,
                             int64_t *out_index);
/* FundamentalMatrix::new(projection, max_dimension).find_ransac(matches) in one call (fundamentalmatrix.rs:72-147
 * and optimize_result :231-257): cvhip_ransac_affine for projection 0 (max_dimension unused); for projection 1
 * cvhip_ransac_perspective, then the LM refit of the winner on its inliers (cvhip_optimize_perspective_f_device: the
 * reference's loop, values equal to the host function's) and the inliers of the refitted matrix.  This is what reconstruction.rs:502-526
 * calls.  out_F: 9 doubles row-major; out_inlier_mask: N bytes (may be NULL).
 * progress / report_matches (either may be NULL) replace `Option<&PL>` (fundamentalmatrix.rs:41-47, 103): the reference
 * reports iteration / ransac_k per iteration (:119-123) and the running maximum of the inlier counts (:126-131); here
 * both are reported once per 50 000-iteration round on the calling thread - pos = finished rounds / 20 - and, because
 * report_matches needs the round's best count on the host, a listener costs one 4-byte read per round that a call
 * without one only pays where the early exit (:135-141) can fire. */
int cvhip_find_ransac(cvhip_device *dev, int projection, const uint32_t *matches, uint32_t N, double max_dimension,
                      uint64_t seed, double *out_F, uint32_t *out_inlier_count, uint8_t *out_inlier_mask,
                      cvhip_progress_fn progress, cvhip_matches_fn report_matches, void *user);
/* optimize_perspective_f (fundamentalmatrix.rs:391-426) as optimize_result applies it to the winning model
 * (:246): the reference's own Levenberg-Marquardt loop (least_squares, :515-621) with its analytic Jacobian
 * (f_jacobian, :473-512) over the 7 free parameters of F (:429-449), then the rank test (:418-423).  Host
 * arithmetic, as in the reference (no device needed): matches = the n inliers, 4 x uint32 each.  out_F = the
 * refitted matrix and *out_refined = 1, or F itself and 0 where the reference's function returns None
 * (`.unwrap_or(res.f)`).  The loop is restated as written, including where it departs from the textbook
 * method (see the implementation's header); it is deterministic, so the oracle's restatement is compared
 * value for value. */
int cvhip_optimize_perspective_f(const double *F, const uint32_t *matches, uint32_t n, double *out_F, int *out_refined);
/* The same function on the device - what cvhip_find_ransac uses: one workgroup runs the loop's statements with the
 * per-observation work spread over its threads and every long dot product evaluated by eight threads, one per
 * partial sum of the reference's `dot`, so additions happen in the serial function's order and the result equals
 * cvhip_optimize_perspective_f's bit for bit (tested).  ~19 000 inliers: 12 ms on the host, well under 1 ms here.
 * F, matches: host or device pointers; out_F (9 doubles), out_refined: host. */
int cvhip_optimize_perspective_f_device(cvhip_device *dev, const double *F, const uint32_t *matches, uint32_t n,
                                        double *out_F, int *out_refined);

#ifdef __cplusplus
}
#endif
#endif
