// track_kernels.hip — the dense consumer's track extension on the device-resident forward grid.
//
// Replaces Triangulation::extend_tracks (zlogic/cybervision src/triangulation.rs:1330-1419), the first thing the
// perspective pipeline does with a finished dense correlation (triangulation.rs:638, 697): every existing track
// that has a point in image 1 looks for the nearest dense match within `search_radius` of it (squared distance,
// FIRST minimum in row-major scan order, :1362-1382) and takes that match's image-2 point; the merged points are
// then cleared from the remaining grid - at the MATCHED point's coordinates, as the reference does (:1391-1393) -
// and every remaining Some cell starts a new track, in scan order (:1397-1416).  Integer only: bit-exact.
#include "cvhip_internal.hpp"

#include <string>

namespace cvhip {

__global__ __launch_bounds__(256) void extend_tracks_match_kernel(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                                   uint32_t k, uint32_t gw, uint32_t gh,
                                                                   const int2 *__restrict__ track_p1,
                                                                   unsigned long long n_tracks, uint32_t radius,
                                                                   int2 *__restrict__ out_p2, uint8_t *__restrict__ removed,
                                                                   uint32_t *__restrict__ oob)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tracks) return;
    const int2 p = track_p1[t];
    int2 res = make_int2(-1, -1);
    if (p.x >= 0 && p.y >= 0) { // track.get(image1_index)?
        const uint32_t px = (uint32_t)p.x, py = (uint32_t)p.y;
        // :1358-1361 - saturating_sub below, min(.., width/height) above: [p - r, p + r) clipped
        const uint32_t min_x = px > radius ? px - radius : 0u, min_y = py > radius ? py - radius : 0u;
        const uint32_t max_x = min(px + radius, gw), max_y = min(py + radius, gh);
        bool have = false;
        unsigned long long best = 0;
        for (uint32_t y = min_y; y < max_y; y++)
            for (uint32_t x = min_x; x < max_x; x++) {
                uint32_t mx, my;
                if (!full_res_match(cells, lw, lh, k, x, y, mx, my)) continue;
                const unsigned long long dx = x > px ? x - px : px - x, dy = y > py ? y - py : py - y;
                const unsigned long long d = dx * dx + dy * dy;
                if (!have || d < best) { // is_none_or(distance < min_distance): the first minimum wins
                    have = true;
                    best = d;
                    res = make_int2((int)mx, (int)my);
                }
            }
        if (have) {
            // :1391-1393: *remaining_points.val_mut(track_point.x, track_point.y) = None - the image-1 grid indexed
            // with the image-2 point; Grid::val_mut asserts the bounds (data.rs:61-64)
            if ((uint32_t)res.x < gw && (uint32_t)res.y < gh) removed[(size_t)res.y * gw + res.x] = 1;
            else atomicAdd(oob, 1u);
        }
    }
    out_p2[t] = res;
}

__device__ __forceinline__ bool remaining_cell(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh, uint32_t k,
                                               uint32_t gw, uint32_t gh, const uint8_t *__restrict__ removed, size_t i,
                                               uint32_t &gx, uint32_t &gy, uint32_t &mx, uint32_t &my)
{
    if (i >= (size_t)gw * gh) return false;
    gx = (uint32_t)(i % gw);
    gy = (uint32_t)(i / gw);
    return full_res_match(cells, lw, lh, k, gx, gy, mx, my) && !removed[i];
}

__global__ __launch_bounds__(256) void extend_tracks_count_kernel(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                                   uint32_t k, uint32_t gw, uint32_t gh,
                                                                   const uint8_t *__restrict__ removed,
                                                                   uint32_t *__restrict__ block_counts)
{
    uint32_t gx, gy, mx, my;
    const bool f = remaining_cell(cells, lw, lh, k, gw, gh, removed, (size_t)blockIdx.x * 256 + threadIdx.x, gx, gy, mx, my);
    __shared__ uint32_t wsum[4];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void extend_tracks_write_kernel(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                                   uint32_t k, uint32_t gw, uint32_t gh,
                                                                   const uint8_t *__restrict__ removed,
                                                                   const uint32_t *__restrict__ block_offsets,
                                                                   unsigned long long cap, uint32_t *__restrict__ out_new_p1,
                                                                   uint32_t *__restrict__ out_new_p2)
{
    uint32_t gx = 0, gy = 0, mx = 0, my = 0;
    const bool f = remaining_cell(cells, lw, lh, k, gw, gh, removed, (size_t)blockIdx.x * 256 + threadIdx.x, gx, gy, mx, my);
    __shared__ uint32_t wsum[4];
    const unsigned long long b = __ballot(f);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
    __syncthreads();
    if (f) {
        unsigned long long off = block_offsets[blockIdx.x];
        for (uint32_t w = 0; w < wv; w++) off += wsum[w];
        off += (unsigned long long)__popcll(b & ((1ull << lane) - 1ull));
        if (off < cap) {
            reinterpret_cast<uint2 *>(out_new_p1)[off] = make_uint2(gx, gy);
            reinterpret_cast<uint2 *>(out_new_p2)[off] = make_uint2(mx, my);
        }
    }
}

void launch_extend_tracks_match(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                                const int2 *track_p1, unsigned long long n_tracks, uint32_t radius, int2 *out_p2,
                                uint8_t *removed, uint32_t *oob, hipStream_t s)
{
    if (!n_tracks) return;
    hipLaunchKernelGGL(extend_tracks_match_kernel, dim3((unsigned)((n_tracks + 255) / 256)), dim3(256), 0, s, cells, lw, lh, k,
                       gw, gh, track_p1, n_tracks, radius, out_p2, removed, oob);
}

void launch_extend_tracks_new(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                              const uint8_t *removed, uint32_t *block_counts, uint32_t *total, uint32_t *out_new_p1,
                              uint32_t *out_new_p2, unsigned long long cap, hipStream_t s)
{
    const uint32_t nblocks = (uint32_t)(((size_t)gw * gh + 255) / 256);
    hipLaunchKernelGGL(extend_tracks_count_kernel, dim3(nblocks), dim3(256), 0, s, cells, lw, lh, k, gw, gh, removed,
                       block_counts);
    launch_scan_u32(block_counts, nblocks, total, s);
    if (cap)
        hipLaunchKernelGGL(extend_tracks_write_kernel, dim3(nblocks), dim3(256), 0, s, cells, lw, lh, k, gw, gh, removed,
                           block_counts, cap, out_new_p1, out_new_p2);
}

} // namespace cvhip

using namespace cvhip;

namespace {
bool on_device(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
} // namespace

extern "C" int cvhip_extend_tracks(cvhip_ctx *ctx, const int32_t *track_p1, uint64_t n_tracks, uint32_t max_dimension2,
                                   int32_t *out_track_p2, uint32_t *out_new_p1, uint32_t *out_new_p2, uint64_t cap,
                                   uint64_t *out_n_new)
{
    if (!ctx || !out_n_new) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(cvhip::flush_level_calls(ctx));
    if (n_tracks && (!track_p1 || !out_track_p2)) return fail(CVHIP_ERR_INVALID, "track arrays are null");
    if (cap && (!out_new_p1 || !out_new_p2)) return fail(CVHIP_ERR_INVALID, "new-track arrays are null");
    CVHIP_TRY_HIP(hipSetDevice(ctx->dev->d.ordinal));
    hipStream_t s = ctx->dev->d.stream;
    DirState &ds = ctx->dir[0];
    *out_n_new = 0;
    if (!ds.valid) { // nothing correlated: no matches to merge, no new tracks
        if (n_tracks && !on_device(out_track_p2))
            for (uint64_t i = 0; i < 2 * n_tracks; i++) out_track_p2[i] = -1;
        else if (n_tracks)
            CVHIP_TRY_HIP(hipMemsetAsync(out_track_p2, 0xFF, n_tracks * 2 * sizeof(int32_t), s));
        return CVHIP_OK;
    }
    const size_t n = (size_t)ds.gw * ds.gh;
    const uint32_t nblocks = (uint32_t)((n + 255) / 256);
    // scratch that is free between pairs: the contender words (8 B per pixel) hold the removal map, the
    // search-interval buffer the block counts (+ total, + the out-of-bounds flag)
    if ((size_t)nblocks + 2 > ctx->max_px || n > ctx->max_px * sizeof(unsigned long long))
        return fail(CVHIP_ERR_INVALID, "image too small for the scratch buffers");
    uint8_t *removed = reinterpret_cast<uint8_t *>(ctx->contenders);
    uint32_t *counts = ctx->range, *total = ctx->range + nblocks, *oob = ctx->range + nblocks + 1;
    // EXTEND_TRACKS_SEARCH_RADIUS = 3, TRACKS_RADIUS_DENOMINATOR = 1000 (triangulation.rs:16, 19, 1346-1350)
    const uint32_t radius = max_dimension2 > 1000 ? (uint32_t)((uint64_t)3 * max_dimension2 / 1000) : 3u;
    const bool tp1_dev = n_tracks ? on_device(track_p1) : true, tp2_dev = n_tracks ? on_device(out_track_p2) : true;
    const bool n1_dev = cap ? on_device(out_new_p1) : true, n2_dev = cap ? on_device(out_new_p2) : true;
    int2 *d_tp1 = reinterpret_cast<int2 *>(const_cast<int32_t *>(track_p1)), *d_tp2 = reinterpret_cast<int2 *>(out_track_p2);
    uint32_t *d_n1 = out_new_p1, *d_n2 = out_new_p2;
    hipError_t e = hipMemsetAsync(removed, 0, n, s);
    if (e == hipSuccess) e = hipMemsetAsync(oob, 0, sizeof(uint32_t), s);
    if (e == hipSuccess && !tp1_dev) {
        e = hipMalloc(&d_tp1, n_tracks * sizeof(int2));
        if (e == hipSuccess) e = hipMemcpyAsync(d_tp1, track_p1, n_tracks * sizeof(int2), hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && !tp2_dev) e = hipMalloc(&d_tp2, n_tracks * sizeof(int2));
    if (e == hipSuccess && cap && !n1_dev) e = hipMalloc(&d_n1, (size_t)cap * 2 * sizeof(uint32_t));
    if (e == hipSuccess && cap && !n2_dev) e = hipMalloc(&d_n2, (size_t)cap * 2 * sizeof(uint32_t));
    uint32_t h_total = 0, h_oob = 0;
    if (e == hipSuccess) {
        launch_extend_tracks_match(ds.cells[ds.cur], ds.lw, ds.lh, ds.k, ds.gw, ds.gh, d_tp1, n_tracks, radius, d_tp2, removed,
                                   oob, s);
        launch_extend_tracks_new(ds.cells[ds.cur], ds.lw, ds.lh, ds.k, ds.gw, ds.gh, removed, counts, total, d_n1, d_n2, cap, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_total, total, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_oob, oob, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && n_tracks && !tp2_dev)
        e = hipMemcpyAsync(out_track_p2, d_tp2, n_tracks * sizeof(int2), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    const uint64_t written = h_total < cap ? h_total : cap;
    if (e == hipSuccess && written && !n1_dev)
        e = hipMemcpy(out_new_p1, d_n1, (size_t)written * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && written && !n2_dev)
        e = hipMemcpy(out_new_p2, d_n2, (size_t)written * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (!tp1_dev && d_tp1) (void)hipFree(d_tp1);
    if (!tp2_dev && d_tp2) (void)hipFree(d_tp2);
    if (cap && !n1_dev && d_n1) (void)hipFree(d_n1);
    if (cap && !n2_dev && d_n2) (void)hipFree(d_n2);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("extend_tracks: ") + hipGetErrorString(e));
    if (h_oob) return fail(CVHIP_ERR_INVALID, "Index out of bounds (a merged match lies outside the image-1 grid; the reference panics here, data.rs:61-64)");
    *out_n_new = h_total;
    return CVHIP_OK;
}
