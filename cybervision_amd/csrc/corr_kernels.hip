// corr_kernels.hip — dense stereo-correlation kernels for gfx950 (MI355X).
//
// Semantics are the reference's --mode=cpu path (zlogic/cybervision src/correlation/mod.rs);
// each kernel cites the lines it implements.  Everything that decides a result (window
// statistics, the 121-term correlation sums, f64 epipolar geometry and range statistics) is
// evaluated in the reference's own operation order with separately rounded mul and add
// (-ffp-contract=off), so match indices AND scores are bit-identical to the CPU path.
//
// Level grids are kept COMPACT: level k's matches live in an lw x lh array in level
// coordinates instead of being scattered with stride 2^k into a full-resolution sparse grid
// (mod.rs:311-316).  Because every finer level overwrites all cells a coarser level wrote,
// the neighbour scan of estimate_search_range (mod.rs:481-517) and the cross-check window
// (mod.rs:595-623) visit exactly the same matches in the same row-major order either way
// (SURVEY.md §8a N3); only the empty cells are skipped.
#include "cvhip_internal.hpp"

namespace cvhip {

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t f64_to_u32_sat(double v) // Rust `as usize`, clamped to 2^31
{
    if (!(v > 0.0)) return 0u;
    if (v >= 2147483648.0) return 0x80000000u;
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t f32_to_u32_sat(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 2147483648.0f) return 0x80000000u;
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t sat_sub_u32(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }
__device__ __forceinline__ bool finite_f32(float v) { return fabsf(v) < __builtin_inff(); }
__device__ __forceinline__ bool finite_f64(double v) { return fabs(v) < __builtin_inf(); }

// 12 bytes starting at an arbitrary byte address (gfx950 global loads may be unaligned).
struct Row12 {
    uint32_t a, b, c;
};
__device__ __forceinline__ Row12 load_row12(const uint8_t *p)
{
    Row12 r;
    __builtin_memcpy(&r.a, p, 4);
    __builtin_memcpy(&r.b, p + 4, 4);
    __builtin_memcpy(&r.c, p + 8, 4);
    return r;
}
__device__ __forceinline__ float byte_f32(uint32_t v, int i) { return (float)((v >> (8 * i)) & 0xFFu); }

// EpipolarLine (mod.rs:83-87) for level pixel (px, py); get_epipolar_line, mod.rs:386-409.
// F*p1 in nalgebra's gemv order: ((F[i][0]*p0) + F[i][1]*p1) + F[i][2]*p2.
struct Line {
    double cx, cy, ax, ay;
    int ox, oy;
};
__device__ __forceinline__ Line epipolar_line(const CorrParams &p, uint32_t px, uint32_t py)
{
    const double scale = (double)p.scale;
    const double p0 = (double)px / scale, p1 = (double)py / scale;
    double f[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double acc = p.F[i * 3 + 0] * p0;
        acc = p.F[i * 3 + 1] * p1 + acc;
        acc = p.F[i * 3 + 2] * 1.0 + acc;
        f[i] = acc;
    }
    Line e;
    if (fabs(f[0]) > fabs(f[1])) {
        e.cx = -f[1] / f[0];
        e.cy = 1.0;
        e.ax = -scale * f[2] / f[0];
        e.ay = 0.0;
        e.ox = 1;
        e.oy = 0;
    } else {
        e.cx = 1.0;
        e.cy = -f[0] / f[1];
        e.ax = 0.0;
        e.ay = -scale * f[2] / f[1];
        e.ox = 0;
        e.oy = 1;
    }
    return e;
}
__device__ __forceinline__ bool line_finite(const Line &e)
{
    return finite_f64(e.cx) && finite_f64(e.cy) && finite_f64(e.ax) && finite_f64(e.ay);
}
// corridor_end of correlate_point, mod.rs:347-350
__device__ __forceinline__ uint32_t corridor_end_of(const CorrParams &p, const Line &e)
{
    return fabs(e.cx) > fabs(e.cy) ? sat_sub_u32(p.w2, KERNEL_SIZE) : sat_sub_u32(p.h2, KERNEL_SIZE);
}

// ---------------------------------------------------------------------------------------------
// window_stats: compute_image_point_data (mod.rs:632-694) == the avg/stdev half of
// compute_point_data (mod.rs:702-735).  stats[i] = (avg, stdev), NaN outside the 5-px border.
// avg: the reference sums u8 values in f32; every partial sum is an integer < 2^24, so an
// integer sum converted once is bit-identical.  stdev: serial row-major f32 sum of squares.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void window_stats_kernel(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                            float2 *__restrict__ stats)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float nan = __builtin_nanf("");
    float2 out = make_float2(nan, nan);
    if (x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < w && y + KERNEL_SIZE < h) {
        const uint8_t *base = img + (size_t)(y - KERNEL_SIZE) * w + (x - KERNEL_SIZE);
        uint32_t isum = 0;
        Row12 rows[KERNEL_WIDTH];
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) {
            rows[r] = load_row12(base + (size_t)r * w);
            isum += __builtin_amdgcn_udot4(rows[r].a, 0x01010101u, 0u, false);
            isum += __builtin_amdgcn_udot4(rows[r].b, 0x01010101u, 0u, false);
            isum += __builtin_amdgcn_udot4(rows[r].c, 0x00010101u, 0u, false);
        }
        const float avg = (float)isum / (float)KERNEL_POINT_COUNT;
        float sd = 0.0f;
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) {
#pragma unroll
            for (int c = 0; c < KERNEL_WIDTH; c++) {
                const uint32_t wv = c < 4 ? rows[r].a : (c < 8 ? rows[r].b : rows[r].c);
                const float delta = byte_f32(wv, c & 3) - avg;
                sd += delta * delta;
            }
        }
        out = make_float2(avg, sqrtf(sd / (float)KERNEL_POINT_COUNT));
    }
    stats[(size_t)y * w + x] = out;
}

void launch_window_stats(const uint8_t *img, uint32_t w, uint32_t h, float2 *stats, hipStream_t s)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(window_stats_kernel, grid, dim3(256), 0, s, img, w, h, stats);
}

// ---------------------------------------------------------------------------------------------
// search_range: estimate_search_range (mod.rs:468-540) on the compact previous-level grid.
// One thread per searched pixel; writes start | end << 16, or RANGE_NONE.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void search_range_kernel(CorrParams p, const float2 *__restrict__ stats1,
                                                            const uint2 *__restrict__ prev,
                                                            uint32_t *__restrict__ range)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = p.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w1 || y >= p.row1) return;
    uint32_t out = RANGE_NONE;
    const bool interior = x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < p.w1 && y + KERNEL_SIZE < p.h1;
    if (interior) {
        const float2 st1 = stats1[(size_t)y * p.w1 + x];
        if (finite_f32(st1.y) && !(fabsf(st1.y) < p.min_stdev)) { // mod.rs:334 (same outcome, skipped early)
            const Line e = epipolar_line(p, x, y);
            if (line_finite(e)) {
                const uint32_t corridor_start = KERNEL_SIZE;
                const uint32_t corridor_end = corridor_end_of(p, e);
                const float scale = p.scale;
                // mod.rs:481-491, window in FULL-RES cells
                uint32_t x_min = f32_to_u32_sat(floorf((float)sat_sub_u32(x, NEIGHBOR_DISTANCE) / scale));
                uint32_t x_max = f32_to_u32_sat(ceilf((float)(x + NEIGHBOR_DISTANCE) / scale));
                uint32_t y_min = f32_to_u32_sat(floorf((float)sat_sub_u32(y, NEIGHBOR_DISTANCE) / scale));
                uint32_t y_max = f32_to_u32_sat(ceilf((float)(y + NEIGHBOR_DISTANCE) / scale));
                x_min = min(x_min, p.gw);
                x_max = min(x_max, p.gw);
                y_min = min(y_min, p.gh);
                y_max = min(y_max, p.gh);
                const bool corridor_vertical = fabs(e.cy) > fabs(e.cx);
                // occupied cells are the previous level's: full-res X = x' << pk, x' < pw
                const uint32_t step = 1u << p.pk;
                const uint32_t xs0 = (x_min + step - 1) >> p.pk, xs1 = min((x_max + step - 1) >> p.pk, p.pw);
                const uint32_t ys0 = (y_min + step - 1) >> p.pk, ys1 = min((y_max + step - 1) >> p.pk, p.ph);
                const double dscale = (double)scale;

                double mid_corridor = 0.0;
                uint32_t neighbor_count = 0;
                for (uint32_t yy = ys0; yy < ys1; yy++) {
                    for (uint32_t xx = xs0; xx < xs1; xx++) {
                        const uint32_t cell = prev[(size_t)yy * p.pw + xx].x;
                        if (cell == CELL_NONE) continue;
                        const double p2x = dscale * (double)((cell & 0xFFFFu) << p.pk);
                        const double p2y = dscale * (double)((cell >> 16) << p.pk);
                        const double corridor_pos = corridor_vertical ? (p2y - e.ay) / e.cy : (p2x - e.ax) / e.cx;
                        neighbor_count += 1;
                        mid_corridor += corridor_pos;
                    }
                }
                if (neighbor_count != 0) {
                    mid_corridor /= (double)neighbor_count;
                    double range_stdev = 0.0;
                    for (uint32_t yy = ys0; yy < ys1; yy++) {
                        for (uint32_t xx = xs0; xx < xs1; xx++) {
                            const uint32_t cell = prev[(size_t)yy * p.pw + xx].x;
                            if (cell == CELL_NONE) continue;
                            const double p2x = dscale * (double)((cell & 0xFFFFu) << p.pk);
                            const double p2y = dscale * (double)((cell >> 16) << p.pk);
                            const double corridor_pos =
                                corridor_vertical ? (p2y - e.ay) / e.cy : (p2x - e.ax) / e.cx;
                            const double delta = corridor_pos - mid_corridor;
                            range_stdev += delta * delta;
                        }
                    }
                    range_stdev = sqrt(range_stdev / (double)neighbor_count);
                    const uint32_t center = f64_to_u32_sat(round(mid_corridor));
                    const uint32_t length = f64_to_u32_sat(round(p.min_range + range_stdev * p.extend_range));
                    uint32_t s0 = sat_sub_u32(center, length);
                    s0 = s0 < corridor_start ? corridor_start : (s0 > corridor_end ? corridor_end : s0);
                    const uint64_t s1w = (uint64_t)center + (uint64_t)length; // saturating_add
                    uint32_t s1 = s1w > (uint64_t)corridor_end ? corridor_end : (uint32_t)s1w;
                    s1 = s1 < s0 ? s0 : s1;
                    out = s0 | (s1 << 16);
                }
            }
        }
    }
    range[(size_t)y * p.w1 + x] = out;
}

void launch_search_range(const CorrParams &p, const float2 *stats1, const uint2 *prev, uint32_t *range,
                         hipStream_t s)
{
    if (p.row1 <= p.row0) return;
    dim3 grid((p.w1 + 63) / 64, (p.row1 - p.row0 + 3) / 4);
    hipLaunchKernelGGL(search_range_kernel, grid, dim3(256), 0, s, p, stats1, prev, range);
}

// ---------------------------------------------------------------------------------------------
// search: correlate_point + correlate_corridor_area (mod.rs:321-384, 411-466).
// One thread per searched pixel; the 121 window deltas live in registers; for every candidate
// the 121-term sum is a serial f32 chain in row-major order (mul, then add).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void search_kernel(CorrParams p, const uint8_t *__restrict__ img1,
                                                      const uint8_t *__restrict__ img2,
                                                      const float2 *__restrict__ stats1,
                                                      const float2 *__restrict__ stats2,
                                                      const uint32_t *__restrict__ range,
                                                      uint2 *__restrict__ out,
                                                      unsigned long long *__restrict__ cand_counter)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = p.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool in_image = x < p.w1 && y < p.row1;
    uint32_t best_xy = CELL_NONE;
    float best_corr = __builtin_nanf("");
    uint32_t evaluated = 0;

    const bool interior =
        in_image && x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < p.w1 && y + KERNEL_SIZE < p.h1;
    if (interior) {
        const float2 st1 = stats1[(size_t)y * p.w1 + x];
        const Line e = epipolar_line(p, x, y);
        bool ok = finite_f32(st1.y) && !(fabsf(st1.y) < p.min_stdev) && line_finite(e); // mod.rs:334-345
        uint32_t r0 = KERNEL_SIZE, r1 = corridor_end_of(p, e);
        if (ok && !p.first_pass) { // mod.rs:351-364
            const uint32_t rg = range[(size_t)y * p.w1 + x];
            ok = rg != RANGE_NONE;
            r0 = rg & 0xFFFFu;
            r1 = rg >> 16;
        }
        if (ok && r0 < r1) {
            // compute_point_data deltas (mod.rs:727-731); avg identical to stats1.x
            float d1[KERNEL_POINT_COUNT];
            {
                const uint8_t *base = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (x - KERNEL_SIZE);
#pragma unroll
                for (int r = 0; r < KERNEL_WIDTH; r++) {
                    const Row12 row = load_row12(base + (size_t)r * p.w1);
#pragma unroll
                    for (int c = 0; c < KERNEL_WIDTH; c++) {
                        const uint32_t wv = c < 4 ? row.a : (c < 8 ? row.b : row.c);
                        d1[r * KERNEL_WIDTH + c] = byte_f32(wv, c & 3) - st1.x;
                    }
                }
            }
            const float stdev1 = st1.y;
            bool have = false;
            float bcorr = 0.0f;
            uint32_t bx = 0, by = 0;
            for (int off = -p.corridor_size; off <= p.corridor_size; off++) { // mod.rs:371-381
                const double offx = (double)(off * e.ox), offy = (double)(off * e.oy);
                for (uint32_t i = r0; i < r1; i++) { // mod.rs:423
                    const double x2d = (e.cx * (double)i + e.ax) + offx;
                    const double y2d = (e.cy * (double)i + e.ay) + offy;
                    const uint32_t x2 = f64_to_u32_sat(floor(x2d));
                    const uint32_t y2 = f64_to_u32_sat(floor(y2d));
                    if (x2 < KERNEL_SIZE || x2 >= p.w2 - KERNEL_SIZE || y2 < KERNEL_SIZE || y2 >= p.h2 - KERNEL_SIZE)
                        continue;
                    const float2 st2 = stats2[(size_t)y2 * p.w2 + x2];
                    if (!finite_f32(st2.y) || fabsf(st2.y) < p.min_stdev) continue;
                    evaluated++;
                    const float avg2 = st2.x;
                    float corr = 0.0f;
                    const uint8_t *base = img2 + (size_t)(y2 - KERNEL_SIZE) * p.w2 + (x2 - KERNEL_SIZE);
#pragma unroll
                    for (int r = 0; r < KERNEL_WIDTH; r++) {
                        const Row12 row = load_row12(base + (size_t)r * p.w2);
#pragma unroll
                        for (int c = 0; c < KERNEL_WIDTH; c++) {
                            const uint32_t wv = c < 4 ? row.a : (c < 8 ? row.b : row.c);
                            const float delta2 = byte_f32(wv, c & 3) - avg2;
                            corr += d1[r * KERNEL_WIDTH + c] * delta2;
                        }
                    }
                    corr /= stdev1 * st2.y * (float)KERNEL_POINT_COUNT; // mod.rs:454
                    if (corr >= p.threshold && (!have || corr > bcorr)) { // mod.rs:456-464
                        have = true;
                        bcorr = corr;
                        bx = x2;
                        by = y2;
                    }
                }
            }
            if (have) {
                best_xy = bx | (by << 16);
                best_corr = bcorr;
            }
        }
    }
    if (in_image) out[(size_t)y * p.w1 + x] = make_uint2(best_xy, __float_as_uint(best_corr));
    if (cand_counter) {
        // wave-level sum, one atomic per wave
        uint32_t v = evaluated;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_down(v, sft, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(cand_counter, (unsigned long long)v);
    }
}

void launch_search(const CorrParams &p, const uint8_t *img1, const uint8_t *img2, const float2 *stats1,
                   const float2 *stats2, const uint32_t *range, uint2 *out, unsigned long long *cand_counter,
                   hipStream_t s)
{
    if (p.row1 <= p.row0) return;
    dim3 grid((p.w1 + 63) / 64, (p.row1 - p.row0 + 3) / 4);
    hipLaunchKernelGGL(search_kernel, grid, dim3(256), 0, s, p, img1, img2, stats1, stats2, range, out, cand_counter);
}

// ---------------------------------------------------------------------------------------------
// cross_check: cross_check_filter / cross_check_point (mod.rs:552-624) in level coordinates.
// search_area = 4 * round(1/scale) full-res cells == 4 level cells on both grids, and a level
// match (x2, y2) is the full-res match (x2, y2) << k, so the window test reduces to +-4 level
// cells.  Each thread owns one cell of `own` and only reads `other`.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_check_kernel(uint2 *__restrict__ own, const uint2 *__restrict__ other,
                                                           uint32_t ow, uint32_t oh, uint32_t rw, uint32_t rh)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const uint32_t cell = own[(size_t)y * ow + x].x;
    if (cell == CELL_NONE) return;
    const uint32_t sa = CROSS_CHECK_SEARCH_AREA;
    const uint32_t mx = cell & 0xFFFFu, my = cell >> 16;
    const uint32_t min_x = min(sat_sub_u32(mx, sa), rw), max_x = min(mx + sa + 1, rw);
    const uint32_t min_y = min(sat_sub_u32(my, sa), rh), max_y = min(my + sa + 1, rh);
    const uint32_t r_min_x = sat_sub_u32(x, sa), r_max_x = x + sa + 1;
    const uint32_t r_min_y = sat_sub_u32(y, sa), r_max_y = y + sa + 1;
    bool found = false;
    for (uint32_t sy = min_y; sy < max_y && !found; sy++) {
        for (uint32_t sx = min_x; sx < max_x; sx++) {
            const uint32_t rm = other[(size_t)sy * rw + sx].x;
            if (rm == CELL_NONE) continue;
            const uint32_t rx = rm & 0xFFFFu, ry = rm >> 16;
            if (rx >= r_min_x && rx < r_max_x && ry >= r_min_y && ry < r_max_y) {
                found = true;
                break;
            }
        }
    }
    if (!found) own[(size_t)y * ow + x] = make_uint2(CELL_NONE, 0x7FC00000u);
}

void launch_cross_check(uint2 *own, const uint2 *other, uint32_t ow, uint32_t oh, uint32_t rw, uint32_t rh,
                        hipStream_t s)
{
    dim3 grid((ow + 63) / 64, (oh + 3) / 4);
    hipLaunchKernelGGL(cross_check_kernel, grid, dim3(256), 0, s, own, other, ow, oh, rw, rh);
}

// ---------------------------------------------------------------------------------------------
// expand_grid: the scatter of mod.rs:311-316 plus the Match position of mod.rs:459-462, applied
// once at complete(): full-res cell (x << k, y << k) = level cell (x, y) with the match scaled
// back by round(x2 / scale) = x2 << k.  All other full-res cells are None.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_grid_kernel(const uint2 *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                           uint32_t k, uint32_t gw, uint32_t gh,
                                                           int32_t *__restrict__ out_xy, float *__restrict__ out_corr)
{
    const uint32_t gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= gw || gy >= gh) return;
    int32_t ox = -1, oy = -1;
    float oc = __builtin_nanf("");
    const uint32_t mask = (1u << k) - 1u;
    if ((gx & mask) == 0 && (gy & mask) == 0) {
        const uint32_t lx = gx >> k, ly = gy >> k;
        if (lx < lw && ly < lh) {
            const uint2 c = cells[(size_t)ly * lw + lx];
            if (c.x != CELL_NONE) {
                ox = (int32_t)((c.x & 0xFFFFu) << k);
                oy = (int32_t)((c.x >> 16) << k);
                oc = __uint_as_float(c.y);
            }
        }
    }
    const size_t o = (size_t)gy * gw + gx;
    reinterpret_cast<int2 *>(out_xy)[o] = make_int2(ox, oy);
    if (out_corr) out_corr[o] = oc;
}

void launch_expand_grid(const uint2 *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                        int32_t *out_xy, float *out_corr, hipStream_t s)
{
    dim3 grid((gw + 63) / 64, (gh + 3) / 4);
    hipLaunchKernelGGL(expand_grid_kernel, grid, dim3(256), 0, s, cells, lw, lh, k, gw, gh, out_xy, out_corr);
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
void launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t s)
{
    if (!n) return;
    const unsigned blocks = (unsigned)min((size_t)2048, (n + 255) / 256);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, s, p, v, n);
}

} // namespace cvhip
