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

#include <algorithm>
#include <cstdlib>
#include <type_traits>

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

// Workgroups are handed to the 8 XCDs round-robin in launch order, and every XCD has its own L2.  The tiled
// kernels below re-read their neighbours' halo rows/columns, so tile ids are remapped so that one XCD walks a
// contiguous run of tiles (bijective for any grid size): the halos then hit in that XCD's L2 instead of HBM.
struct TileId {
    uint32_t x, y;
};
__device__ __forceinline__ TileId xcd_tile()
{
    const uint32_t nx = gridDim.x, nwg = nx * gridDim.y, orig = blockIdx.y * nx + blockIdx.x;
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
    const uint32_t id = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
    TileId t;
    t.y = id / nx;
    t.x = id - t.y * nx;
    return t;
}

// The job of this workgroup when one launch carries the passes of both directions (blockIdx.z): the two SearchJob
// structs are the kernel's first two arguments, i.e. they sit back to back at the start of the kernel-argument
// segment, and the job is addressed THERE - a uniform pointer into constant memory - so that every member stays a
// scalar load the compiler can repeat wherever it needs the value, exactly like a plain by-value argument.  (Selecting
// between the two by-value structs instead copies them: whole-struct selection went through 700 B of scratch and
// made the box kernel 2.5x slower; member-by-member selection kept ~70 values live in SGPRs and spilled 30-80.)
typedef const __attribute__((address_space(4))) SearchJob *KernargJobPtr;
__device__ __forceinline__ const SearchJob &this_job()
{
    static_assert(sizeof(SearchJob) % 8 == 0, "the second job must directly follow the first in the kernarg segment");
    // (the address-space cast is undone by the compiler once this is inlined: the loads stay s_load from constant memory)
    return *(const SearchJob *)((KernargJobPtr)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.z);
}

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
// (avg, stdev) of a pixel from its statistics word {window sum | VALID << 31, f32 bits of stdev} (window_stats_kernel):
// avg = (sum of the window as f32) / 121 is the reference's own expression (mod.rs:657-671; the f32 sum of 121 bytes is
// an exact integer), so nothing is lost by not storing it.  Pixels outside the 5-px border carry stdev = NaN.
__device__ __forceinline__ float2 stats_of(uint2 w)
{
    return make_float2((float)(w.x & 0x7FFFFFFFu) / (float)KERNEL_POINT_COUNT, __uint_as_float(w.y));
}

// Serial row-major accumulation over one 11-byte window row, exactly as the reference's scalar loops do it
// (every subtraction, product and addition separately rounded, additions in element order); the independent
// subtractions and products go through the packed-f32 pipe two at a time (v_pk_add_f32 / v_pk_mul_f32 round
// each half like the scalar instructions).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t row12_word(const Row12 &r, int c) { return c < 4 ? r.a : (c < 8 ? r.b : r.c); }
// corr += (a_k - avg1) * (b_k - avg2), k = 0..10   (mod.rs:447-452 with the deltas of mod.rs:727-731)
__device__ __forceinline__ float row_corr_acc(float corr, const Row12 &ra, const Row12 &rb, float avg1, float avg2)
{
    const v2f a1 = {avg1, avg1}, a2 = {avg2, avg2};
#pragma unroll
    for (int c = 0; c < KERNEL_WIDTH - 1; c += 2) {
        const v2f fa = {byte_f32(row12_word(ra, c), c & 3), byte_f32(row12_word(ra, c + 1), (c + 1) & 3)};
        const v2f fb = {byte_f32(row12_word(rb, c), c & 3), byte_f32(row12_word(rb, c + 1), (c + 1) & 3)};
        const v2f pr = (fa - a1) * (fb - a2);
        corr += pr.x;
        corr += pr.y;
    }
    const float d1 = byte_f32(ra.c, 2) - avg1, d2 = byte_f32(rb.c, 2) - avg2;
    return corr + d1 * d2;
}
// sd += (a_k - avg)^2, k = 0..10   (mod.rs:727-733)
__device__ __forceinline__ float row_sq_acc(float sd, const Row12 &ra, float avg)
{
    const v2f a1 = {avg, avg};
#pragma unroll
    for (int c = 0; c < KERNEL_WIDTH - 1; c += 2) {
        const v2f fa = {byte_f32(row12_word(ra, c), c & 3), byte_f32(row12_word(ra, c + 1), (c + 1) & 3)};
        const v2f d = fa - a1;
        const v2f pr = d * d;
        sd += pr.x;
        sd += pr.y;
    }
    const float d1 = byte_f32(ra.c, 2) - avg;
    return sd + d1 * d1;
}

// EpipolarLine (mod.rs:83-87) for level pixel (px, py); get_epipolar_line, mod.rs:386-409.
// F*p1 in nalgebra's gemv order: ((F[i][0]*p0) + F[i][1]*p1) + F[i][2]*p2.
struct Line {
    double cx, cy, ax, ay;
    int ox, oy;
};
__device__ __forceinline__ Line epipolar_line(const CorrParams &p, uint32_t px, uint32_t py)
{
    const double scale = (double)p.scale;
    // p / scale (mod.rs:389-391) with scale = 2^-k: the quotient is exact, and so is this product
    const double up = (double)(1u << p.k), p0 = (double)px * up, p1 = (double)py * up;
    if (p.affine) { // the direction is a host-evaluated constant (CorrParams::affine); only the offset is per pixel
        double f2 = p.F[6] * p0;
        f2 = p.F[7] * p1 + f2;
        f2 = p.F[8] * 1.0 + f2;
        const double off = -scale * f2 / p.aff_div;
        Line e;
        const bool first = p.affine == 1;
        e.cx = first ? p.aff_c : 1.0;
        e.cy = first ? 1.0 : p.aff_c;
        e.ax = first ? off : 0.0;
        e.ay = first ? 0.0 : off;
        e.ox = first ? 1 : 0;
        e.oy = first ? 0 : 1;
        return e;
    }
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
//
// istats[i] = {window sum s (exact integer) | VALID << 31, f32 bits of stdev}: everything
// search2_kernel needs per candidate in one 8-byte word (avg = (float)s / 121 exactly, see above).
// VALID mirrors the reference's per-candidate test "stdev finite and >= min_stdev" (mod.rs:439)
// evaluated on the reference's own f32 stdev.
constexpr int WS_PITCH = 88; // bytes per staged row: 64 + 2*5 window columns, dword aligned start, padded

struct StatsJob { // one image's statistics pass
    const uint8_t *img;
    uint32_t w, h, row0, row1;
    uint2 *istats;
};

// Two pixels' chains at once: sd.x += (g_k.x - avg.x)^2 and sd.y += (g_k.y - avg.y)^2, k = 0..10 - each half is the serial
// chain of row_sq_acc on its own pixel (v_pk_add_f32 / v_pk_mul_f32 round each half like the scalar instructions), three
// instructions per two squared deviations instead of four.
__device__ __forceinline__ v2f row_sq_acc2(v2f sd, const v2f (&g)[KERNEL_WIDTH], v2f avg)
{
#pragma unroll
    for (int c = 0; c < KERNEL_WIDTH; c++) {
        const v2f d = g[c] - avg;
        sd += d * d;
    }
    return sd;
}

// Both images of a level in one launch (blockIdx.z picks the image; the grid covers the larger one).  zero_words:
// eight u32 cleared by the first thread - the work-list counts of the level's two search passes, which start
// after this kernel on the same stream.
constexpr int WS_PX = 8;                     // vertically adjacent pixels per lane
constexpr int WS_HALF = WS_PX / 2;           // pixel q shares its packed chain with pixel q + WS_HALF
constexpr int WS_ROWS = 4 * WS_PX;           // pixel rows per workgroup
constexpr int WS_LROWS = KERNEL_WIDTH + WS_PX - 1; // image rows one lane's windows cover
__global__ __launch_bounds__(256) void window_stats_kernel(StatsJob ja, StatsJob jb, float min_stdev,
                                                            uint32_t *__restrict__ zero_words)
{
    // The 64x32 tile's 74x42 source bytes are staged once in LDS (one dword load per thread instead of 33
    // unaligned loads per pixel, which made the kernel address-unit-bound); each lane then reads its 12
    // bytes per window row as four aligned LDS dwords and funnel-shifts them into place.  A lane's eight pixels
    // (x, y) .. (x, y + 7) cover 18 image rows R0 .. R17, extracted and summed once.  The serial chains of mod.rs:727-733
    // run two pixels to a packed instruction: pixel q's window row j is image row j + q, so the pair of rows
    // {R_r, R_r+4}, converted into the two halves of 11 register pairs, feeds step (r - q, .) of pixels q and q + 4 for
    // q = 0 .. 3 - every row is converted twice for eight pixels, and the per-pixel operation order is untouched.
    __shared__ uint32_t tile[(WS_ROWS + KERNEL_WIDTH - 1) * (WS_PITCH / 4)];
    if (zero_words && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) zero_words[threadIdx.x] = 0u;
    const StatsJob &job = blockIdx.z == 0 ? ja : jb;
    const uint8_t *__restrict__ img = job.img;
    const uint32_t w = job.w, h = job.h, row0 = job.row0, row1 = job.row1;
    uint2 *__restrict__ istats = job.istats;
    const TileId tid = xcd_tile();
    const uint32_t x0 = tid.x * 64, y0 = row0 + tid.y * WS_ROWS;
    if (x0 >= w || y0 >= row1) return; // the grid covers the larger image
    const int sx = (int)x0 - 8, sy = (int)y0 - KERNEL_SIZE; // staged origin; sx is 0 mod 4 relative to x0
    for (uint32_t u = threadIdx.x; u < (uint32_t)(WS_ROWS + KERNEL_WIDTH - 1) * (WS_PITCH / 4); u += 256) {
        const uint32_t r = u / (WS_PITCH / 4), c4 = (u - r * (WS_PITCH / 4)) * 4;
        const int gy = sy + (int)r, gx = sx + (int)c4;
        uint32_t v = 0;
        if (gy >= 0 && gy < (int)h && gx >= 0 && gx < (int)w) // bytes past the row end are never used
            __builtin_memcpy(&v, img + (size_t)gy * w + gx, 4);
        tile[u] = v;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t x = x0 + lane, yq = y0 + WS_PX * wv;
    if (x >= w || yq >= row1) return;
    const float nan = __builtin_nanf("");
    uint2 iout[WS_PX];
    bool in[WS_PX];
    bool any_in = false;
    const bool col_in = x >= KERNEL_SIZE && x + KERNEL_SIZE < w;
#pragma unroll
    for (int q = 0; q < WS_PX; q++) {
        iout[q] = make_uint2(0u, __float_as_uint(nan)); // outside the border: not VALID, stdev NaN
        in[q] = col_in && yq + q >= KERNEL_SIZE && yq + q + KERNEL_SIZE < h && yq + q < row1;
        any_in = any_in || in[q];
    }
    if (any_in) {
        const uint32_t off = lane + 3u; // window starts at byte (x - 5) - (x0 - 8) of the staged row
        const uint32_t d0 = off >> 2, sh = off & 3u;
        Row12 rows[WS_LROWS];
        uint32_t rs[WS_LROWS];
#pragma unroll
        for (int r = 0; r < WS_LROWS; r++) {
            const uint32_t *src = &tile[(WS_PX * wv + r) * (WS_PITCH / 4) + d0];
            const uint32_t q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
            rows[r].a = __builtin_amdgcn_alignbyte(q1, q0, sh);
            rows[r].b = __builtin_amdgcn_alignbyte(q2, q1, sh);
            rows[r].c = __builtin_amdgcn_alignbyte(q3, q2, sh);
            uint32_t t = __builtin_amdgcn_udot4(rows[r].a, 0x01010101u, 0u, false);
            t = __builtin_amdgcn_udot4(rows[r].b, 0x01010101u, t, false);
            rs[r] = __builtin_amdgcn_udot4(rows[r].c, 0x00010101u, t, false);
        }
        uint32_t isum[WS_PX];
        isum[0] = 0;
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) isum[0] += rs[r];
#pragma unroll
        for (int q = 1; q < WS_PX; q++) isum[q] = isum[q - 1] - rs[q - 1] + rs[q - 1 + KERNEL_WIDTH];
        v2f avg2[WS_HALF], sd2[WS_HALF];
#pragma unroll
        for (int q = 0; q < WS_HALF; q++) {
            avg2[q] = v2f{(float)isum[q] / (float)KERNEL_POINT_COUNT, (float)isum[q + WS_HALF] / (float)KERNEL_POINT_COUNT};
            sd2[q] = v2f{0.0f, 0.0f};
        }
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH + WS_HALF - 1; r++) {
            // one pair of rows at a time: without this the compiler converts all rows up front
            asm volatile("" : "+v"(rows[r].a), "+v"(rows[r].b), "+v"(rows[r].c));
#pragma unroll
            for (int q = 0; q < WS_HALF; q++) asm volatile("" : "+v"(sd2[q]));
            v2f g[KERNEL_WIDTH];
#pragma unroll
            for (int c = 0; c < KERNEL_WIDTH; c++)
                g[c] = v2f{byte_f32(row12_word(rows[r], c), c & 3), byte_f32(row12_word(rows[r + WS_HALF], c), c & 3)};
#pragma unroll
            for (int q = 0; q < WS_HALF; q++)
                if (r >= q && r < q + KERNEL_WIDTH) sd2[q] = row_sq_acc2(sd2[q], g, avg2[q]);
        }
#pragma unroll
        for (int q = 0; q < WS_PX; q++) {
            if (!in[q]) continue;
            const float sd = q < WS_HALF ? sd2[q % WS_HALF].x : sd2[q % WS_HALF].y;
            const float stdev = sqrtf(sd / (float)KERNEL_POINT_COUNT);
            const bool valid = finite_f32(stdev) && !(fabsf(stdev) < min_stdev);
            iout[q] = make_uint2(isum[q] | (valid ? 0x80000000u : 0u), __float_as_uint(stdev));
        }
    }
#pragma unroll
    for (int q = 0; q < WS_PX; q++)
        if (yq + q < row1) istats[(size_t)(yq + q) * w + x] = iout[q];
}

void launch_window_stats_pair(const uint8_t *img_a, uint32_t wa, uint32_t ha, uint2 *istats_a,
                              const uint8_t *img_b, uint32_t wb, uint32_t hb, uint2 *istats_b,
                              uint32_t row0, uint32_t row1, float min_stdev, uint32_t *zero_words, hipStream_t s, uint32_t lds_ballast)
{
    const StatsJob ja{img_a, wa, ha, row0, row1 < ha ? row1 : ha, istats_a};
    const StatsJob jb{img_b, wb, hb, row0, row1 < hb ? row1 : hb, istats_b};
    const uint32_t rows_a = ja.row1 > row0 ? ja.row1 - row0 : 0u, rows_b = jb.row1 > row0 ? jb.row1 - row0 : 0u;
    const uint32_t rows = rows_a > rows_b ? rows_a : rows_b;
    if (rows == 0) {
        if (zero_words) (void)hipMemsetAsync(zero_words, 0, 8 * sizeof(uint32_t), s);
        return;
    }
    dim3 grid(((wa > wb ? wa : wb) + 63) / 64, (rows + WS_ROWS - 1) / WS_ROWS, 2);
    // lds_ballast: dynamic LDS the kernel never touches - it only limits how many of its workgroups a CU holds, so that a
    // launch that runs as filler beside another stream's small kernels leaves them LDS and wave slots
    hipLaunchKernelGGL(window_stats_kernel, grid, dim3(256), lds_ballast, s, ja, jb, min_stdev, zero_words);
}

// ---------------------------------------------------------------------------------------------
// search_range: estimate_search_range (mod.rs:468-540) on the compact previous-level grid.
// One thread per searched pixel; writes start | end << 16, or RANGE_NONE.
//
// Exact simplifications (bit-identical to the reference):
//  * corridor_pos = (scale*m - add) / coeff (mod.rs:508-511) always selects the axis whose
//    coefficient get_epipolar_line set to exactly 1.0 and whose `add` it set to exactly 0.0
//    (mod.rs:397-408: the branch taken fixes |other coeff| <= 1 - 2^-53 < 1), so it equals scale*m.
//  * scale*m = coord_prev << (pk - k) is an integer, so the f64 sum for the mean is exact and can
//    be an integer sum; only the squared-deviation sum needs f64 in the reference's scan order.
// The previous-level cells a 64x4 pixel tile can see (<= 44 x 14 for consecutive levels) are staged
// once in LDS: every cell is visited twice by up to ~100 pixels.
// ---------------------------------------------------------------------------------------------
constexpr int SR_TILE_W = 80, SR_TILE_H = 24; // LDS window of previous-level cells per workgroup
constexpr int SR_WIN = 11;                    // cells per axis of one pixel's window, consecutive levels
// sum path: block m's window is the SRF_TAPS x SRF_TAPS cells m-5 .. m+4 (see search_range_kernel); zero-padded
// cell window of a 64 x 4 block tile: columns bx0-5 .. bx0+67, rows by0-5 .. by0+7
constexpr int SRF_TAPS = 10, SRF_LEFT = 5;
constexpr int SRF_W = 64 + SRF_TAPS - 1, SRF_H = 4 + SRF_TAPS - 1, SRF_PITCH = SRF_W;
static_assert(SRF_PITCH == SRF_W, "the staging loop writes cell u at word u");

__device__ __forceinline__ void neighbor_window(const CorrParams &p, uint32_t x, uint32_t y, uint32_t &xs0,
                                                uint32_t &xs1, uint32_t &ys0, uint32_t &ys1)
{
    // mod.rs:481-491, window in FULL-RES cells, then the occupied (previous-level) cells inside it:
    // full-res X = x' << pk with x' < pw
    // v / scale in f32 with scale = 2^-k and v < 2^24: exact, equal to v * 2^k
    const float up = (float)(1u << p.k);
    uint32_t x_min, x_max, y_min, y_max;
    if (p.k <= 12u && p.w1 < 0x10000u && p.h1 < 0x10000u && x < 0x10000u && y < 0x10000u) {
        // integers below 2^17 times 2^k <= 2^12: the f32 products are exact integers below 2^29, floor and ceil leave them
        // alone and the casts do not saturate - the same values without the conversions (uniform test but for x, y, which
        // only exceed 2^16 on the wrapped column -1 of block 0; that pixel is rejected before its window is looked at)
        x_min = sat_sub_u32(x, NEIGHBOR_DISTANCE) << p.k;
        x_max = (x + NEIGHBOR_DISTANCE) << p.k;
        y_min = sat_sub_u32(y, NEIGHBOR_DISTANCE) << p.k;
        y_max = (y + NEIGHBOR_DISTANCE) << p.k;
    } else {
        x_min = f32_to_u32_sat(floorf((float)sat_sub_u32(x, NEIGHBOR_DISTANCE) * up));
        x_max = f32_to_u32_sat(ceilf((float)(x + NEIGHBOR_DISTANCE) * up));
        y_min = f32_to_u32_sat(floorf((float)sat_sub_u32(y, NEIGHBOR_DISTANCE) * up));
        y_max = f32_to_u32_sat(ceilf((float)(y + NEIGHBOR_DISTANCE) * up));
    }
    x_min = min(x_min, p.gw);
    x_max = min(x_max, p.gw);
    y_min = min(y_min, p.gh);
    y_max = min(y_max, p.gh);
    const uint32_t step = 1u << p.pk;
    xs0 = (x_min + step - 1) >> p.pk;
    xs1 = min((x_max + step - 1) >> p.pk, p.pw);
    ys0 = (y_min + step - 1) >> p.pk;
    ys1 = min((y_max + step - 1) >> p.pk, p.ph);
}

// One thread per 2x2 block of searched pixels {2bx-1, 2bx} x {2by-1, 2by}: for consecutive levels
// (pk == k + 1) those four pixels see exactly the same previous-level cells —
//   ceil((2m-1-10)/2) = ceil((2m-10)/2) = m-5   and   ceil((2m-1+10)/2) = ceil((2m+10)/2) = m+5,
// clamping included — so the neighbour statistics are computed once and shared; each pixel then applies
// its own validity tests and corridor bounds.  (Pixels of one block on different corridor axes, possible
// only for perspective geometry, each get their own axis' statistics.)
__device__ __forceinline__ void search_range_body(const CorrParams &p, const uint32_t *__restrict__ prev,
                                                  uint32_t *__restrict__ range, int mode)
{
    // chain path: raw cells of the tile's window (+ slack for the predicated row reads);
    // sum path: the same words hold FA | FB | HA | HB (see below)
    __shared__ uint32_t lds[2 * SRF_H * SRF_PITCH + 2 * SRF_H * 64];
    __shared__ uint32_t axis_vote;
    static_assert(2 * SRF_H * SRF_PITCH + 2 * SRF_H * 64 >= SR_TILE_W * SR_TILE_H + SR_WIN, "LDS plan");
    uint32_t *const cells = lds;
    // tile of 64 x 4 blocks = pixels [128*bx0 - 1, 128*bx0 + 127] x [by_first*2 - 1, ...]
    const TileId tid = xcd_tile();
    const uint32_t bxi = tid.x * 64 + (threadIdx.x & 63);
    const uint32_t byi = (p.row0 >> 1) + tid.y * 4 + (threadIdx.x >> 6);
    const uint32_t tile_x0 = tid.x * 128, tile_y0 = ((p.row0 >> 1) + tid.y * 4) * 2;
    if (threadIdx.x == 0) axis_vote = 0u;

    // the (up to) four pixels of this block and their per-pixel tests (mod.rs:334-345)
    uint32_t px[4], py[4];
    bool valid[4];
    uint32_t cend[4];
    int axis[4]; // 1: corridor position = y of the previous match (the (1, 0)-offset branch), 0: x
    bool any_valid = false, shared = p.pk == p.k + 1;
    // Affine F (first two columns zero): F*p = (F02, F12, f2(p)) for every pixel - the products with the zero
    // entries are exact zeros - so the line's direction, axis and corridor end are the same everywhere and only
    // the offset -scale*f2/f{0|1} differs.  It is finite (mod.rs:338-345) whenever |f2| and the divisor are far
    // from the f64 range limits; then the two f64 divisions per pixel are skipped.  Anything else takes the
    // per-pixel path below.
    // (CorrParams::range_quick, evaluated on the host: the direction constants are finite - p.affine - the divisor is far
    // from the range limits, and |F20| x / scale + |F21| y / scale + |F22| stays below 1e149 over the whole image, so that
    // no pixel's f2 needs looking at)
    const bool quick = p.range_quick != 0;
    Line e0;
    e0.cx = p.affine == 1 ? p.aff_c : 1.0;
    e0.cy = p.affine == 1 ? 1.0 : p.aff_c;
    e0.ax = e0.ay = 0.0; // (the offset is not used here)
    e0.ox = p.affine == 1 ? 1 : 0;
    e0.oy = p.affine == 1 ? 0 : 1;
    // sum path (below) of an affine F: its cell loads do not depend on the per-pixel tests, so they are issued first
    // and the two round trips to memory overlap
    const bool params_ok = p.min_range >= 0.0 && p.min_range < 1e9 && p.extend_range >= 0.0 && p.extend_range < 1e9;
    // positions of the previous level are < max(w2, h2) / 2 + 1 <= 4097: 100 squares fit 32 bits, 100 values 22 bits
    const bool sums_ok = mode != 1 && shared && params_ok && p.w2 <= 8192u && p.h2 <= 8192u;
    const int tile_bx0 = (int)(tid.x * 64), tile_by0 = (int)((p.row0 >> 1) + tid.y * 4);
    constexpr int SRF_PER_THREAD = (SRF_H * SRF_W + 255) / 256;
    auto load_cell = [&](uint32_t u) -> uint32_t {
        const uint32_t r = u / SRF_W, c = u - r * SRF_W;
        const int gx = tile_bx0 - SRF_LEFT + (int)c, gy = tile_by0 - SRF_LEFT + (int)r;
        const bool in = u < (uint32_t)(SRF_H * SRF_W) && gx >= 0 && gy >= 0 && (uint32_t)gx < p.pw && (uint32_t)gy < p.ph;
        return in ? prev[(size_t)gy * p.pw + (uint32_t)gx] : CELL_NONE;
    };
    uint32_t early[SRF_PER_THREAD] = {};
    if (quick && sums_ok) {
#pragma unroll
        for (int i = 0; i < SRF_PER_THREAD; i++) early[i] = load_cell(threadIdx.x + 256u * i);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        px[q] = 2 * bxi + (q & 1) - 1; // wraps to 0xFFFFFFFF for bxi == 0, q even: rejected below
        py[q] = 2 * byi + (q >> 1) - 1;
        valid[q] = false;
        cend[q] = 0;
        axis[q] = 0;
        const uint32_t x = px[q], y = py[q];
        if (x >= p.w1 || y >= p.row1 || y < p.row0 || y >= p.h1) continue;
        const bool interior = x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < p.w1 && y + KERNEL_SIZE < p.h1;
        if (!interior) continue;
        // (mod.rs:334, the searched pixel's stdev test, is not repeated here: every search kernel applies it in
        // pixel_setup BEFORE it reads this pixel's interval, so the interval of a rejected pixel is never looked at -
        // and not reading the statistics word saves 8 B per pixel of HBM traffic)
        Line e = e0;
        if (!quick) {
            e = epipolar_line(p, x, y);
            if (!line_finite(e)) continue;
        }
        valid[q] = true;
        cend[q] = corridor_end_of(p, e);
        axis[q] = e.ox == 1 ? 1 : 0;
        any_valid = true;
    }

    // Which corridor axes do the valid pixels of this workgroup use?  None: nothing to scan.  Exactly one (always,
    // for affine F) and consecutive levels: the windows of neighbouring blocks overlap in all but one column / row, and
    // count, sum and sum of squares of the corridor positions are integers, so they are taken as separable 10-tap box
    // sums over the tile (2 x 10 LDS reads per block instead of 2 x 100) and the f64 chain of mod.rs:515-529 is only
    // run where the integer result leaves the rounding of mod.rs:532 open (sum_stats below).
    // (Affine F: F*p has the same first two components for every pixel, so the axis is e0's and no vote is needed;
    // a tile without valid pixels then stages cells for nothing.)
    uint32_t vote = e0.ox == 1 ? 2u : 1u;
    if (!quick) {
        uint32_t my = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) my |= valid[q] ? 1u << axis[q] : 0u;
        const uint32_t wave_axes = (__ballot(my & 1u) ? 1u : 0u) | (__ballot(my & 2u) ? 2u : 0u);
        __syncthreads(); // axis_vote = 0 is visible
        if ((threadIdx.x & 63) == 0 && wave_axes) atomicOr(&axis_vote, wave_axes);
        __syncthreads();
        vote = axis_vote;
    }
    const bool sums = sums_ok && (vote == 1u || vote == 2u);
    uint32_t *const FA = lds, *const FB = lds + SRF_H * SRF_PITCH, *const HA = lds + 2 * SRF_H * SRF_PITCH,
                    *const HB = HA + SRF_H * 64;
    bool staged = false;
    uint32_t tx0 = 0, ty0 = 0;
    uint32_t win_a = 0, win_b = 0; // this block's window: count << 22 | sum of positions, sum of squared positions
    if (vote == 0u) {
        // no valid pixel in the tile
    } else if (sums) {
        const uint32_t ash = vote == 2u ? 16u : 0u;
#pragma unroll
        for (int i = 0; i < SRF_PER_THREAD; i++) {
            const uint32_t u = threadIdx.x + 256u * i;
            const uint32_t cell = quick ? early[i] : load_cell(u);
            const uint32_t v = (cell >> ash) & 0xFFFFu;
            if (u < (uint32_t)(SRF_H * SRF_W)) {
                FA[u] = cell != CELL_NONE ? (1u << 22) | v : 0u; // SRF_PITCH == SRF_W
                FB[u] = cell != CELL_NONE ? v * v : 0u;
            }
        }
        __syncthreads();
        {
            // rows first: column t of FA (waves 0 and 1) or of FB (waves 2 and 3), its four 10-row sums by sliding - 15
            // additions for the column - then 10 taps along the row per block (integer sums: any order)
            const uint32_t wv = threadIdx.x >> 6, col = (wv & 1u) * 64u + (threadIdx.x & 63u);
            if (col < (uint32_t)SRF_W) {
                const uint32_t *src = (wv & 2u) ? FB : FA;
                uint32_t *dst = (wv & 2u) ? HB : HA;
                uint32_t v[SRF_H];
#pragma unroll
                for (int r = 0; r < SRF_H; r++) v[r] = src[r * SRF_PITCH + col];
                uint32_t sum = 0u;
#pragma unroll
                for (int r = 0; r < SRF_TAPS; r++) sum += v[r];
                dst[col] = sum;
#pragma unroll
                for (int r = 1; r < SRF_H - SRF_TAPS + 1; r++) {
                    sum += v[r + SRF_TAPS - 1] - v[r - 1];
                    dst[r * SRF_PITCH + col] = sum;
                }
            }
        }
        __syncthreads();
        const uint32_t r0 = threadIdx.x >> 6, c = threadIdx.x & 63u;
#pragma unroll
        for (int j = 0; j < SRF_TAPS; j++) {
            win_a += HA[r0 * SRF_PITCH + c + j];
            win_b += HB[r0 * SRF_PITCH + c + j];
        }
    } else {
        // window of the whole tile = union of its corner pixels' windows (the bounds are monotone in x, y)
        uint32_t tx1, ty1, ux0, ux1, uy0, uy1;
        neighbor_window(p, sat_sub_u32(tile_x0, 1), sat_sub_u32(tile_y0, 1), tx0, ux1, ty0, uy1);
        neighbor_window(p, min(tile_x0 + 126, p.w1 - 1), min(tile_y0 + 6, p.h1 - 1), ux0, tx1, uy0, ty1);
        const uint32_t tw = tx1 > tx0 ? tx1 - tx0 : 0u, th = ty1 > ty0 ? ty1 - ty0 : 0u;
        staged = tw <= (uint32_t)SR_TILE_W && th <= (uint32_t)SR_TILE_H;
        if (staged) {
            for (uint32_t u = threadIdx.x; u < tw * th; u += 256) {
                const uint32_t r = u / tw, c = u - r * tw;
                cells[r * SR_TILE_W + c] = prev[(size_t)(ty0 + r) * p.pw + (tx0 + c)];
            }
        }
        __syncthreads();
    }

    // neighbour statistics per corridor axis; res[a] = {center, length} or invalid
    bool have[2] = {false, false};
    uint32_t center[2] = {0, 0}, length[2] = {0, 0};
    auto block_stats = [&](uint32_t x, uint32_t y, int a) {
        uint32_t xs0, xs1, ys0, ys1;
        neighbor_window(p, x, y, xs0, xs1, ys0, ys1);
        const uint32_t ash = a ? 16u : 0u;
        const uint32_t up = p.pk - p.k; // corridor_pos = coordinate << up, an integer
        auto cell_at = [&](uint32_t xx, uint32_t yy) -> uint32_t {
            return staged ? cells[(yy - ty0) * SR_TILE_W + (xx - tx0)] : prev[(size_t)yy * p.pw + xx];
        };
        unsigned long long isum = 0;
        uint32_t neighbor_count = 0;
        double range_stdev = 0.0, mid_corridor = 0.0;
        const bool small = staged && xs1 - xs0 <= (uint32_t)SR_WIN && xs1 > xs0;
        if (small) {
            // unrolled, predicated rows: all LDS offsets immediate, no per-cell loop control; still row-major
            const uint32_t nx = xs1 - xs0;
            for (uint32_t yy = ys0; yy < ys1; yy++) {
                const uint32_t *row = &cells[(yy - ty0) * SR_TILE_W + (xs0 - tx0)];
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)SR_WIN; j++) {
                    const uint32_t cell = row[j];
                    const bool ok = j < nx && cell != CELL_NONE;
                    neighbor_count += ok ? 1u : 0u;
                    isum += ok ? (unsigned long long)(((cell >> ash) & 0xFFFFu) << up) : 0ull;
                }
            }
            if (neighbor_count != 0) {
                mid_corridor = (double)isum / (double)neighbor_count; // exact sum, one rounding
                for (uint32_t yy = ys0; yy < ys1; yy++) {
                    const uint32_t *row = &cells[(yy - ty0) * SR_TILE_W + (xs0 - tx0)];
#pragma unroll
                    for (uint32_t j = 0; j < (uint32_t)SR_WIN; j++) {
                        const uint32_t cell = row[j];
                        const bool ok = j < nx && cell != CELL_NONE;
                        const double delta = (double)(((cell >> ash) & 0xFFFFu) << up) - mid_corridor;
                        const double next = range_stdev + delta * delta;
                        range_stdev = ok ? next : range_stdev;
                    }
                }
            }
        } else {
            for (uint32_t yy = ys0; yy < ys1; yy++)
                for (uint32_t xx = xs0; xx < xs1; xx++) {
                    const uint32_t cell = cell_at(xx, yy);
                    if (cell == CELL_NONE) continue;
                    neighbor_count += 1;
                    isum += (unsigned long long)(((cell >> ash) & 0xFFFFu) << up);
                }
            if (neighbor_count != 0) {
                mid_corridor = (double)isum / (double)neighbor_count;
                for (uint32_t yy = ys0; yy < ys1; yy++)
                    for (uint32_t xx = xs0; xx < xs1; xx++) {
                        const uint32_t cell = cell_at(xx, yy);
                        if (cell == CELL_NONE) continue;
                        const double delta = (double)(((cell >> ash) & 0xFFFFu) << up) - mid_corridor;
                        range_stdev += delta * delta;
                    }
            }
        }
        have[a] = neighbor_count != 0;
        if (have[a]) {
            range_stdev = sqrt(range_stdev / (double)neighbor_count);
            center[a] = f64_to_u32_sat(round(mid_corridor));
            length[a] = f64_to_u32_sat(round(p.min_range + range_stdev * p.extend_range));
        }
    };
    // The same statistics from the integer box sums.  With n cells at positions c_i = v_i << up:
    //   mean = fl(sum c_i / n) as in the chain;   T = n * sum v_i^2 - (sum v_i)^2 = n * sum (v_i - mean_v)^2, exact.
    // The chain's sum S of rounded squared deviations from the rounded mean satisfies S = (4^up T / n)(1 + e),
    // |e| <= (n + 2) 2^-53 (sum of non-negative terms; the rounded mean adds n (mean 2^-53)^2, far below that because
    // T >= n - 1 when T != 0), so sqrt(S / n), times extend_range, plus min_range is within 67 * 2^-53 relative of
    // L = min_range + extend_range * 2^up sqrt(T) / n, and L as evaluated here within 5 * 2^-53.  round() of the two
    // agrees unless L is within 2^-45 (L + 1) of a half-integer (exactly on it, in practice: 2 sqrt(T) / n is an
    // integer for ~0.2 % of the windows): those blocks run the chain over the staged tile; any whose window is not
    // the 10 x 10 cells m-5 .. m+4 runs it over global memory.  T == 0 (all positions equal): the chain's deviations are exact zeros.
    auto sum_stats = [&](uint32_t x, uint32_t y, int a) -> bool {
        uint32_t xs0, xs1, ys0, ys1;
        neighbor_window(p, x, y, xs0, xs1, ys0, ys1);
        if (xs0 != sat_sub_u32(bxi, SRF_LEFT) || xs1 != min(bxi + SRF_TAPS - SRF_LEFT, p.pw) ||
            ys0 != sat_sub_u32(byi, SRF_LEFT) || ys1 != min(byi + SRF_TAPS - SRF_LEFT, p.ph))
            return false;
        // test hooks: mode 2 sends every third block of a sum tile through the chain over the staged tile,
        // mode 3 through the chain over global memory
        if (mode == 3 && ((bxi + byi) % 3u) == 0u) return false;
        const bool force_chain = mode == 2 && ((bxi + byi) % 3u) == 0u;
        const uint32_t n = win_a >> 22, sv = win_a & 0x3FFFFFu, up = p.pk - p.k;
        if (n == 0u) {
            have[a] = false;
            return true;
        }
        const double mid_corridor = (double)((unsigned long long)sv << up) / (double)n;
        const unsigned long long T = (unsigned long long)n * win_b - (unsigned long long)sv * sv;
        uint32_t len;
        if (T == 0ull) {
            len = f64_to_u32_sat(round(p.min_range + 0.0 * p.extend_range));
        } else {
            const double v = sqrt((double)T) * (double)(1u << up) / (double)n;
            const double L = p.min_range + v * p.extend_range;
            const double fl = floor(L), fr = L - fl;
            if (force_chain || !(L < 4.0e9) || fabs(fr - 0.5) <= (L + 1.0) * 0x1p-45) {
                // open: the reference's chain (mod.rs:523-529) over this block's window in its row-major order, from
                // the staged tile (FA is 0 for None)
                // ... unless the mean is exact (n times it gives the integer sum back: a dyadic rational with at most 6
                // fractional bits, n <= 100).  Then every deviation (< 2^14, 6 fractional bits), its square (40 bits) and
                // every partial sum (< 2^35 in units of 2^-12) is exact, and the chain's sum IS 4^up T / n, itself such a
                // number, so this quotient is exact too.  (The usual open window - half of the cells at v, half at v + 1,
                // L = 3.5 exactly - is of this kind; a wave with one such lane used to walk all 100 cells for it.)
                const double nd = (double)n;
                double range_stdev = 0.0;
                if (!force_chain && __builtin_fma(mid_corridor, nd, -(double)((unsigned long long)sv << up)) == 0.0) {
                    range_stdev = (double)(T << (2u * up)) / nd;
                } else {
                    const uint32_t r0 = threadIdx.x >> 6, c0 = threadIdx.x & 63u;
                    for (uint32_t r = 0; r < (uint32_t)SRF_TAPS; r++) {
                        const uint32_t *row = &FA[(r0 + r) * SRF_PITCH + c0];
#pragma unroll
                        for (int j = 0; j < SRF_TAPS; j++) {
                            const uint32_t w = row[j];
                            const double delta = (double)((w & 0x3FFFFFu) << up) - mid_corridor;
                            const double next = range_stdev + delta * delta;
                            range_stdev = w != 0u ? next : range_stdev;
                        }
                    }
                }
                range_stdev = sqrt(range_stdev / (double)n);
                len = f64_to_u32_sat(round(p.min_range + range_stdev * p.extend_range));
            } else {
                len = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);
            }
        }
        have[a] = true;
        center[a] = f64_to_u32_sat(round(mid_corridor));
        length[a] = len;
        return true;
    };
    auto finish = [&](int q, int a) -> uint32_t { // mod.rs:530-539 for pixel q with axis a's statistics
        if (!have[a]) return RANGE_NONE;
        const uint32_t corridor_start = KERNEL_SIZE, corridor_end = cend[q];
        uint32_t s0 = sat_sub_u32(center[a], length[a]);
        s0 = s0 < corridor_start ? corridor_start : (s0 > corridor_end ? corridor_end : s0);
        const uint64_t s1w = (uint64_t)center[a] + (uint64_t)length[a]; // saturating_add
        uint32_t s1 = s1w > (uint64_t)corridor_end ? corridor_end : (uint32_t)s1w;
        s1 = s1 < s0 ? s0 : s1;
        return s0 | (s1 << 16);
    };

    uint32_t out[4] = {RANGE_NONE, RANGE_NONE, RANGE_NONE, RANGE_NONE};
    if (any_valid) {
        if (shared) {
            // one scan per axis in use, on behalf of all valid pixels of the block
#pragma unroll
            for (int a = 0; a < 2; a++) {
                int rep = -1;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (valid[q] && axis[q] == a && rep < 0) rep = q;
                if (rep >= 0) {
                    const uint32_t rx = rep == 0 ? px[0] : (rep == 1 ? px[1] : (rep == 2 ? px[2] : px[3]));
                    const uint32_t ry = rep == 0 ? py[0] : (rep == 1 ? py[1] : (rep == 2 ? py[2] : py[3]));
                    if (!(sums && sum_stats(rx, ry, a))) block_stats(rx, ry, a);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (valid[q]) out[q] = finish(q, axis[q]);
        } else {
            // non-consecutive levels: windows differ inside the block, every pixel scans for itself
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (valid[q]) {
                    block_stats(px[q], py[q], axis[q]);
                    out[q] = finish(q, axis[q]);
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t x = px[q], y = py[q];
        if (x < p.w1 && y >= p.row0 && y < p.row1 && y < p.h1) range[(size_t)y * p.w1 + x] = out[q];
    }
}

// One launch serves the search passes of both directions of a level (blockIdx.z picks the job): they are independent
// (each reads its own direction's previous grid and writes its own buffers), and on the small levels neither fills
// the GPU alone.  The grid covers the larger of the two; workgroups beyond a job's image find nothing to do.
__global__ __launch_bounds__(256) void search_range_kernel(SearchJob ja, SearchJob jb, int mode)
{
    const SearchJob &j = this_job();
    search_range_body(j.p, j.prev, j.range, mode);
}

static bool job_active(const SearchJob &j) { return j.p.row1 > j.p.row0; }

void launch_search_range(const SearchJob *jobs, int n, int mode, hipStream_t s)
{
    uint32_t gx = 0, gy = 0;
    for (int i = 0; i < n; i++) {
        const CorrParams &p = jobs[i].p;
        if (!job_active(jobs[i])) continue;
        // blocks by = row0/2 .. row1/2 cover pixel rows 2by-1, 2by; bx = 0 .. w1/2 cover columns 2bx-1, 2bx
        const uint32_t nby = (p.row1 >> 1) - (p.row0 >> 1) + 1, nbx = (p.w1 >> 1) + 1;
        gx = std::max(gx, (nbx + 63) / 64);
        gy = std::max(gy, (nby + 3) / 4);
    }
    if (!gx || !gy) return;
    hipLaunchKernelGGL(search_range_kernel, dim3(gx, gy, (unsigned)n), dim3(256), 0, s, jobs[0], jobs[n - 1], mode);
}

// ---------------------------------------------------------------------------------------------
// search: correlate_point + correlate_corridor_area (mod.rs:321-384, 411-466).
// One thread per searched pixel; the 121 window deltas live in registers; for every candidate
// the 121-term sum is a serial f32 chain in row-major order (mul, then add).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void search_kernel(CorrParams p, const uint8_t *__restrict__ img1,
                                                      const uint8_t *__restrict__ img2,
                                                      const uint2 *__restrict__ stats1,
                                                      const uint2 *__restrict__ stats2,
                                                      const uint32_t *__restrict__ range,
                                                      uint32_t *__restrict__ out, float *__restrict__ out_score,
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
        const float2 st1 = stats_of(stats1[(size_t)y * p.w1 + x]);
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
                    const float2 st2 = stats_of(stats2[(size_t)y2 * p.w2 + x2]);
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
    if (in_image) {
        out[(size_t)y * p.w1 + x] = best_xy;
        out_score[(size_t)y * p.w1 + x] = best_corr;
    }
    if (cand_counter) {
        // wave-level sum, one atomic per wave
        uint32_t v = evaluated;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_down(v, sft, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(cand_counter, (unsigned long long)v);
    }
}

void launch_search(const CorrParams &p, const uint8_t *img1, const uint8_t *img2, const uint2 *stats1,
                   const uint2 *stats2, const uint32_t *range, uint32_t *out, float *out_score,
                   unsigned long long *cand_counter, hipStream_t s)
{
    if (p.row1 <= p.row0) return;
    dim3 grid((p.w1 + 63) / 64, (p.row1 - p.row0 + 3) / 4);
    hipLaunchKernelGGL(search_kernel, grid, dim3(256), 0, s, p, img1, img2, stats1, stats2, range, out, out_score, cand_counter);
}

// ---------------------------------------------------------------------------------------------
// search2: the same correlate_point / correlate_corridor_area (mod.rs:321-384, 411-466) with an
// exact-integer pre-filter.
//
// For every candidate c of a searched pixel the reference computes
//     f(c) = fl( S_f / fl(fl(sd1*sd2)*121) ),   S_f = serial f32 sum of fl(d1_k * d2_k)      (mod.rs:442-454)
// where S_f approximates  sum (a_k - mean1)(b_k - mean2) = N / 121  with the EXACT integer
//     N = 121*S12 - s1*s2,   S12 = sum a_k*b_k,   s = window sum          (all < 2^31 for u8 images).
// S12 is formed exactly with v_dot4_u32_u8 (4 multiply-adds per lane-op instead of 3 ops per term) and
//     g(c) = (float)N / (121 * 121 * sd1 * sd2)
// uses the reference's OWN f32 stdevs, so the denominator's rounding cancels and only the numerator's
// matters.  Error bound |f(c) - g(c)| <= DELTA (derivation in DESIGN.md §4), in units of u = 2^-24:
//   deltas d = fl(a - avg): |rounding| <= 2^-17 = 128u each; the error of avg itself enters only at
//     second order because the true deviations sum to zero  ->  128u*(1/sd1 + 1/sd2) <= 256u  (sd >= 1)
//   121 rounded products + 121 serial additions (gamma_122)                      <= 122u
//   final divide, denominator products, the filter's own f32 steps               <= 15u
// total < 393u = 2.35e-5.
//
// Decision rule: the reference keeps the FIRST candidate (in stripe, i order) that maximises f
// among those with f >= threshold.  Such a candidate c* satisfies g(c*) >= max g - 2*DELTA and
// g(c*) >= threshold - DELTA, and so does every candidate that ties with it.  The filter therefore
// records, in order, the (few) candidates inside that band, and only those are re-evaluated with the
// reference's exact serial f32 arithmetic; the winner among them under the reference's own rule
// (mod.rs:456-457) is the reference's winner, with the reference's exact score.  Pixels whose band
// holds more than S2_K candidates re-evaluate their whole corridor exactly.
//
// Memory plan: one 256-thread workgroup = a 64x4 tile of searched pixels.  The bounding box of
// all 11x11 candidate windows of the tile is staged ONCE from HBM into LDS as eight byte-shifted
// copies, so that any window row (11 bytes at an arbitrary byte address) is one aligned
// ds_read_b64 + ds_read_b32 from the copy selected by (address & 7); copies are 32 B (mod 256 B)
// apart, so 64 lanes reading consecutive candidates hit 64 distinct bank pairs.
// ---------------------------------------------------------------------------------------------
constexpr int S2_K = 4;                 // contenders kept per pixel before falling back to the full corridor
constexpr int S2_LDS_BYTES = 40448;     // 8 tile copies + candidate statistics; 3 workgroups per CU
constexpr int S2_LDS_BYTES_STEEP = 65024; // steep / column-major / per-pixel lines: taller candidate boxes, 2 workgroups per CU
constexpr float S2_DELTA = 2.5e-5f;     // >= 393 * 2^-24 (see above)

// Contender word written by the filter kernel, one u64 per searched pixel:
//   bits [0, 60): up to four 15-bit candidate codes (stripe index << 11 | i - r0), oldest first
//   bits [60, 63): count 0..4; CW_WHOLE = re-evaluate the whole corridor exactly
constexpr unsigned long long CW_WHOLE = 5ull;
constexpr int S3_OUT_PX = 53; // pixels per box-kernel wave (see search3_box_kernel)
constexpr unsigned long long CW_FALLBACK = 6ull; // search3_box_kernel -> search3_fallback_kernel

// Tile work lists between the kernels of one search pass (search version 3): the box kernel appends the tiles it
// declined (for the candidate filter) and both filters append the tiles that hold a CW_WHOLE pixel (for the
// whole-corridor kernel); the consumers run as small persistent grids over the lists instead of dispatching a
// workgroup per image tile just to find nothing to do.  Entry: x0 | row-tile index << 16 | (64-wide tile) << 31;
// a box tile is 53 pixels wide.
__device__ __forceinline__ void worklist_push(WorkList wl, uint32_t entry)
{
    if (wl.count) wl.items[atomicAdd(wl.count, 1u)] = entry;
}
// A 256-thread tile of searched pixels for the per-candidate kernels: lanes along x and the four waves on four
// rows (nl <= 64 pixels wide) - or, for the transposed box kernel's tiles, lanes along y and the waves on four
// columns (nl pixels tall).  Work-list entry: x0 | tile index along y << 16 | transposed << 30 | 64-wide << 31.
struct PixTile {
    uint32_t x0, y0, nl;
    bool tr;
};
__device__ __forceinline__ PixTile tile_of_entry(uint32_t entry, uint32_t row0)
{
    PixTile t;
    // (bits 31:30 = 11: a row-major tile 58 pixels wide - half of a two-columns-per-lane box tile, search3_box2_kernel)
    const bool half_pair = (entry >> 30) == 3u;
    t.tr = ((entry >> 30) & 1u) != 0u && !half_pair;
    t.x0 = entry & 0xFFFFu;
    const uint32_t idx = (entry >> 16) & 0x3FFFu;
    t.y0 = row0 + idx * (t.tr ? (uint32_t)S3_OUT_PX : 4u);
    t.nl = t.tr ? (uint32_t)S3_OUT_PX : (half_pair ? 58u : ((entry >> 31) ? 64u : (uint32_t)S3_OUT_PX));
    return t;
}
__device__ __forceinline__ bool tile_pixel(const PixTile &t, uint32_t &x, uint32_t &y)
{
    const uint32_t l = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    x = t.x0 + (t.tr ? wv : l);
    y = t.y0 + (t.tr ? l : wv);
    return l < t.nl;
}
constexpr uint32_t CW_MAX_LEN = 2048u;  // i - r0 must fit in 11 bits
constexpr uint32_t S2_FIXED_PITCH = 128u;

struct CandXY {
    uint32_t x, y;
};
__device__ __forceinline__ CandXY candidate_xy(const Line &e, uint32_t i, int off)
{
    // mod.rs:424-429
    const double x2d = (e.cx * (double)i + e.ax) + (double)(off * e.ox);
    const double y2d = (e.cy * (double)i + e.ay) + (double)(off * e.oy);
    CandXY c;
    c.x = f64_to_u32_sat(floor(x2d));
    c.y = f64_to_u32_sat(floor(y2d));
    return c;
}
__device__ __forceinline__ bool candidate_in_bounds(const CorrParams &p, const CandXY &c)
{
    return !(c.x < KERNEL_SIZE || c.x >= p.w2 - KERNEL_SIZE || c.y < KERNEL_SIZE || c.y >= p.h2 - KERNEL_SIZE);
}

// Per-pixel setup shared by both kernels: mod.rs:321-364.  Returns false when the pixel yields None
// without looking at any candidate.
struct PixelSetup {
    float2 st1;
    Line e;
    uint32_t r0, r1;
};
__device__ __forceinline__ bool pixel_setup(const CorrParams &p, uint32_t x, uint32_t y,
                                            const uint2 *__restrict__ stats1, const uint32_t *__restrict__ range,
                                            PixelSetup &ps)
{
    if (!(x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < p.w1 && y + KERNEL_SIZE < p.h1)) return false;
    ps.st1 = stats_of(stats1[(size_t)y * p.w1 + x]);
    ps.e = epipolar_line(p, x, y);
    if (!(finite_f32(ps.st1.y) && !(fabsf(ps.st1.y) < p.min_stdev) && line_finite(ps.e))) return false; // :334-345
    ps.r0 = KERNEL_SIZE;
    ps.r1 = corridor_end_of(p, ps.e);
    if (!p.first_pass) { // mod.rs:351-364
        const uint32_t rg = range[(size_t)y * p.w1 + x];
        if (rg == RANGE_NONE) return false;
        ps.r0 = rg & 0xFFFFu;
        ps.r1 = rg >> 16;
    }
    return ps.r0 < ps.r1;
}

// a settled pixel's result: the match plane always, the score plane only where the pass's scores are the reference's
// (CorrParams::need_scores - the planes of other passes are never read, complete() reports NaN for them)
__device__ __forceinline__ void store_cell(const CorrParams &p, uint32_t *__restrict__ out, float *__restrict__ out_score, size_t i,
                                           uint2 cell)
{
    out[i] = cell.x;
    if (p.need_scores) out_score[i] = __uint_as_float(cell.y);
}

// ---- kernel A: filter ---------------------------------------------------------------------------------
// One tile: `width` (<= 64) pixels from x0 in the four rows of row tile `ytile`.
// Returns (uniformly for the workgroup) whether the tile holds whole-corridor pixels; only then are its contender
// words written (each lane reads back only its own) and, when a work list is given, the tile is queued on it.
// MODE (the first pass in two launches, search2_filter_split_kernel): 1 = this workgroup walks ONE stripe of the tile's
// corridors (`stripe_sel`) and leaves its contenders in `split`; 2 = no walk: the stripes' contenders are merged and the
// exact phase runs.  0 = everything in one workgroup.
template <bool COUNT, int MODE = 0>
__device__ __forceinline__ bool search2_filter_tile(const CorrParams &p, const uint8_t *__restrict__ img1,
                                                    const uint8_t *__restrict__ img2, const uint2 *__restrict__ stats1,
                                                    const uint2 *__restrict__ istats1, const uint2 *__restrict__ istats2,
                                                    const uint32_t *__restrict__ range,
                                                    unsigned long long *__restrict__ contenders, uint32_t *__restrict__ out,
                                                    float *__restrict__ out_score,
                                                    unsigned long long *__restrict__ counters, int only_fallback,
                                                    PixTile tl, WorkList whole_list, uint8_t *__restrict__ tile,
                                                    uint32_t lds_bytes, int stripe_sel = 0, unsigned long long *split = nullptr,
                                                    uint32_t tile_index = 0, uint32_t n_tiles = 0)
{
    // `tile` is the launch's dynamic LDS (lds_bytes: S2_LDS_BYTES, or S2_LDS_BYTES_STEEP where the host expects
    // tall candidate boxes)
    __shared__ int bb[4]; // min x, min y, max x, max y of in-bounds candidate centres

    const uint32_t lane = threadIdx.x & 63;
    uint32_t x, y;
    const bool in_image = tile_pixel(tl, x, y) && x < p.w1 && y < p.row1;
    if (threadIdx.x == 0) {
        bb[0] = 0x7FFFFFFF;
        bb[1] = 0x7FFFFFFF;
        bb[2] = -1;
        bb[3] = -1;
    }

    PixelSetup ps;
    ps.st1 = make_float2(0.0f, 1.0f);
    ps.e.cx = ps.e.cy = ps.e.ax = ps.e.ay = 0.0;
    ps.e.ox = ps.e.oy = 0;
    ps.r0 = ps.r1 = 0;
    // only_fallback: second stage behind search3_box_kernel, which left the pixels it could not walk as a box
    // marked CW_FALLBACK; everything else is already settled and must not be touched
    bool mine = in_image;
    if (only_fallback) {
        mine = in_image && (uint32_t)(contenders[(size_t)y * p.w1 + x] >> 60) == (uint32_t)CW_FALLBACK;
        if (!__syncthreads_or(mine ? 1 : 0)) return false;
    }
    const bool active = mine && pixel_setup(p, x, y, stats1, range, ps);
    const Line &e = ps.e;
    const uint32_t r0 = ps.r0, r1 = ps.r1;
    const uint32_t len = active ? r1 - r0 : 0u;
    const int cs = p.corridor_size;

    // ---- bounding box of the in-bounds candidate centres (corners of the monotone corridor) -------
    int lminx = 0x7FFFFFFF, lminy = 0x7FFFFFFF, lmaxx = -1, lmaxy = -1;
    if (active) {
        const int xlo = KERNEL_SIZE, xhi = (int)p.w2 - KERNEL_SIZE - 1, ylo = KERNEL_SIZE, yhi = (int)p.h2 - KERNEL_SIZE - 1;
#pragma unroll
        for (int corner = 0; corner < 4; corner++) {
            const CandXY c = candidate_xy(e, (corner & 1) ? r1 - 1 : r0, (corner & 2) ? cs : -cs);
            const int cx = min(max((int)min(c.x, 0x7FFFFFFFu), xlo), xhi);
            const int cy = min(max((int)min(c.y, 0x7FFFFFFFu), ylo), yhi);
            lminx = min(lminx, cx);
            lmaxx = max(lmaxx, cx);
            lminy = min(lminy, cy);
            lmaxy = max(lmaxy, cy);
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        lminx = min(lminx, __shfl_xor(lminx, sft, 64));
        lminy = min(lminy, __shfl_xor(lminy, sft, 64));
        lmaxx = max(lmaxx, __shfl_xor(lmaxx, sft, 64));
        lmaxy = max(lmaxy, __shfl_xor(lmaxy, sft, 64));
    }
    __syncthreads(); // bb initialised
    if (lane == 0 && lmaxx >= 0) {
        atomicMin(&bb[0], lminx);
        atomicMin(&bb[1], lminy);
        atomicMax(&bb[2], lmaxx);
        atomicMax(&bb[3], lmaxy);
    }
    __syncthreads();
    const bool any_active = bb[2] >= 0;
    const int cx0 = bb[0], cy0 = bb[1];
    const int bx0 = cx0 - KERNEL_SIZE, by0 = cy0 - KERNEL_SIZE;
    const int bwid = bb[2] - bb[0] + 2 * KERNEL_SIZE + 2; // + 1 for the 12th byte of a row read
    const int bhei = bb[3] - bb[1] + 2 * KERNEL_SIZE + 1;
    // row pitch, multiple of 8; the common case (box <= 128 B wide) uses a compile-time pitch so that every
    // LDS row offset in the candidate loop is an immediate
    // ... unless only the box's own width (tall narrow boxes: column-major lines) makes the eight copies fit
    const uint32_t cw = any_active ? (uint32_t)(bb[2] - bb[0] + 1) : 0u, ch = any_active ? (uint32_t)(bb[3] - bb[1] + 1) : 0u;
    auto copy_stride = [&](uint32_t pitch) { return ((pitch * (uint32_t)bhei + 16u + 255u) & ~255u) + 32u; };
    const uint32_t Ptight = (uint32_t)((bwid + 7) & ~7);
    const bool fixed_fits = bwid <= (int)S2_FIXED_PITCH && 8u * copy_stride(S2_FIXED_PITCH) + cw * ch * 8u <= lds_bytes;
    const uint32_t P = !any_active ? 8u : (fixed_fits ? S2_FIXED_PITCH : Ptight);
    const uint32_t CS = any_active ? copy_stride(P) : 0u; // copy stride
    // per-candidate statistics (istats2) of every candidate centre in the box, staged next to the tile
    const uint32_t IS_OFF = 8u * CS;
    const bool use_lds = any_active && IS_OFF + cw * ch * 8u <= lds_bytes;
    const uint2 *lds_is = reinterpret_cast<const uint2 *>(tile + IS_OFF);

    // ---- stage the target tile (8 byte-shifted copies) and the candidate statistics -----------------
    if (use_lds) {
        const uint32_t cols4 = P >> 2;
        const uint32_t units = cols4 * (uint32_t)bhei;
        for (uint32_t u = threadIdx.x; u < units; u += 256) {
            const uint32_t row = u / cols4, c4 = (u - row * cols4) << 2;
            const Row12 g = load_row12(img2 + (size_t)(by0 + (int)row) * p.w2 + (size_t)(bx0 + (int)c4));
            const uint32_t o = row * P + c4;
            *reinterpret_cast<uint32_t *>(tile + 0 * CS + o) = g.a;
            *reinterpret_cast<uint32_t *>(tile + 1 * CS + o) = __builtin_amdgcn_alignbyte(g.b, g.a, 1);
            *reinterpret_cast<uint32_t *>(tile + 2 * CS + o) = __builtin_amdgcn_alignbyte(g.b, g.a, 2);
            *reinterpret_cast<uint32_t *>(tile + 3 * CS + o) = __builtin_amdgcn_alignbyte(g.b, g.a, 3);
            *reinterpret_cast<uint32_t *>(tile + 4 * CS + o) = g.b;
            *reinterpret_cast<uint32_t *>(tile + 5 * CS + o) = __builtin_amdgcn_alignbyte(g.c, g.b, 1);
            *reinterpret_cast<uint32_t *>(tile + 6 * CS + o) = __builtin_amdgcn_alignbyte(g.c, g.b, 2);
            *reinterpret_cast<uint32_t *>(tile + 7 * CS + o) = __builtin_amdgcn_alignbyte(g.c, g.b, 3);
        }
        for (uint32_t u = threadIdx.x; u < cw * ch; u += 256) {
            const uint32_t row = u / cw, col = u - row * cw;
            // {window sum, f32 stdev}; candidates the reference skips (stdev non-finite or < min_stdev,
            // mod.rs:437-441) get stdev = +inf so that their acceptance threshold can never be reached
            uint2 v = istats2[(size_t)(cy0 + (int)row) * p.w2 + (size_t)(cx0 + (int)col)];
            v.y = (v.x & 0x80000000u) ? v.y : 0x7F800000u;
            v.x &= 0x7FFFFFFFu;
            *reinterpret_cast<uint2 *>(tile + IS_OFF + 8u * u) = v;
        }
    }
    __syncthreads();

    unsigned long long word = 0ull; // 0 = settled here, CW_WHOLE = left to the exact kernel
    uint2 cell = make_uint2(CELL_NONE, 0x7FC00000u);
    uint32_t evaluated = 0, multi = 0, whole = 0, exact_evals = 0;
    // the filter's result for this pixel: the contender list (walk order), and - SPLIT - the contenders' filter scores
    float runmax = -__builtin_inff();
    unsigned long long clist = 0ull;
    uint32_t count = 0;
    float gl[S2_K] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr bool SPLIT = MODE == 1;
    static_assert(S2_K == 4, "gl[] is written by explicit selects");
    // what follows the walk: overflow to the whole-corridor kernel, or the exact re-evaluation of the contenders
    // (a: the searched window's rows, 12th byte zero)
    const auto settle = [&](const Row12 (&a)[KERNEL_WIDTH]) {

        if (count > (uint32_t)S2_K) {
            word = CW_WHOLE << 60;
            whole = 1;
            evaluated = 0; // the exact kernel walks (and counts) the whole corridor
        } else {
            // ---- exact re-evaluation of the contenders, right here: the tile and the candidate statistics
            // are still in LDS.  mod.rs:442-464 — the reference's serial f32 chain and acceptance rule.
            multi = count > 1 ? 1u : 0u;
            bool have = false;
            float bcorr = 0.0f;
            uint32_t bxy = 0;
            const float avg1 = ps.st1.x, sdev1 = ps.st1.y;
            // positions-only pass (CorrParams::need_scores): a single contender clearly above the threshold IS the
            // match - every other candidate has g < g* - 2 delta, hence f < f*, and f* >= g* - delta >= threshold
            uint32_t ecount = count;
            if (!p.need_scores && count == 1u && runmax >= p.threshold + S2_DELTA) {
                const uint32_t code = (uint32_t)clist & 0x7FFFu;
                const CandXY c = candidate_xy(e, r0 + (code & 0x7FFu), (int)(code >> 11) - cs);
                cell = make_uint2(c.x | (c.y << 16), __float_as_uint(runmax));
                ecount = 0u;
            }
            for (uint32_t j = 0; j < ecount; j++) {
                const uint32_t code = (uint32_t)(clist >> (15u * j)) & 0x7FFFu;
                const CandXY c = candidate_xy(e, r0 + (code & 0x7FFu), (int)(code >> 11) - cs);
                const uint2 is2 = lds_is[(c.y - (uint32_t)cy0) * cw + (c.x - (uint32_t)cx0)];
                const float avg2 = (float)is2.x / (float)KERNEL_POINT_COUNT; // == compute_point_avg (exact sum)
                const float sdev2 = __uint_as_float(is2.y);
                const uint32_t a0 = (uint32_t)((int)c.y - KERNEL_SIZE - by0) * P + (uint32_t)((int)c.x - KERNEL_SIZE - bx0);
                const uint32_t phi = a0 & 7u;
                const uint8_t *src = tile + phi * CS + (a0 - phi);
                float corr = 0.0f;
#pragma unroll
                for (int r = 0; r < KERNEL_WIDTH; r++) {
                    const uint2 lo = *reinterpret_cast<const uint2 *>(src + r * P);
                    const uint32_t hi = *reinterpret_cast<const uint32_t *>(src + r * P + 8);
                    Row12 ar = a[r];
                    // opaque per iteration: keeps the 121 searched-window deltas from being hoisted out of the
                    // contender loop into 121 live registers
                    asm volatile("" : "+v"(ar.a), "+v"(ar.b), "+v"(ar.c));
                    Row12 br;
                    br.a = lo.x;
                    br.b = lo.y;
                    br.c = hi;
                    corr = row_corr_acc(corr, ar, br, avg1, avg2);
                }
                corr /= sdev1 * sdev2 * (float)KERNEL_POINT_COUNT; // mod.rs:454
                exact_evals++;
                if (corr >= p.threshold && (!have || corr > bcorr)) { // mod.rs:456-464
                    have = true;
                    bcorr = corr;
                    bxy = c.x | (c.y << 16);
                }
            }
            if (have) cell = make_uint2(bxy, __float_as_uint(bcorr));
        }
    };
    if (active && (!use_lds || len > CW_MAX_LEN)) {
        word = CW_WHOLE << 60;
        whole = 1;
    } else if (active) {
        // searched window, packed bytes: a[] has the 12th byte zero, ap[] = (0, a0..a10).  A 12-byte target
        // row read at address q then serves BOTH the candidate whose window starts at q (with a[]) and the
        // one starting at q + 1 (with ap[]): two candidates per LDS row read.
        Row12 a[KERNEL_WIDTH], ap[KERNEL_WIDTH];
        {
            const uint8_t *base = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (x - KERNEL_SIZE);
#pragma unroll
            for (int r = 0; r < KERNEL_WIDTH; r++) {
                a[r] = load_row12(base + (size_t)r * p.w1);
                a[r].c &= 0x00FFFFFFu;
                ap[r].a = a[r].a << 8;
                ap[r].b = __builtin_amdgcn_alignbyte(a[r].b, a[r].a, 3);
                ap[r].c = __builtin_amdgcn_alignbyte(a[r].c, a[r].b, 3);
            }
        }
        const int s1 = (int)(istats1[(size_t)y * p.w1 + x].x & 0x7FFFFFFFu);
        const float k1 = ps.st1.y * (float)(KERNEL_POINT_COUNT * KERNEL_POINT_COUNT); // 121*121*sd1
        const float c1 = 1.0f / k1;

        const float thr_lo = p.threshold - S2_DELTA;
        // Acceptance band in the integer domain.  g = N * c1 / sd2 >= lim  <=>  (float)N >= lim * k1 * sd2
        // (k1, sd2 > 0).  limk = lim * k1 shaved by 2^-20 keeps the cheap test free of false negatives
        // against the handful of f32 roundings on either side; a hit is then re-tested with g itself.
        float limk = thr_lo * k1 * (1.0f - 9.5367431640625e-7f);
        auto score = [&](int num, float sd2) -> float { return (float)num * (c1 * __builtin_amdgcn_rcpf(sd2)); };
        // band bookkeeping: candidate `code` (stripe << 11 | i - r0) joins the contender list
        auto record = [&](float g, uint32_t code) {
            const float lim = fmaxf(runmax - 2.0f * S2_DELTA, thr_lo);
            if (g >= lim) {
                if (g > runmax + 2.0f * S2_DELTA) { // everything recorded so far is out of the band
                    count = 0;
                    clist = 0ull;
                }
                runmax = fmaxf(runmax, g);
                limk = fmaxf(runmax - 2.0f * S2_DELTA, thr_lo) * k1 * (1.0f - 9.5367431640625e-7f);
                if (count < (uint32_t)S2_K) clist |= (unsigned long long)code << (15u * count);
                if (SPLIT) { // (the merge re-tests every contender against the tile-wide band)
                    gl[0] = count == 0u ? g : gl[0];
                    gl[1] = count == 1u ? g : gl[1];
                    gl[2] = count == 2u ? g : gl[2];
                    gl[3] = count == 3u ? g : gl[3];
                }
                count = min(count + 1u, (uint32_t)S2_K + 1u);
            }
        };
        const bool major_x = e.ox == 0; // candidates advance along x (x2 == i exactly), stripes shift y

        // One corridor walk, instantiated for the fixed and for the dynamic row pitch.
        auto walk = [&](auto pitch_tag) {
            constexpr uint32_t PF = decltype(pitch_tag)::value;
            const uint32_t Pp = PF ? PF : P;
            for (int off = -cs; off <= cs; off++) {
                if (SPLIT && off != stripe_sel) continue;
                // The minor coordinate is monotone in i; if it is the same at both ends of the interval it
                // is the same everywhere and the per-candidate f64 evaluation can be skipped.
                const CandXY cf = candidate_xy(e, r0, off), cl = candidate_xy(e, r1 - 1, off);
                const bool minor_const = major_x ? (cf.y == cl.y) : (cf.x == cl.x);
                const uint32_t tbase = (uint32_t)(off + cs) << 11;
                if (major_x) {
                    // Candidates advance along x; the row y2 is constant over the stripe or steps occasionally.
                    // Two horizontally adjacent candidates in the same row share their LDS row reads; the two
                    // dot chains are independent and interleaved.
                    const uint32_t i_hi = min(r1, p.w2 - KERNEL_SIZE);
                    uint32_t i = max(r0, (uint32_t)KERNEL_SIZE);
                    while (i < i_hi) {
                        const bool has2 = i + 1 < i_hi;
                        uint32_t ya = cf.y, yb = cf.y;
                        if (!minor_const) {
                            ya = candidate_xy(e, i, off).y;
                            yb = has2 ? candidate_xy(e, i + 1, off).y : ya;
                        }
                        const bool pair = has2 && yb == ya;
                        const uint32_t ti = tbase + (i - r0);
                        const uint32_t icur = i;
                        i += pair ? 2u : 1u;
                        if (ya < KERNEL_SIZE || ya >= p.h2 - KERNEL_SIZE) continue;
                        const uint2 *isrow = lds_is + (ya - (uint32_t)cy0) * cw + (icur - (uint32_t)cx0);
                        const uint2 is0 = isrow[0];
                        uint2 is1 = isrow[pair ? 1 : 0];
                        is1.y = pair ? is1.y : 0x7F800000u; // no second candidate: unreachable threshold
                        const uint32_t a0 =
                            (uint32_t)((int)ya - KERNEL_SIZE - by0) * Pp + icur - (uint32_t)(KERNEL_SIZE + bx0);
                        const uint32_t phi = a0 & 7u;
                        const uint8_t *src = tile + phi * CS + (a0 - phi);
                        // all 22 row reads in flight first, then the two independent dot chains interleaved
                        uint2 lo[KERNEL_WIDTH];
                        uint32_t hi[KERNEL_WIDTH];
#pragma unroll
                        for (int r = 0; r < KERNEL_WIDTH; r++) {
                            lo[r] = *reinterpret_cast<const uint2 *>(src + r * Pp);
                            hi[r] = *reinterpret_cast<const uint32_t *>(src + r * Pp + 8);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        uint32_t s12a = 0, s12b = 0;
#pragma unroll
                        for (int r = 0; r < KERNEL_WIDTH; r++) {
                            s12a = __builtin_amdgcn_udot4(a[r].a, lo[r].x, s12a, false);
                            s12b = __builtin_amdgcn_udot4(ap[r].a, lo[r].x, s12b, false);
                            s12a = __builtin_amdgcn_udot4(a[r].b, lo[r].y, s12a, false);
                            s12b = __builtin_amdgcn_udot4(ap[r].b, lo[r].y, s12b, false);
                            s12a = __builtin_amdgcn_udot4(a[r].c, hi[r], s12a, false);
                            s12b = __builtin_amdgcn_udot4(ap[r].c, hi[r], s12b, false);
                        }
                        // Pin both sums here so the two chains stay interleaved in one block.
                        asm volatile("" : "+v"(s12a), "+v"(s12b));
                        const float sd0 = __uint_as_float(is0.y), sd1 = __uint_as_float(is1.y);
                        if (COUNT) evaluated += (sd0 < __builtin_inff() ? 1u : 0u) + (sd1 < __builtin_inff() ? 1u : 0u);
                        // all factors are below 2^24: full-rate 24-bit multiplies instead of v_mul_lo_u32
                        const int n0 = __mul24((int)s12a, KERNEL_POINT_COUNT) - __mul24(s1, (int)is0.x);
                        const int n1 = __mul24((int)s12b, KERNEL_POINT_COUNT) - __mul24(s1, (int)is1.x);
                        // one rarely-taken branch for the pair: the larger margin (float)N - limk*sd2 decides (the
                        // fused multiply-add only pre-screens - limk is shaved by 2^-20 - and record() re-tests)
                        if (fmaxf(__builtin_fmaf(-limk, sd0, (float)n0), __builtin_fmaf(-limk, sd1, (float)n1)) >= 0.0f) {
                            if (sd0 < __builtin_inff()) record(score(n0, sd0), ti);
                            if (sd1 < __builtin_inff()) record(score(n1, sd1), ti + 1);
                        }
                    }
                } else {
                    for (uint32_t i = r0; i < r1; i++) {
                        CandXY c;
                        if (minor_const) { // column-constant stripe (x2 fixed, y2 == i)
                            c.x = cf.x;
                            c.y = i;
                        } else {
                            c = candidate_xy(e, i, off);
                        }
                        if (!candidate_in_bounds(p, c)) continue;
                        const uint2 is2 = lds_is[(c.y - (uint32_t)cy0) * cw + (c.x - (uint32_t)cx0)];
                        const uint32_t a0 =
                            (uint32_t)((int)c.y - KERNEL_SIZE - by0) * Pp + (uint32_t)((int)c.x - KERNEL_SIZE - bx0);
                        const uint32_t phi = a0 & 7u;
                        const uint8_t *src = tile + phi * CS + (a0 - phi);
                        uint32_t s12 = 0, s12x = 0;
#pragma unroll
                        for (int r = 0; r < KERNEL_WIDTH; r++) {
                            const uint2 lo = *reinterpret_cast<const uint2 *>(src + r * Pp);
                            const uint32_t hi = *reinterpret_cast<const uint32_t *>(src + r * Pp + 8);
                            s12 = __builtin_amdgcn_udot4(a[r].a, lo.x, s12, false);
                            s12x = __builtin_amdgcn_udot4(a[r].b, lo.y, s12x, false);
                            s12 = __builtin_amdgcn_udot4(a[r].c, hi, s12, false);
                        }
                        asm volatile("" : "+v"(s12), "+v"(s12x));
                        const float sd2 = __uint_as_float(is2.y);
                        if (COUNT) evaluated += sd2 < __builtin_inff() ? 1u : 0u;
                        const int num = __mul24((int)(s12 + s12x), KERNEL_POINT_COUNT) - __mul24(s1, (int)is2.x);
                        if ((float)num >= limk * sd2 && sd2 < __builtin_inff()) record(score(num, sd2), tbase + (i - r0));
                    }
                }
            }
        };
        if (MODE != 2) {
            if (P == S2_FIXED_PITCH)
                walk(std::integral_constant<uint32_t, S2_FIXED_PITCH>{});
            else
                walk(std::integral_constant<uint32_t, 0u>{});
        }
        if (MODE == 0) settle(a);
    }
    if constexpr (MODE == 1) {
        // This workgroup's contenders of its stripe, per pixel: {runmax | count << 32, list, scores} - 32 bytes at
        // [(tile * stripes + stripe) * 256 + thread].  The merge is the next launch.
        const uint32_t NS = (uint32_t)(2 * cs + 1), sub = (uint32_t)(stripe_sel + cs);
        unsigned long long *const rec = split + ((size_t)(tile_index * NS + sub) * 256u + threadIdx.x) * 4u;
        rec[0] = (unsigned long long)__float_as_uint(runmax) | ((unsigned long long)count << 32);
        rec[1] = clist;
        rec[2] = (unsigned long long)__float_as_uint(gl[0]) | ((unsigned long long)__float_as_uint(gl[1]) << 32);
        rec[3] = (unsigned long long)__float_as_uint(gl[2]) | ((unsigned long long)__float_as_uint(gl[3]) << 32);
        if (COUNT && counters) { // (the candidates it looked at are counted here)
            uint32_t v0 = evaluated;
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) v0 += __shfl_down(v0, sft, 64);
            if (lane == 0 && v0) atomicAdd(&counters[0], (unsigned long long)v0);
        }
        return false;
    }
    if constexpr (MODE == 2) {
        if (active && !whole) {
            // The tile-wide band: a contender within 2 delta of the best score of ALL stripes was recorded by its own
            // stripe's walk (whose running maximum was never larger) and survived that walk's resets (a reset means a
            // score more than 2 delta above it).  Stripes in corridor order, a stripe's list in walk order: the merged
            // list is in the reference's order, as the one-workgroup walk leaves it.
            const uint32_t NS = (uint32_t)(2 * cs + 1);
            const unsigned long long *const base = split + ((size_t)tile_index * NS * 256u + threadIdx.x) * 4u;
            float rg = -__builtin_inff();
            bool overflow = false;
            for (uint32_t q = 0; q < NS; q++) {
                const unsigned long long w0 = base[(size_t)q * 256u * 4u];
                rg = fmaxf(rg, __uint_as_float((uint32_t)w0));
                overflow = overflow || (uint32_t)(w0 >> 32) > (uint32_t)S2_K;
            }
            const float lim = fmaxf(rg - 2.0f * S2_DELTA, p.threshold - S2_DELTA);
            count = 0;
            clist = 0ull;
            for (uint32_t q = 0; q < NS; q++) {
                const unsigned long long *r4 = base + (size_t)q * 256u * 4u;
                const uint32_t n = min((uint32_t)(r4[0] >> 32), (uint32_t)S2_K);
                const unsigned long long cl = r4[1], g01 = r4[2], g23 = r4[3];
                for (uint32_t j = 0; j < n; j++) {
                    const unsigned long long gw = j < 2u ? g01 : g23;
                    const float g = __uint_as_float((uint32_t)(gw >> (32u * (j & 1u))));
                    if (g >= lim) {
                        if (count < (uint32_t)S2_K) clist |= ((cl >> (15u * j)) & 0x7FFFull) << (15u * count);
                        count = min(count + 1u, (uint32_t)S2_K + 1u);
                    }
                }
            }
            if (overflow) count = (uint32_t)S2_K + 1u;
            runmax = rg;
            Row12 a[KERNEL_WIDTH];
            const uint8_t *base1 = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (x - KERNEL_SIZE);
#pragma unroll
            for (int r = 0; r < KERNEL_WIDTH; r++) {
                a[r] = load_row12(base1 + (size_t)r * p.w1);
                a[r].c &= 0x00FFFFFFu;
            }
            settle(a);
        }
    }
    const int any_whole = __syncthreads_or(whole ? 1 : 0); // every thread of the workgroup is still here
    if (mine) {
        if (any_whole) contenders[(size_t)y * p.w1 + x] = word;
        if (!whole) store_cell(p, out, out_score, (size_t)y * p.w1 + x, cell);
    }
    if (counters) {
        uint32_t v0 = evaluated, v1 = exact_evals, v2 = multi, v3 = whole;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            v0 += __shfl_down(v0, sft, 64);
            v1 += __shfl_down(v1, sft, 64);
            v2 += __shfl_down(v2, sft, 64);
            v3 += __shfl_down(v3, sft, 64);
        }
        if (lane == 0) {
            if (v0) atomicAdd(&counters[0], (unsigned long long)v0);
            if (v1) atomicAdd(&counters[1], (unsigned long long)v1);
            if (v2) atomicAdd(&counters[2], (unsigned long long)v2);
            if (v3) atomicAdd(&counters[3], (unsigned long long)v3);
        }
    }
    if (whole_list.count && threadIdx.x == 0 && any_whole)
        worklist_push(whole_list, tl.x0 | (((tl.y0 - p.row0) / 4u) << 16) | (tl.nl == 64u ? 0x80000000u : (tl.nl == 58u ? 0xC0000000u : 0u)));
    return any_whole != 0;
}

template <bool COUNT>
__global__ __launch_bounds__(256, 3) void search2_filter_kernel(SearchJob ja, SearchJob jb, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    const SearchJob &j = this_job();
    const TileId tid = xcd_tile();
    (void)search2_filter_tile<COUNT>(j.p, j.img1, j.img2, j.stats1, j.stats1, j.stats2, j.range, j.contenders, j.out, j.out_score,
                                     j.counters, 0, PixTile{tid.x * 64u, j.p.row0 + tid.y * 4u, 64u, false}, j.whole,
                                     dyn_lds, lds_bytes);
}

// The first pass (64^2 .. 127^2 pixels, every candidate of the whole line: 54 per stripe) in two launches: 2 cs + 1
// workgroups per tile walk one stripe each, then one workgroup per tile merges their contenders and evaluates them.  The
// one-launch form is 32 workgroups of one wave per SIMD that each walk 270 candidates in series - 73 us of dependent
// latency on an idle chip, which every rank of a multi-GPU run pays in full.  (A single launch whose last workgroup
// per tile does the merge needs a device-scope release / acquire between workgroups on different XCDs - a write-back of
// the XCD's L2, 30 - 100 us with other kernels' dirty lines in it: measured slower than this.)
// blockIdx.x = tile column * stripes + stripe (walk);  tile column (merge).
template <bool COUNT, int MODE>
__global__ __launch_bounds__(256, 3) void search2_filter_split_kernel(SearchJob ja, SearchJob jb, uint32_t lds_bytes, uint32_t tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    const SearchJob &j = this_job();
    const uint32_t NS = MODE == 1 ? (uint32_t)(2 * j.p.corridor_size + 1) : 1u, tx = blockIdx.x / NS, sub = blockIdx.x - tx * NS;
    const uint32_t tile_index = blockIdx.y * tiles_x + tx, n_tiles = gridDim.y * tiles_x;
    (void)search2_filter_tile<COUNT, MODE>(j.p, j.img1, j.img2, j.stats1, j.stats1, j.stats2, j.range, j.contenders, j.out, j.out_score,
                                           j.counters, 0, PixTile{tx * 64u, j.p.row0 + blockIdx.y * 4u, 64u, false}, j.whole, dyn_lds,
                                           lds_bytes, (int)sub - j.p.corridor_size, j.split, tile_index, n_tiles);
}

// ---- kernel A3: displacement-plane box filter ------------------------------------------------------------
//
// Same decision rule and the same exact integer N as search2_filter_kernel, but S12 is no longer a 121-term
// dot product per (pixel, candidate).  For a displacement (dx, dy) shared by a whole row segment,
//     S12(x; dx, dy) = sum_{c = x-5}^{x+5} C(c; dx, dy),   C(c; dx, dy) = sum_{j=-5}^{5} img1(c, y+j) * img2(c+dx, y+dy+j)
// i.e. an 11-tap horizontal box sum of per-column products.  One lane owns one image column: C is 3-4
// v_dot4_u32_u8 of column-packed bytes, the box sum is a wave-wide prefix sum (6 DPP adds) and
// P(l) - P(l-11) (one ds_bpermute): ~16 VALU + 3 LDS operations per (pixel, displacement) instead of 33 dot4 +
// ~30 VALU + 12 LDS.  A wave walks the bounding box of its pixels' displacement sets; a pixel records only the
// displacements that are candidates of its own corridor (mod.rs:411-429), so the contender set is the same
// superset-of-the-band as before, and the exact re-evaluation below decides exactly as the reference does.
//
// The box walk needs every pixel's candidate set to be a rectangle of displacements: all 2*cs+1 stripes keep
// their minor coordinate over the pixel's interval (rectified pairs: always; tilted epipolar lines: only where
// the line does not step inside the interval).  A workgroup where that fails, or whose box is much larger
// than its pixels' own sets (disparity discontinuities, the first pass), taller than 9 rows or wider than 61
// columns, marks its pixels CW_FALLBACK and search3_fallback_kernel walks them candidate by candidate.
//
// Layout (TR = false; TR = true is the same with the image axes exchanged): lane l <-> image column U0 - 6 + l,
// and lanes 11..63 also own the searched pixel whose window ENDS in that column (x = U0 - 11 + l), so
// S12 = P(l) - P(l - 11).  The target image is staged TRANSPOSED, once per wave row: copy w holds, for every
// column, the 20 bytes of rows V0 + w + dy0 - 5 ... as 5 dwords (one ds_read_b128 + one ds_read_b32 per lane and
// dx serve all <= 9 dy planes); plane s = dy - dy0 multiplies them with the searched column pre-shifted by s & 3
// bytes (a[s & 3][k] against dword (s >> 2) + k).
constexpr int S3_LANE0 = 11;
constexpr int S3_OUT = S3_OUT_PX;
constexpr int S3_COLS = 128;  // staged columns per copy / per statistics row
// LDS plan.  The lean (rectified) instantiation walks at most 9 dy planes: 20 staged rows = 9 + 10 window rows (+1),
// 5 dwords per line and copy, four row-shifted copies (one per wave row) - 22.5 KB, static.  The stepped
// instantiations - epipolar lines of ANY slope, and perspective pairs - stage one tall copy of every line in dynamic
// LDS and read 24 bytes (6 dwords) of it per step, which serve up to 14 - 3 = 11 planes at any byte alignment of the
// wave's first row (box_body.inc; sizes per launch: CorrParams::box_pd / box_sh).
template <bool STEP, bool WIDE = true> struct S3Plan {
    static constexpr int MAXH = STEP ? 11 : 9;                       // dy planes (stepped: per step)
    static constexpr int NDW = STEP ? 6 : 5;                         // dwords read per line and step
    static constexpr int TAIL_BYTES = 4;                             // (lean) per line: the dword after the first four
    static constexpr int B16_OFF = 0;                                // (lean) [4][128] uint4: rows 0..15 of the copy
    static constexpr int TAIL_OFF = 4 * S3_COLS * 16;                // (lean) [4][128] u32: rows 16..19
    static constexpr int IS_OFF = TAIL_OFF + 4 * S3_COLS * TAIL_BYTES; // (lean) [MAXH + 3][128] uint2: candidate statistics
    static constexpr int LDS_BYTES = IS_OFF + (MAXH + 3) * S3_COLS * 8;
    // Stepped plan, two widths: boxes of up to 61 steps (128 lines, 128 cells per statistics row) where the lines are
    // shallow enough for that to leave five workgroups per CU; up to 33 steps (100 lines, 96 cells) for steeper lines,
    // whose boxes are taller - the statistics rows are what fills the LDS.
    static constexpr int LINES = STEP && !WIDE ? 100 : S3_COLS;      // target lines staged per workgroup
    static constexpr int ISP = STEP && !WIDE ? 96 : S3_COLS;         // cells per row of candidate statistics
};
// dynamic LDS of a stepped launch: LINES lines of pd dwords, then sh statistics rows of ISP cells
static inline uint32_t search3_step_lds_bytes(uint32_t pd, uint32_t sh, bool wide)
{
    const uint32_t plan = wide ? (uint32_t)S3Plan<true, true>::LINES * pd * 4u + sh * (uint32_t)S3Plan<true, true>::ISP * 8u
                               : (uint32_t)S3Plan<true, false>::LINES * pd * 4u + sh * (uint32_t)S3Plan<true, false>::ISP * 8u;
    return plan > 4u * 128u * 12u ? plan : 4u * 128u * 12u; // (at least the four waves' contender queues: box_body.inc)
}

__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t v)
{
    // inclusive prefix sum over the 64 lanes: Kogge-Stone inside each row of 16 (row_shr 1, 2, 4, 8; lanes
    // shifted in from outside the row read 0), then the row totals (row_bcast:15 -> rows 1 and 3, row_bcast:31
    // -> rows 2 and 3).  EXEC must be all ones.
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
// wave-wide min / max with the same DPP pattern (lane 63 ends up with the total); EXEC all ones
__device__ __forceinline__ int wave_min_i32(int v)
{
    const int id = 0x7FFFFFFF;
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xA, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

__device__ __forceinline__ uint32_t load_dword_checked(const uint8_t *__restrict__ img, int w, int h, int row, int col)
{
    uint32_t v = 0;
    if (row < 0 || row >= h) return 0u;
    const uint8_t *p = img + (size_t)row * (size_t)w;
    if (col >= 0 && col + 3 < w) {
        __builtin_memcpy(&v, p + col, 4);
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (col + b >= 0 && col + b < w) v |= (uint32_t)p[col + b] << (8 * b);
    }
    return v;
}

// STEP = false: rectified pairs only (every candidate set a rectangle; anything else is declined);
// STEP = true: additionally lines that step inside a pixel's interval (per-step offset and plane window).
// TR = false: lanes run along x and the four waves of a workgroup along y (row-major epipolar lines: the long side
// of the displacement box is dx);  TR = true: the same kernel with the two image axes exchanged - lanes along y,
// waves along x, the packed 11-byte vectors are image ROWS - for column-major lines.  Below, u is the lane axis
// and v the other one; S12 is the same integer either way.
// Launch bound of the lean instantiation: 6 waves/SIMD (76 VGPRs, no spill).  7 waves - what its 22.5 KB of LDS would
// admit - was measured 1.8 % faster (5.89 vs 6.00 ms per 4096^2 pair) but only with 10 VGPRs spilled: 28 B of scratch
// per lane x 54 M threads put +1.17 GB per step on the L2 write-back counter (whole step 3.0 -> 4.2 GB).  Not taken.
// (the candidate-counting instantiations - profiling and tests only - need 85 registers (lean) and ~100 (stepped): five and
// four waves rather than 28 B of scratch)
#ifndef CVHIP_STEP_WAVES
#define CVHIP_STEP_WAVES 5
#endif
template <bool COUNT, bool STEP, bool TR>
__global__ __launch_bounds__(256, STEP ? (COUNT ? 4 : CVHIP_STEP_WAVES) : ((TR || COUNT) ? 5 : 6)) void search3_box_kernel(SearchJob ja, SearchJob jb)
{
    const SearchJob &j = this_job(); // both directions of a level in one launch (see search_range_kernel)
    const CorrParams &p = j.p;
    const uint8_t *__restrict__ const img1 = j.img1, *__restrict__ const img2 = j.img2;
    const uint2 *__restrict__ const stats1 = j.stats1, *__restrict__ const istats1 = j.stats1, *__restrict__ const istats2 = j.stats2;
    const uint32_t *__restrict__ const range = j.range;
    unsigned long long *__restrict__ const contenders = j.contenders, *__restrict__ const counters = j.counters;
    uint32_t *__restrict__ const out = j.out;
    float *__restrict__ const out_score = j.out_score;
    const WorkList declined = j.declined, whole_list = j.whole;
    constexpr bool WIDE = true;
#include "box_body.inc"
}
// One job per launch, plain by-value arguments.  The stepped instantiations sit at the register limit of their
// occupancy target (96 VGPRs, 5 waves per SIMD): addressing the job through the kernel-argument pointer costs them
// either 24 spilled VGPRs or one wave of occupancy (3-degree pair, level 0: 5.7 ms -> 7.6 / 6.6 ms), so their two
// directions stay two launches of this form.
template <bool COUNT, bool STEP, bool TR, bool WIDE>
__global__ __launch_bounds__(256, STEP ? (COUNT ? 4 : CVHIP_STEP_WAVES) : (TR ? 5 : 6)) void search3_box_single_kernel(
    CorrParams p, const uint8_t *__restrict__ img1, const uint8_t *__restrict__ img2, const uint2 *__restrict__ stats1,
    const uint2 *__restrict__ istats1, const uint2 *__restrict__ istats2, const uint32_t *__restrict__ range,
    unsigned long long *__restrict__ contenders, uint32_t *__restrict__ out, float *__restrict__ out_score,
    unsigned long long *__restrict__ counters, WorkList declined, WorkList whole_list)
{
#include "box_body.inc"
}

// ---------------------------------------------------------------------------------------------------------------
// search3_box2_kernel: the lean box walk (rectified pairs: exactly axis-parallel row-major lines, rectangles only) with TWO
// image columns per lane (round 5; scripts/micro/box_group.hip: 0.70-0.77 of the one-column group's time per pixel and plane).
// Lane l >= 1 owns the image columns cE = X0 - 7 + 2l and cO = cE + 1 (lane 0 is a dummy pair with zero products, so that the
// prefix sum needs no special case at the left edge) and lanes 6..63 own the two searched pixels whose windows END in those
// columns: x = X0 + 2 (l - 6) + {0, 1}, 116 pixels per wave.  With C_E, C_O the lane's column products of one displacement,
// Q = wave prefix sum of C_E + C_O (the second dot4 chain accumulates onto the first) and R = Q - C_O,
//     S12(window ending at cE) = R(l) - Q(l - 6),      S12(window ending at cO) = Q(l) - R(l - 5)
// - per plane two chains of 3-4 v_dot4_u32_u8, ONE six-step DPP prefix sum and two ds_bpermute for two pixels.  The integers
// (N = 121 S12 - s1 s2), the band test, the contender bookkeeping, DELTA and the exact re-evaluation are search3_box_kernel's.
// LDS (33 KB): the target lines as four row-shifted copies of 20 bytes like the lean plan, 192 columns per copy with the even
// and the odd columns in separate halves (a lane's two 16-byte reads are then unit-stride across the wave at every step),
// tails and candidate statistics in plain column order (two adjacent 4- / 8-byte cells: ds_read2).
// ---------------------------------------------------------------------------------------------------------------
constexpr int P2_OUT = 116, P2_LANE0 = 6;
constexpr int P2_COLS = 192; // staged target columns per copy: 126 + 61 - 1 + 3
constexpr int P2_ISP = 192;  // cells per row of candidate statistics: 116 + 61 - 1
constexpr int P2_MAXH = 9, P2_NDW = 5;
constexpr int P2_B16_OFF = 0, P2_TAIL_OFF = 4 * P2_COLS * 16, P2_IS_OFF = P2_TAIL_OFF + 4 * P2_COLS * 4;
constexpr int P2_LDS_BYTES = P2_IS_OFF + (P2_MAXH + 3) * P2_ISP * 8;
struct P2Pixel { // one of a lane's two searched pixels
    bool is_out, active, has, whole;
    uint32_t x, r0;
    int lox, loy;
    uint32_t wx, wy;
    uint32_t s1;
    float k1, c1, limk, runmax;
    unsigned long long clist, word;
    uint32_t count, evaluated, ecount;
    uint2 cell;
};
template <bool COUNT>
__global__ __launch_bounds__(256, 3) void search3_box2_kernel(SearchJob ja, SearchJob jb)
{
    const SearchJob &j = this_job();
    const CorrParams &p = j.p;
    const uint8_t *__restrict__ const img1 = j.img1, *__restrict__ const img2 = j.img2;
    const uint2 *__restrict__ const stats1 = j.stats1, *__restrict__ const istats2 = j.stats2;
    const uint32_t *__restrict__ const range = j.range;
    unsigned long long *__restrict__ const contenders = j.contenders, *__restrict__ const counters = j.counters;
    uint32_t *__restrict__ const out = j.out;
    float *__restrict__ const out_score = j.out_score;
    const WorkList declined = j.declined, whole_list = j.whole;

    __shared__ __attribute__((aligned(16))) uint8_t lds[P2_LDS_BYTES];
    __shared__ int bb[6]; // min dx, min dy, max dx, max dy, max candidates of one pixel, 1 = some pixel is not a rectangle
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const TileId tid = xcd_tile();
    const int X0 = (int)tid.x * P2_OUT;
    const uint32_t V0 = p.row0 + tid.y * 4, y = V0 + w;
    const int cs = p.corridor_size;
    if (threadIdx.x == 0) {
        bb[0] = 0x7FFFFFFF;
        bb[1] = 0x7FFFFFFF;
        bb[2] = -0x7FFFFFFF;
        bb[3] = -0x7FFFFFFF;
        bb[4] = 0;
        bb[5] = 0;
    }
    // ---- the two pixels' candidate sets in displacement space (box_body.inc's setup, rectangles only) ------------------
    P2Pixel px[2];
    bool any_odd = false;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        P2Pixel &o = px[s];
        const int xi = X0 + 2 * ((int)lane - P2_LANE0) + s;
        o.x = (uint32_t)xi;
        o.is_out = lane >= (uint32_t)P2_LANE0 && o.x < p.w1 && y < p.row1;
        o.whole = false;
        o.word = 0ull;
        o.cell = make_uint2(CELL_NONE, 0x7FC00000u);
        o.evaluated = o.ecount = o.count = 0u;
        o.clist = 0ull;
        o.runmax = -__builtin_inff();
        o.lox = o.loy = 0;
        o.wx = o.wy = 0u;
        o.r0 = 0u;
        PixelSetup ps;
        ps.st1 = make_float2(0.0f, 1.0f);
        ps.e.cx = ps.e.cy = ps.e.ax = ps.e.ay = 0.0;
        ps.e.ox = ps.e.oy = 0;
        ps.r0 = ps.r1 = 0;
        o.active = o.is_out && pixel_setup(p, o.x, y, stats1, range, ps);
        bool simple = true;
        if (o.active) {
            const Line &e = ps.e;
            const bool major_x = e.ox == 0;
            const double cmn = major_x ? e.cy : e.cx, amn = major_x ? e.ay : e.ax;
            const int omn = major_x ? e.oy : e.ox;
            const double vf = cmn * (double)ps.r0 + amn;
            uint32_t m0 = 0;
            bool consecutive = true;
            for (int off = -cs; off <= cs; off++) {
                const uint32_t mf = f64_to_u32_sat(floor(vf + (double)(off * omn)));
                if (off == -cs) m0 = mf;
                consecutive = consecutive && mf == m0 + (uint32_t)(off + cs);
            }
            const uint32_t lim2 = major_x ? p.w2 : p.h2;
            const uint32_t ilo = max(ps.r0, (uint32_t)KERNEL_SIZE), ihi = min(ps.r1, sat_sub_u32(lim2, KERNEL_SIZE));
            const uint32_t nmaj = ihi > ilo ? ihi - ilo : 0u;
            const bool sane = (ps.r1 - ps.r0) <= CW_MAX_LEN && m0 < 0x40000000u && ilo < 0x40000000u;
            // (this kernel is only launched for exactly axis-parallel row-major lines; anything else declines)
            simple = consecutive && sane && major_x && e.cy == 0.0;
            o.r0 = ps.r0;
            o.lox = (int)ilo - xi;
            o.wx = nmaj;
            o.loy = (int)m0 - (int)y;
            o.wy = (uint32_t)(2 * cs + 1);
        }
        o.has = o.active && simple && o.wx > 0u && o.wy > 0u;
        any_odd = any_odd || (o.active && !simple);
        o.s1 = o.has ? (stats1[(size_t)y * p.w1 + o.x].x & 0x7FFFFFFFu) : 0u;
        o.k1 = ps.st1.y * (float)(KERNEL_POINT_COUNT * KERNEL_POINT_COUNT); // 121*121*sd1
        o.c1 = 1.0f / o.k1;
    }
    const float thr_lo = p.threshold - S2_DELTA;
#pragma unroll
    for (int s = 0; s < 2; s++) px[s].limk = px[s].has ? thr_lo * px[s].k1 * (1.0f - 9.5367431640625e-7f) : __builtin_inff();
    // wave-uniform bounds of the wave's displacement box
    const int big = 0x7FFFFFFF;
    const int mnx = wave_min_i32(min(px[0].has ? px[0].lox : big, px[1].has ? px[1].lox : big));
    const int mny = wave_min_i32(min(px[0].has ? px[0].loy : big, px[1].has ? px[1].loy : big));
    const int mxx = wave_max_i32(max(px[0].has ? px[0].lox + (int)px[0].wx - 1 : -big, px[1].has ? px[1].lox + (int)px[1].wx - 1 : -big));
    const int mxy = wave_max_i32(max(px[0].has ? px[0].loy + (int)px[0].wy - 1 : -big, px[1].has ? px[1].loy + (int)px[1].wy - 1 : -big));
    const int need = wave_max_i32(max(px[0].has ? (int)min(px[0].wx * px[0].wy, 0x3FFFFFFFu) : 0, px[1].has ? (int)min(px[1].wx * px[1].wy, 0x3FFFFFFFu) : 0));
    const bool wave_has = mxx >= mnx;
    const bool wave_odd = __any(any_odd);
    __syncthreads(); // bb initialised
    if (lane == 0) {
        if (wave_has) {
            atomicMin(&bb[0], mnx);
            atomicMin(&bb[1], mny);
            atomicMax(&bb[2], mxx);
            atomicMax(&bb[3], mxy);
            atomicMax(&bb[4], need);
        }
        if (wave_odd) atomicOr(&bb[5], 1);
    }
    __syncthreads();
    const bool any_has = bb[2] >= bb[0];
    const int dx0 = bb[0], dy0 = bb[1];
    const int W = bb[2] - bb[0] + 1, H = bb[3] - bb[1] + 1;
    const int C0 = X0 - 5 + dx0;   // target column of lane 1's first slot at the box's first step
    const int C0a = C0 & ~3;
    const int colshift = C0 - C0a;
    const int ncol = colshift + 126 + W - 1;
    bool eligible = !bb[5];
    if (any_has) eligible = eligible && ncol <= P2_COLS && 116 + W - 1 <= P2_ISP && H <= P2_MAXH && (long long)W * H <= 3ll * bb[4] + 16;
    if (!eligible || !any_has) {
        // nothing to search (every pixel None), or left to the candidate-by-candidate kernel: two 58-pixel tiles on its list
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (px[s].is_out) {
                const size_t pix = (size_t)y * p.w1 + px[s].x;
                const bool fb = px[s].active && !eligible;
                if (!eligible) contenders[pix] = fb ? (CW_FALLBACK << 60) : 0ull;
                if (!fb) out[pix] = CELL_NONE;
            }
        if (!eligible && threadIdx.x == 0) {
            worklist_push(declined, (uint32_t)X0 | (tid.y << 16) | 0xC0000000u);
            if ((uint32_t)X0 + 58u < p.w1) worklist_push(declined, ((uint32_t)X0 + 58u) | (tid.y << 16) | 0xC0000000u);
        }
        return;
    }
    const int NPL = H > 5 ? 9 : 5; // planes staged: group A = 0..4, group B = 5..8

    // ---- stage the target lines (4 row-shifted copies of 20 bytes, even / odd columns apart) and the candidate statistics
    {
        const int R0 = (int)V0 + dy0 - KERNEL_SIZE; // first row of copy 0
        const int nb = (ncol + 3) >> 2;             // groups of four columns (<= 48)
        const int kmax = NPL > 5 ? P2_NDW : P2_NDW - 1; // (five planes never read the tail dword)
        const bool inside = R0 >= 0 && R0 + 3 + 4 * (P2_NDW - 1) + 3 < (int)p.h2 && C0a >= 0 && C0a + 4 * nb <= (int)p.w2;
        const int b = (int)(threadIdx.x & 63u);
        if (b < nb) {
            for (int wk = (int)(threadIdx.x >> 6); wk < 4 * P2_NDW; wk += 4) {
                const int cw = wk / P2_NDW, k = wk - cw * P2_NDW;
                if (k >= kmax) continue;
                const int rr = R0 + cw + 4 * k, col = C0a + 4 * b;
                uint32_t d0, d1, d2, d3;
                if (inside) {
                    const uint8_t *src = img2 + (size_t)rr * p.w2 + (size_t)col;
                    __builtin_memcpy(&d0, src, 4);
                    __builtin_memcpy(&d1, src + p.w2, 4);
                    __builtin_memcpy(&d2, src + 2 * (size_t)p.w2, 4);
                    __builtin_memcpy(&d3, src + 3 * (size_t)p.w2, 4);
                } else {
                    d0 = load_dword_checked(img2, (int)p.w2, (int)p.h2, rr + 0, col);
                    d1 = load_dword_checked(img2, (int)p.w2, (int)p.h2, rr + 1, col);
                    d2 = load_dword_checked(img2, (int)p.w2, (int)p.h2, rr + 2, col);
                    d3 = load_dword_checked(img2, (int)p.w2, (int)p.h2, rr + 3, col);
                }
                // 4x4 byte transpose: t.c = rows rr..rr+3 of column col + c
                const uint32_t p01 = __builtin_amdgcn_perm(d1, d0, 0x05010400u), q01 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);
                const uint32_t p23 = __builtin_amdgcn_perm(d3, d2, 0x05010400u), q23 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
                uint4 t;
                t.x = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                t.y = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
                t.z = __builtin_amdgcn_perm(q23, q01, 0x05040100u);
                t.w = __builtin_amdgcn_perm(q23, q01, 0x07060302u);
                if (k < 4) {
                    // columns 4b, 4b + 2 -> even half entries 2b, 2b + 1; columns 4b + 1, 4b + 3 -> odd half, the same entries
                    uint32_t *ev = reinterpret_cast<uint32_t *>(lds + P2_B16_OFF + (size_t)(cw * P2_COLS + 2 * b) * 16u) + k;
                    uint32_t *od = ev + (P2_COLS / 2) * 4;
                    ev[0] = t.x;
                    od[0] = t.y;
                    ev[4] = t.z;
                    od[4] = t.w;
                } else {
                    *reinterpret_cast<uint4 *>(lds + P2_TAIL_OFF + (size_t)(cw * P2_COLS + 4 * b) * 4u) = t;
                }
            }
        }
        const int isp = 116 + W - 1, isrows = NPL + 3;
        const int gu0 = X0 + dx0; // target x of pixel 0 at the box's first step
        const int gv0 = (int)V0 + dy0;
        const int c = (int)threadIdx.x;
        if (c < isp) {
            const int gx = gu0 + c;
            for (int r = 0; r < isrows; r++) {
                const int gy = gv0 + r;
                // {-window sum, f32 stdev}; centres outside the image or skipped by the reference (mod.rs:430-441): stdev = +inf
                uint2 v = make_uint2(0u, 0x7F800000u);
                if (gy >= 0 && gy < (int)p.h2 && gx >= 0 && gx < (int)p.w2) {
                    const uint2 tt = istats2[(size_t)gy * p.w2 + (size_t)gx];
                    if (tt.x & 0x80000000u) v = make_uint2(0u - (tt.x & 0x7FFFFFFFu), tt.y);
                }
                *reinterpret_cast<uint2 *>(lds + P2_IS_OFF + (size_t)(r * P2_ISP + c) * 8u) = v;
            }
        }
    }
    // this lane's two searched columns, rows y-5 .. y+5, packed and pre-shifted by 0..3 bytes (lane 0: zeros)
    uint32_t a[2][4][4];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int uc = X0 - 7 + 2 * (int)lane + s;
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        if (lane >= 1u && uc >= 0 && uc < (int)p.w1 && y >= (uint32_t)KERNEL_SIZE && y + KERNEL_SIZE < p.h1) {
            const uint8_t *pc = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (size_t)uc;
            uint32_t bv[KERNEL_WIDTH];
#pragma unroll
            for (int r = 0; r < KERNEL_WIDTH; r++) bv[r] = pc[(size_t)r * p.w1];
            a0 = bv[0] | (bv[1] << 8) | (bv[2] << 16) | (bv[3] << 24);
            a1 = bv[4] | (bv[5] << 8) | (bv[6] << 16) | (bv[7] << 24);
            a2 = bv[8] | (bv[9] << 8) | (bv[10] << 16);
        }
        a[s][0][0] = a0;
        a[s][0][1] = a1;
        a[s][0][2] = a2;
        a[s][0][3] = 0u;
#pragma unroll
        for (int q = 1; q < 4; q++) { // 128-bit left shift by q bytes
            a[s][q][0] = a0 << (8 * q);
            a[s][q][1] = __builtin_amdgcn_alignbyte(a1, a0, 4 - q);
            a[s][q][2] = __builtin_amdgcn_alignbyte(a2, a1, 4 - q);
            a[s][q][3] = __builtin_amdgcn_alignbyte(0u, a2, 4 - q);
        }
    }
    __syncthreads();

    const auto record = [&](P2Pixel &o, float g, uint32_t code) {
        const float lim = fmaxf(o.runmax - 2.0f * S2_DELTA, thr_lo);
        if (g >= lim) {
            if (g > o.runmax + 2.0f * S2_DELTA) { // everything recorded so far is out of the band
                o.count = 0;
                o.clist = 0ull;
            }
            o.runmax = fmaxf(o.runmax, g);
            o.limk = fmaxf(o.runmax - 2.0f * S2_DELTA, thr_lo) * o.k1 * (1.0f - 9.5367431640625e-7f);
            if (o.count < (uint32_t)S2_K) o.clist |= (unsigned long long)code << (15u * o.count);
            o.count = min(o.count + 1u, (uint32_t)S2_K + 1u);
        }
    };

    if (wave_has) {
        // staged column of the lane's first slot at the wave's first step (lane 0 reads lane 1's: its products are zeros anyway)
        const int j0 = colshift + 2 * ((int)max(lane, 1u) - 1) + (mnx - dx0);
        const uint8_t *const bCopy = lds + P2_B16_OFF + (size_t)(w * P2_COLS) * 16u;
        const uint8_t *const bTail = lds + P2_TAIL_OFF + (size_t)(w * P2_COLS + j0) * 4u;
        // candidate statistics of the lane's first pixel at the wave's first step (lanes below 6 read lane 6's)
        const int p0 = 2 * ((int)max(lane, (uint32_t)P2_LANE0) - P2_LANE0) + (mnx - dx0);
        const uint8_t *const bIS = lds + P2_IS_OFF + (size_t)(w * P2_ISP + p0) * 8u;
        const int idx5 = (int)(lane >= 5u ? lane - 5u : 0u) * 4, idx6 = (int)(lane >= 6u ? lane - 6u : 0u) * 4;
        const int nsteps = mxx - mnx + 1;
        const int mid = nsteps >> 1;
        for (int t = 0; t < nsteps; t++) {
            const int step = t < nsteps - mid ? mid + t : nsteps - 1 - t; // inside out (see box_body.inc)
            const int dx = mnx + step;
            const bool mxe = px[0].has && (uint32_t)(dx - px[0].lox) < px[0].wx, mxo = px[1].has && (uint32_t)(dx - px[1].lox) < px[1].wx;
            // the two columns' lines: staged columns jE = j0 + step (first slot) and jE + 1, in the half of their parity
            const int jE = j0 + step;
            const uint8_t *const pE = bCopy + (size_t)((jE & 1) * (P2_COLS / 2) + (jE >> 1)) * 16u;
            const uint8_t *const pO = bCopy + (size_t)(((jE + 1) & 1) * (P2_COLS / 2) + ((jE + 1) >> 1)) * 16u;
            const uint4 ra = *reinterpret_cast<const uint4 *>(pE), rb = *reinterpret_cast<const uint4 *>(pO);
            const uint32_t *const tl = reinterpret_cast<const uint32_t *>(bTail + step * 4);
            const uint32_t raw[2][P2_NDW] = {{ra.x, ra.y, ra.z, ra.w, tl[0]}, {rb.x, rb.y, rb.z, rb.w, tl[1]}};
            const uint8_t *const pIS = bIS + step * 8;
            auto group = [&](auto s0_tag, auto n_tag) {
                constexpr int S0 = decltype(s0_tag)::value, N = decltype(n_tag)::value;
                int num[2][N];
                float sd[2][N], mg[2][N];
#pragma unroll
                for (int q = 0; q < N; q++) {
                    const int sp = S0 + q, sh = sp & 3, o = sp >> 2;
                    const uint2 ise = *reinterpret_cast<const uint2 *>(pIS + sp * (P2_ISP * 8));
                    const uint2 iso = *reinterpret_cast<const uint2 *>(pIS + sp * (P2_ISP * 8) + 8);
                    uint32_t c1 = __builtin_amdgcn_udot4(a[1][sh][0], raw[1][o], 0u, false);
                    c1 = __builtin_amdgcn_udot4(a[1][sh][1], raw[1][o + 1], c1, false);
                    c1 = __builtin_amdgcn_udot4(a[1][sh][2], raw[1][o + 2], c1, false);
                    if (sh >= 2) c1 = __builtin_amdgcn_udot4(a[1][sh][3], raw[1][o + 3 < P2_NDW ? o + 3 : P2_NDW - 1], c1, false);
                    uint32_t d = __builtin_amdgcn_udot4(a[0][sh][0], raw[0][o], c1, false); // the first column on top: the pair sum
                    d = __builtin_amdgcn_udot4(a[0][sh][1], raw[0][o + 1], d, false);
                    d = __builtin_amdgcn_udot4(a[0][sh][2], raw[0][o + 2], d, false);
                    if (sh >= 2) d = __builtin_amdgcn_udot4(a[0][sh][3], raw[0][o + 3 < P2_NDW ? o + 3 : P2_NDW - 1], d, false);
                    const uint32_t Q = wave_prefix_sum(d), R = Q - c1;
                    const uint32_t s_odd = Q - (uint32_t)__builtin_amdgcn_ds_bpermute(idx5, (int)R);
                    const uint32_t s_even = R - (uint32_t)__builtin_amdgcn_ds_bpermute(idx6, (int)Q);
                    num[0][q] = __mul24((int)s_even, KERNEL_POINT_COUNT) + __mul24((int)px[0].s1, (int)ise.x);
                    num[1][q] = __mul24((int)s_odd, KERNEL_POINT_COUNT) + __mul24((int)px[1].s1, (int)iso.x);
                    sd[0][q] = __uint_as_float(ise.y);
                    sd[1][q] = __uint_as_float(iso.y);
                }
#pragma unroll
                for (int q = 0; q < N; q++) {
                    mg[0][q] = __builtin_fmaf(-px[0].limk, sd[0][q], (float)num[0][q]);
                    mg[1][q] = __builtin_fmaf(-px[1].limk, sd[1][q], (float)num[1][q]);
                }
                float margin = fmaxf(mg[0][0], mg[1][0]);
#pragma unroll
                for (int q = 1; q < N; q++) margin = fmaxf(margin, fmaxf(mg[0][q], mg[1][q]));
                if (COUNT) {
#pragma unroll
                    for (int q = 0; q < N; q++) {
                        if (mxe && (uint32_t)(dy0 + S0 + q - px[0].loy) < px[0].wy && sd[0][q] < __builtin_inff()) px[0].evaluated++;
                        if (mxo && (uint32_t)(dy0 + S0 + q - px[1].loy) < px[1].wy && sd[1][q] < __builtin_inff()) px[1].evaluated++;
                    }
                }
                if (margin >= 0.0f) { // rarely taken; planes from the middle outwards (see box_body.inc)
#pragma unroll
                    for (int qi = 0; qi < N; qi++) {
                        const int q = (N - 1) / 2 + ((qi & 1) ? (qi + 1) / 2 : -(qi / 2));
                        const int dy = dy0 + S0 + q;
#pragma unroll
                        for (int s = 0; s < 2; s++) {
                            P2Pixel &o = px[s];
                            const bool mx = s ? mxo : mxe;
                            if (mg[s][q] >= 0.0f && mx && (uint32_t)(dy - o.loy) < o.wy && sd[s][q] < __builtin_inff() &&
                                (float)num[s][q] >= o.limk * sd[s][q]) {
                                const uint32_t code = ((uint32_t)(dy - o.loy) << 11) | (uint32_t)((int)o.x + dx - (int)o.r0);
                                record(o, (float)num[s][q] * (o.c1 * __builtin_amdgcn_rcpf(sd[s][q])), code);
                            }
                        }
                    }
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I4 = std::integral_constant<int, 4>;
            using I5 = std::integral_constant<int, 5>;
            group(I0{}, I5{});
            if (NPL > 5) group(I5{}, I4{});
        }
    }

    // ---- exact re-evaluation of the contenders (mod.rs:442-464; box_body.inc's exact phase): a lane evaluates its first
    // pixel's first contender itself, every other contender of its two pixels goes through the queue in LDS (the wave's own
    // copy of the target lines, dead after the walk) to whichever lanes are free - with two pixels per lane that is a second
    // pass of the chain for the wave, as one pass per 58 pixels was before.
    const auto chain = [&](uint32_t qx1, uint32_t qy1, uint32_t qx, uint32_t qy) -> float {
        const float2 st1x = stats_of(stats1[(size_t)qy1 * p.w1 + qx1]);
        const uint2 is2 = istats2[(size_t)qy * p.w2 + qx];
        const float avg1 = st1x.x, sdev1 = st1x.y;
        const float avg2 = (float)(is2.x & 0x7FFFFFFFu) / (float)KERNEL_POINT_COUNT; // == compute_point_avg
        const float sdev2 = __uint_as_float(is2.y);
        const uint8_t *base1 = img1 + (size_t)(qy1 - KERNEL_SIZE) * p.w1 + (qx1 - KERNEL_SIZE);
        const uint8_t *base2 = img2 + (size_t)(qy - KERNEL_SIZE) * p.w2 + (qx - KERNEL_SIZE);
        float corr = 0.0f;
#pragma unroll 1
        for (int rb = 0; rb < 12; rb += 4) {
            Row12 av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rr = min(rb + r, KERNEL_WIDTH - 1);
                av[r] = load_row12(base1 + (size_t)rr * p.w1);
                bv[r] = load_row12(base2 + (size_t)rr * p.w2);
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (rb + r < KERNEL_WIDTH) corr = row_corr_acc(corr, av[r], bv[r], avg1, avg2);
        }
        return corr / (sdev1 * sdev2 * (float)KERNEL_POINT_COUNT); // mod.rs:454
    };
    // the candidate a contender code names (rectified lines: x2 = i, y2 = the stripe's row - candidate_xy with the line re-derived)
    const auto cand_of = [&](const P2Pixel &o, uint32_t code) -> uint32_t {
        const Line ex = epipolar_line(p, o.x, y);
        const CandXY c = candidate_xy(ex, o.r0 + (code & 0x7FFu), (int)(code >> 11) - cs);
        return c.x | (c.y << 16);
    };
#pragma unroll
    for (int s = 0; s < 2; s++) {
        P2Pixel &o = px[s];
        if (o.has) {
            if (o.count > (uint32_t)S2_K) {
                o.word = CW_WHOLE << 60;
                o.whole = true;
                o.evaluated = 0; // the exact kernel walks (and counts) the whole corridor
            } else if (o.count > 0u) {
                o.ecount = o.count;
                if (!p.need_scores && o.count == 1u && o.runmax >= p.threshold + S2_DELTA) { // settled without the chain
                    o.cell = make_uint2(cand_of(o, (uint32_t)o.clist & 0x7FFFu), __float_as_uint(o.runmax));
                    o.ecount = 0u;
                }
            }
        }
    }
    constexpr uint32_t QCAP = 192;
    static_assert(QCAP * 12u <= P2_COLS * 16u, "the queue lives in one wave's copy of the target lines");
    uint32_t *const q_pix = reinterpret_cast<uint32_t *>(lds + P2_B16_OFF + (size_t)(w * P2_COLS) * 16u);
    uint32_t *const q_cand = q_pix + QCAP;
    float *const q_corr = reinterpret_cast<float *>(q_cand + QCAP);
    // queue items of this lane: the first pixel's contenders beyond its first, all of the second pixel's
    uint32_t extras = (px[0].ecount > 1u ? px[0].ecount - 1u : 0u) + px[1].ecount;
    uint32_t q_incl = wave_prefix_sum(extras);
    uint32_t q_total = (uint32_t)__builtin_amdgcn_readlane((int)q_incl, 63);
    if (q_total > QCAP) {
        // more than the queue holds (synthetic ties): the pixels behind the limit re-evaluate their whole corridor in the fallback kernel
        if (q_incl > QCAP && extras > 0u) {
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (px[s].ecount > 0u) {
                    px[s].word = CW_WHOLE << 60;
                    px[s].whole = true;
                    px[s].evaluated = 0;
                    px[s].ecount = 0;
                }
            extras = 0;
        }
        q_incl = wave_prefix_sum(extras);
        q_total = (uint32_t)__builtin_amdgcn_readlane((int)q_incl, 63);
    }
    const uint32_t q_base = q_incl - extras;
    {
        uint32_t at = q_base;
#pragma unroll
        for (int s = 0; s < 2; s++)
            for (uint32_t jx = s == 0 ? 1u : 0u; jx < px[s].ecount; jx++) {
                q_pix[at] = px[s].x | (y << 16);
                q_cand[at] = cand_of(px[s], (uint32_t)(px[s].clist >> (15u * jx)) & 0x7FFFu);
                at++;
            }
    }
    uint32_t own_cxy = 0;
    if (px[0].ecount >= 1u) own_cxy = cand_of(px[0], (uint32_t)px[0].clist & 0x7FFFu);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float own_corr = 0.0f;
    uint32_t exact_evals = 0;
    const unsigned long long idle = __ballot(px[0].ecount == 0u);
    const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
    const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
    uint32_t done = 0;
    bool first_round = true;
    while (first_round || done < q_total) { // (wave-uniform)
        const bool own = first_round && px[0].ecount >= 1u;
        const uint32_t item = done + (first_round ? my_rank : lane);
        const bool take = !own && (!first_round || px[0].ecount == 0u) && item < q_total;
        uint32_t pxy = px[0].x | (y << 16), cxy = own_cxy;
        if (take) {
            pxy = q_pix[item];
            cxy = q_cand[item];
        }
        if (own || take) {
            const float corr = chain(pxy & 0xFFFFu, pxy >> 16, cxy & 0xFFFFu, cxy >> 16);
            exact_evals++;
            if (own) own_corr = corr;
            else q_corr[item] = corr;
        }
        done += first_round ? min(n_idle, q_total) : 64u;
        first_round = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        uint32_t at = q_base;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            P2Pixel &o = px[s];
            bool have = false;
            float bcorr = 0.0f;
            uint32_t bxy = 0, bcode = 0;
            for (uint32_t jx = 0; jx < o.ecount; jx++) { // mod.rs:456-464; "first maximum" = the smallest code among equals
                const uint32_t code = (uint32_t)(o.clist >> (15u * jx)) & 0x7FFFu;
                float corr;
                uint32_t cxy;
                if (s == 0 && jx == 0) {
                    corr = own_corr;
                    cxy = own_cxy;
                } else {
                    corr = q_corr[at];
                    cxy = q_cand[at];
                    at++;
                }
                if (corr >= p.threshold && (!have || corr > bcorr || (corr == bcorr && code < bcode))) {
                    have = true;
                    bcorr = corr;
                    bcode = code;
                    bxy = cxy;
                }
            }
            if (have) o.cell = make_uint2(bxy, __float_as_uint(bcorr));
            if (o.is_out && !o.whole) store_cell(p, out, out_score, (size_t)y * p.w1 + o.x, o.cell);
        }
    }
    if (counters) {
        uint32_t v0 = px[0].evaluated + px[1].evaluated, v1 = exact_evals;
        uint32_t v2 = (px[0].ecount > 1u ? 1u : 0u) + (px[1].ecount > 1u ? 1u : 0u), v3 = (px[0].whole ? 1u : 0u) + (px[1].whole ? 1u : 0u);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            v0 += __shfl_down(v0, sft, 64);
            v1 += __shfl_down(v1, sft, 64);
            v2 += __shfl_down(v2, sft, 64);
            v3 += __shfl_down(v3, sft, 64);
        }
        if (lane == 0) {
            if (v0) atomicAdd(&counters[0], (unsigned long long)v0);
            if (v1) atomicAdd(&counters[1], (unsigned long long)v1);
            if (v2) atomicAdd(&counters[2], (unsigned long long)v2);
            if (v3) atomicAdd(&counters[3], (unsigned long long)v3);
        }
    }
    {
        // only a tile that goes onto the whole-corridor list has its contender words read (all of them)
        const int any_whole = __syncthreads_or((px[0].whole || px[1].whole) ? 1 : 0);
        if (any_whole) {
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (px[s].is_out) contenders[(size_t)y * p.w1 + px[s].x] = px[s].word;
            if (threadIdx.x == 0) {
                worklist_push(whole_list, (uint32_t)X0 | (tid.y << 16) | 0xC0000000u);
                if ((uint32_t)X0 + 58u < p.w1) worklist_push(whole_list, ((uint32_t)X0 + 58u) | (tid.y << 16) | 0xC0000000u);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// search4_mfma_kernel: the filter's numerator as a small dense product on the matrix pipe (int8 MFMA) - for rectified
// affine pairs (exactly axis-parallel row-major lines, five stripes), where search3_box_kernel's lean instantiation ran.
//   S12(pixel, position) = sum over the 11 x 11 window of a * b     with a' = a - 128, b' = b - 128 as int8:
//   sum a' b' = S12 - 128 s1 - 128 s2 + 121 * 128^2   ->   N = 121 S12 - s1 s2 = 121 acc + P + (-s2) (s1 - 15488),
//   P = 15488 s1 - 121^2 * 128^2 - exact integers throughout, so band test, contender bookkeeping, DELTA and the exact
// re-evaluation are search3_box_kernel's, unchanged: the kernels compute the same N.
// v_mfma_i32_16x16x64_i8: rows (M) = 16 consecutive target POSITIONS of one stripe, columns (N) = 16 PIXELS - a tile of
// 8 x 2 (two image rows share all but one of their 16 target rows, and an 8-wide tile wastes less of the diagonal band:
// pixel nx searches positions x + [lo, lo + W), so 8 pixels span W + 7 positions = two 16-position tiles for W <= 25,
// 56 % of the computed pairs candidates, against 37 % for 16 x 1), K = 4 target rows x 16 bytes per instruction.  Lane
// (m | n = l & 15, kb = l >> 4) supplies 16 bytes of target row rho = 4 i + kb at position m for the position operand and
// the same row of pixel n's window - window row j = rho - stripe - ny, eleven bytes and five zeros, all zeros where j is
// not 0..10 - for the pixel operand; both through the same byte map, so the product is exact whatever the instruction's
// internal k order.  The position operand of a tile (rows rho = 0..15: four fragments) serves all five stripes; the pixel
// operands depend on 4 i - stripe only (fifteen distinct fragments, registers for the whole pixel tile).  18 MFMAs per
// position tile and 1280 (pixel, position, stripe) pairs.  Accumulator lane l holds pixel l & 15 and positions
// 4 (l >> 4) .. + 3: one pixel per lane, so its constants (P, s1 - 15488, the acceptance limit) are per-lane scalars.
// LDS: target strip and pixel strip biased by 128 in FOUR byte-shifted copies (a 16-byte fragment at any byte offset o
// is four dwords of copy o & 3: an unaligned ds_read_b128 costs 7x an aligned one on gfx950), candidate statistics
// {-s2, sd2}, and the pixels' walk state (running maximum, limit, contender list): a pixel's candidates arrive in four
// lanes, which update the state one lane group after the other inside the rarely taken hit branch.
// Setup and the exact phase run one pixel per thread (64 x 4 tile), as in the box kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S4_TW = 64, S4_TH = 4;
constexpr int S4_TROWS = 18, S4_TPITCH = 172; // target strip: 4 pixel rows + 4 stripe steps + 10 window rows; 43 dwords per row
constexpr int S4_PROWS = 14, S4_PPITCH = 100; // pixel strip: 4 pixel rows + 10 window rows; 64 + 10 columns (+ alignment, + the fragment's tail)
constexpr int S4_ISROWS = 8, S4_ISP = 160;    // candidate statistics: pixel row + stripe, position
constexpr int S4_WMAX = 60;                   // widest displacement box of a workgroup (strip and statistics rows hold it)
constexpr int S4_T_OFF = 0, S4_TCOPY = S4_TROWS * S4_TPITCH;
constexpr int S4_P_OFF = S4_T_OFF + 4 * S4_TCOPY, S4_PCOPY = S4_PROWS * S4_PPITCH;
constexpr int S4_IS_OFF = (S4_P_OFF + 4 * S4_PCOPY + 15) & ~15;
constexpr int S4_ST_OFF = S4_IS_OFF + S4_ISROWS * S4_ISP * 8;
struct S4PixelState { // per pixel of the workgroup's tile (struct of arrays in LDS)
    float runmax[256], k1[256];
    uint32_t count[256];
    unsigned long long clist[256];
    int s1[256], lox[256], r0[256];
    uint32_t wx[256];
};
constexpr int S4_LDS_BYTES = S4_ST_OFF + (int)sizeof(S4PixelState);
constexpr int S4_QCAP = 12; // events a pixel's queue holds per tile (beyond: the pixel's whole corridor is re-evaluated exactly)
struct S4Merge {
    float runmax[64], g[4][64];
    uint32_t count[64];
    unsigned long long clist[64];
    uint32_t evn[16];
    uint2 ev[16 * S4_QCAP];
};
typedef int s4_i32x4 __attribute__((ext_vector_type(4)));

// `nrows` image rows from `row0`, `nd` dwords from byte column `col0` (a multiple of 4) -> four byte-shifted copies,
// biased by 128: dword D of copy c = image bytes col0 + 4 D + c .. + 3 (zero outside the image: those positions and
// pixels are never candidates)
__device__ __forceinline__ void s4_stage_copies(uint8_t *__restrict__ dst, int copy_bytes, int pitch, const uint8_t *__restrict__ img, int w, int h,
                                                int row0, int col0, int nrows, int nd)
{
    for (int u = (int)threadIdx.x; u < nrows * nd; u += 256) {
        const int r = u / nd, D = u - r * nd;
        const uint32_t d0 = load_dword_checked(img, w, h, row0 + r, col0 + 4 * D) ^ 0x80808080u;
        const uint32_t d1 = load_dword_checked(img, w, h, row0 + r, col0 + 4 * D + 4) ^ 0x80808080u;
        uint32_t *o = reinterpret_cast<uint32_t *>(dst + r * pitch) + D;
        o[0] = d0;
        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(o) + copy_bytes) = __builtin_amdgcn_alignbyte(d1, d0, 1);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(o) + 2 * copy_bytes) = __builtin_amdgcn_alignbyte(d1, d0, 2);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(o) + 3 * copy_bytes) = __builtin_amdgcn_alignbyte(d1, d0, 3);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256, 3) void search4_mfma_kernel(SearchJob ja, SearchJob jb)
{
    const SearchJob &j = this_job();
    const CorrParams &p = j.p;
    const uint8_t *__restrict__ const img1 = j.img1, *__restrict__ const img2 = j.img2;
    const uint2 *__restrict__ const stats1 = j.stats1, *__restrict__ const istats2 = j.stats2;
    unsigned long long *__restrict__ const contenders = j.contenders, *__restrict__ const counters = j.counters;
    uint32_t *__restrict__ const out = j.out;
    float *__restrict__ const out_score = j.out_score;
    __shared__ __attribute__((aligned(16))) uint8_t lds[S4_LDS_BYTES];
    __shared__ int bb[8]; // min dx, max dx, min loy, max loy, largest candidate count of a pixel, 1 = some pixel is no rectangle
    S4PixelState &st = *reinterpret_cast<S4PixelState *>(lds + S4_ST_OFF);
    __shared__ S4Merge mg_lds[4]; // one per wave: the four lanes' contender lists of a pixel meet here when its tile is done

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const TileId tid = xcd_tile();
    const int U0 = (int)tid.x * S4_TW;
    const uint32_t V0 = p.row0 + tid.y * S4_TH;
    const uint32_t x = (uint32_t)U0 + lane, y = V0 + w;
    const int xi = (int)x;
    const bool is_out = x < p.w1 && y < p.row1;
    if (threadIdx.x == 0) {
        bb[0] = 0x7FFFFFFF;
        bb[1] = -0x7FFFFFFF;
        bb[2] = 0x7FFFFFFF;
        bb[3] = -0x7FFFFFFF;
        bb[4] = 0;
        bb[5] = 0;
    }
    // ---- per-pixel setup (one pixel per thread): search3_box_kernel's, for its lean instantiation ---------------
    PixelSetup ps;
    ps.st1 = make_float2(0.0f, 1.0f);
    ps.e.cx = ps.e.cy = ps.e.ax = ps.e.ay = 0.0;
    ps.e.ox = ps.e.oy = 0;
    ps.r0 = ps.r1 = 0;
    const bool active = is_out && pixel_setup(p, x, y, stats1, j.range, ps);
    const Line &e = ps.e;
    const uint32_t r0 = ps.r0, r1 = ps.r1;
    const int cs = p.corridor_size;
    bool simple = true;
    int lox = 0, loy = 0;
    uint32_t wx = 0;
    if (active) {
        // the candidate set must be the rectangle [lox, lox + wx) x [loy, loy + 5): candidates advance along x (x2 == i
        // exactly), the five stripes are consecutive rows that do not depend on i (minor coefficient +-0)
        const bool major_x = e.ox == 0;
        uint32_t m0 = 0;
        bool consecutive = true;
        for (int off = -cs; off <= cs; off++) {
            const uint32_t mf = f64_to_u32_sat(floor((e.cy * (double)r0 + e.ay) + (double)(off * e.oy)));
            if (off == -cs) m0 = mf;
            consecutive = consecutive && mf == m0 + (uint32_t)(off + cs);
        }
        const uint32_t ilo = max(r0, (uint32_t)KERNEL_SIZE), ihi = min(r1, sat_sub_u32(p.w2, KERNEL_SIZE));
        const uint32_t nmaj = ihi > ilo ? ihi - ilo : 0u;
        simple = major_x && cs == 2 && e.cy == 0.0 && e.cx == 1.0 && e.ax == 0.0 && consecutive && (r1 - r0) <= CW_MAX_LEN && m0 < 0x40000000u &&
                 ilo < 0x40000000u;
        lox = (int)ilo - xi;
        wx = nmaj;
        loy = (int)m0 - (int)y;
    }
    const bool has = active && simple && wx > 0u;
    const int mnx = wave_min_i32(has ? lox : 0x7FFFFFFF), mxx = wave_max_i32(has ? lox + (int)wx - 1 : -0x7FFFFFFF);
    const int mny = wave_min_i32(has ? loy : 0x7FFFFFFF), mxy = wave_max_i32(has ? loy : -0x7FFFFFFF);
    const int need = wave_max_i32(has ? (int)min(wx * 5u, 0x3FFFFFFFu) : 0);
    const bool wave_odd = __any(active && !simple);
    __syncthreads(); // bb initialised
    if (lane == 0) {
        if (mxx >= mnx) {
            atomicMin(&bb[0], mnx);
            atomicMax(&bb[1], mxx);
            atomicMin(&bb[2], mny);
            atomicMax(&bb[3], mxy);
            atomicMax(&bb[4], need);
        }
        if (wave_odd) atomicOr(&bb[5], 1);
    }
    __syncthreads();
    const size_t pix = (size_t)y * p.w1 + x;
    const bool any_has = bb[1] >= bb[0];
    const int dx0 = bb[0], W = bb[1] - bb[0] + 1, LOY = bb[2];
    // one stripe origin for the whole workgroup, a box the strips hold, and a walk not much larger than the largest
    // pixel's own candidate set (disparity discontinuities)
    const bool eligible = !bb[5] && (!any_has || (bb[2] == bb[3] && W <= S4_WMAX && (long long)(W + 7) * 5 <= 3ll * bb[4] + 80));
    if (!eligible || !any_has) {
        if (is_out) {
            const bool fb = active && !eligible;
            if (!eligible) contenders[pix] = fb ? (CW_FALLBACK << 60) : 0ull;
            if (!fb) out[pix] = CELL_NONE;
        }
        if (!eligible && threadIdx.x == 0) worklist_push(j.declined, (uint32_t)U0 | (tid.y << 16) | 0x80000000u);
        return;
    }
    // ---- staging ---------------------------------------------------------------------------------------------
    const int TC0 = (U0 + dx0 - KERNEL_SIZE) & ~3;     // image column of byte 0 of the target strip
    const int TR0 = (int)V0 + LOY - KERNEL_SIZE;      // image row of its row 0
    const int PC0 = (U0 - KERNEL_SIZE) & ~3, PR0 = (int)V0 - KERNEL_SIZE;
    {
        const int tnd = min((U0 + dx0 - KERNEL_SIZE - TC0 + S4_TW + W + 38 + 3) >> 2, S4_TPITCH / 4 - 1); // bytes the fragments can touch
        s4_stage_copies(lds + S4_T_OFF, S4_TCOPY, S4_TPITCH, img2, (int)p.w2, (int)p.h2, TR0, TC0, S4_TROWS, tnd);
        s4_stage_copies(lds + S4_P_OFF, S4_PCOPY, S4_PPITCH, img1, (int)p.w1, (int)p.h1, PR0, PC0, S4_PROWS, S4_PPITCH / 4 - 1);
        // candidate statistics {-s2, sd2}: row = pixel row + stripe, entry = position - (U0 + dx0); centres outside the
        // image or skipped by the reference (mod.rs:430-441) get sd2 = +inf: their limit can never be reached
        const int isp = min(S4_TW + W + 32, S4_ISP);
        const int gu0 = U0 + dx0, gv0 = (int)V0 + LOY;
        for (int u = (int)threadIdx.x; u < isp * S4_ISROWS; u += 256) {
            const int r = u / isp, c = u - r * isp;
            const int gy = gv0 + r, gx = gu0 + c;
            uint2 v = make_uint2(0u, 0x7F800000u);
            if (gy >= 0 && gy < (int)p.h2 && gx >= 0 && gx < (int)p.w2) {
                const uint2 tt = istats2[(size_t)gy * p.w2 + (size_t)gx];
                if (tt.x & 0x80000000u) v = make_uint2(0u - (tt.x & 0x7FFFFFFFu), tt.y);
            }
            *reinterpret_cast<uint2 *>(lds + S4_IS_OFF + (size_t)(r * S4_ISP + c) * 8u) = v;
        }
    }
    const uint32_t s1 = has ? (stats1[pix].x & 0x7FFFFFFFu) : 0u;
    const float k1 = ps.st1.y * (float)(KERNEL_POINT_COUNT * KERNEL_POINT_COUNT); // 121*121*sd1
    const float thr_lo = p.threshold - S2_DELTA;
    {
        const uint32_t pi = threadIdx.x; // = w * 64 + lane
        st.runmax[pi] = -__builtin_inff();
        st.k1[pi] = k1;
        st.count[pi] = 0u;
        st.clist[pi] = 0ull;
        st.s1[pi] = (int)s1;
        st.lox[pi] = lox;
        st.r0[pi] = (int)r0;
        st.wx[pi] = has ? wx : 0u;
    }
    __syncthreads();

    // ---- the walk: wave w takes pixel tiles w, w + 4, w + 8, w + 12 of the sixteen 8 x 2 tiles ------------------
    uint32_t evaluated = 0;
    {
        const uint32_t n = lane & 15u, q = lane >> 4, nx = n & 7u, ny = n >> 3;
        for (uint32_t t = w; t < 16u; t += 4u) {
            const uint32_t tx = t & 7u, th = t >> 3;
            const uint32_t pi = (2u * th + ny) * 64u + 8u * tx + nx;
            const uint32_t pwx = st.wx[pi];
            const int plox = st.lox[pi];
            const int dmin = wave_min_i32(pwx ? plox : 0x7FFFFFFF), dmax = wave_max_i32(pwx ? plox + (int)pwx - 1 : -0x7FFFFFFF);
            if (dmax < dmin) continue; // (wave-uniform) nothing to search in this tile
            const int ntiles = (dmax - dmin + 1 + 7 + 15) >> 4;
            const int x0t = U0 + 8 * (int)tx;
            const int ps1 = st.s1[pi], P = 15488 * ps1 - 239878144, Qn = ps1 - 15488;
            const float kk1 = st.k1[pi];
            // acceptance band in the integer domain, as in the box kernel; lanes without candidates never pass
            float limk = pwx ? thr_lo * kk1 * (1.0f - 9.5367431640625e-7f) : __builtin_inff();
            S4Merge &mm = mg_lds[w];
            if (lane < 16u) mm.evn[lane] = 0u; // (the previous tile's readers are behind the wave barrier at its end)
            // pixel operands: fragment k <-> 4 i - stripe = k - 3: window row j = k - 3 + kb - ny of pixel n, i.e. strip row
            // 2 th + kb + k - 3 (ny cancels), eleven bytes and five zeros; zero where j is not 0..10
            s4_i32x4 G[15];
            {
                const int op = x0t + (int)nx - KERNEL_SIZE - PC0; // byte offset of the pixel's window in its strip rows
                const uint8_t *const base = lds + S4_P_OFF + (op & 3) * S4_PCOPY + (op >> 2) * 4;
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    const int jrow = k - 3 + (int)q - (int)ny, prow = 2 * (int)th + (int)q + k - 3;
                    const bool ok = jrow >= 0 && jrow <= 10;
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(base + (ok ? prow : 0) * S4_PPITCH);
                    G[k][0] = ok ? (int)src[0] : 0;
                    G[k][1] = ok ? (int)src[1] : 0;
                    G[k][2] = ok ? (int)(src[2] & 0x00FFFFFFu) : 0;
                    G[k][3] = 0;
                }
            }
            // position fragments: byte offset of position (pt, m) in its strip rows, statistics entry of (pt, 4 q)
            const int P0 = x0t + dmin; // position of (pt = 0, m = 0)
            const int ob = P0 + (int)n - KERNEL_SIZE - TC0;
            const uint8_t *posb = lds + S4_T_OFF + (ob & 3) * S4_TCOPY + (ob >> 2) * 4 + (2 * (int)th + (int)q) * S4_TPITCH;
            const uint8_t *isb = lds + S4_IS_OFF + (size_t)((2 * (int)th + (int)ny) * S4_ISP + (P0 - (U0 + dx0)) + 4 * (int)q) * 8u;
            const int dbase = dmin + 4 * (int)q - (int)nx; // displacement of this lane's first position in tile 0
            // position tiles from the middle outwards (the ranges are centred on the previous level's prediction)
            for (int ti = 0; ti < ntiles; ti++) {
                const int mid = ntiles >> 1, pt = (ti & 1) ? mid - ((ti + 1) >> 1) : mid + (ti >> 1);
                if (pt < 0 || pt >= ntiles) continue; // (ntiles even: the order 1, 2, 0, 3 ... visits index -1 / ntiles once)
                s4_i32x4 A[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(posb + 16 * pt + 4 * i * S4_TPITCH);
                    A[i][0] = (int)src[0];
                    A[i][1] = (int)src[1];
                    A[i][2] = (int)src[2];
                    A[i][3] = (int)src[3];
                }
                s4_i32x4 acc[5];
#pragma unroll
                for (int s = 0; s < 5; s++) acc[s] = s4_i32x4{0, 0, 0, 0};
                // stripe s, instruction i: pixel fragment k = 4 i - s + 3
                acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], G[3], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], G[2], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], G[1], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], G[0], acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], G[3], acc[4], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], G[7], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], G[6], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], G[5], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], G[4], acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], G[7], acc[4], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], G[11], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], G[10], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], G[9], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], G[8], acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], G[11], acc[4], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], G[14], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], G[13], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], G[12], acc[3], 0, 0, 0);
                const int d0 = dbase + 16 * pt; // displacement of acc[.][0]
                // stripes from the middle outwards: every record raises the limit
#pragma unroll
                for (int si = 0; si < 5; si++) {
                    const int s = si == 0 ? 2 : (si == 1 ? 1 : (si == 2 ? 3 : (si == 3 ? 0 : 4)));
                    const uint2 *isp = reinterpret_cast<const uint2 *>(isb + (size_t)(s * S4_ISP + 16 * pt) * 8u);
                    int num[4];
                    float sd[4], mg[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint2 is2 = isp[r];
                        num[r] = __mul24(acc[s][r], KERNEL_POINT_COUNT) + P + __mul24((int)is2.x, Qn);
                        sd[r] = __uint_as_float(is2.y);
                        mg[r] = __builtin_fmaf(-limk, sd[r], (float)num[r]);
                    }
                    if (COUNT) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if ((uint32_t)(d0 + r - plox) < pwx && sd[r] < __builtin_inff()) evaluated++;
                    }
                    const float margin = fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3]));
                    if (__any(margin >= 0.0f)) {
                        // Passing candidates are EVENTS: (score, code) appended to the pixel's queue in LDS (sixteen pixels
                        // x four lanes x twenty values per position tile: some lane has one at most steps, so what a lane
                        // does here has to be short - the contender bookkeeping waits until the tile is done).  The limit
                        // rises at once: a candidate with score g puts the maximum at g or above.
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (mg[r] >= 0.0f) {
                                const int d = d0 + r;
                                if ((uint32_t)(d - plox) < pwx && sd[r] < __builtin_inff() && (float)num[r] >= limk * sd[r]) {
                                    const float c1 = 1.0f / kk1; // as in the box kernel
                                    const float g = (float)num[r] * (c1 * __builtin_amdgcn_rcpf(sd[r]));
                                    const uint32_t code = ((uint32_t)s << 11) | (uint32_t)(x0t + (int)nx + d - st.r0[pi]);
                                    const uint32_t at = atomicAdd(&mm.evn[n], 1u);
                                    if (at < (uint32_t)S4_QCAP) mm.ev[n * S4_QCAP + at] = make_uint2(__float_as_uint(g), code);
                                    limk = fmaxf(limk, fmaxf(g - 2.0f * S2_DELTA, thr_lo) * kk1 * (1.0f - 9.5367431640625e-7f));
                                }
                            }
                        }
                        // the pixel's limit: the largest of its four lanes' (lanes l, l ^ 16, l ^ 32, l ^ 48)
                        {
                            const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(limk), __float_as_uint(limk), false, false);
                            limk = fmaxf(__uint_as_float(h[0]), __uint_as_float(h[1]));
                            const auto g16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(limk), __float_as_uint(limk), false, false);
                            limk = fmaxf(__uint_as_float(g16[0]), __uint_as_float(g16[1]));
                        }
                    }
                }
            }
            // ---- the tile is done: the pixels' events go through the box kernel's record() - lane (n, q) takes events
            // q, q + 4, ... of pixel n onto a list of its own (scores kept), then the four lists of a pixel are merged.
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float runmax = -__builtin_inff(), gs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            uint32_t count = 0;
            unsigned long long clist = 0ull;
            {
                const uint32_t nev = mm.evn[n];
                if (nev > (uint32_t)S4_QCAP) { // more events than the queue holds: the whole corridor, exactly
                    count = (uint32_t)S2_K + 1u;
                    runmax = __builtin_inff();
                }
                for (uint32_t i = q; i < min(nev, (uint32_t)S4_QCAP); i += 4u) {
                    const uint2 evv = mm.ev[n * S4_QCAP + i];
                    const float g = __uint_as_float(evv.x);
                    const float lim = fmaxf(runmax - 2.0f * S2_DELTA, thr_lo);
                    if (g >= lim) {
                        if (g > runmax + 2.0f * S2_DELTA) { // everything recorded so far is out of the band
                            count = 0;
                            clist = 0ull;
                        }
                        runmax = fmaxf(runmax, g);
                        if (count < (uint32_t)S2_K) {
                            clist |= (unsigned long long)evv.y << (15u * count);
                            if (count == 0u) gs[0] = g;
                            else if (count == 1u) gs[1] = g;
                            else if (count == 2u) gs[2] = g;
                            else gs[3] = g;
                        }
                        count = min(count + 1u, (uint32_t)S2_K + 1u);
                    }
                }
            }
            // ---- the tile is done: merge the four lanes' lists of every pixel (through LDS, the wave's own slots) -------
            // Everything within 2 delta of the pixel's maximum was within 2 delta of its lane's running maximum when it was
            // met, so it is on that lane's list unless a reset dropped it - and a reset only drops what is more than
            // 2 delta below a score that is itself not above the maximum.  A lane whose list overflowed inside the band
            // sends the pixel to the whole-corridor evaluation, as an overflowing pixel list does.
            {
                mm.runmax[lane] = runmax;
                mm.count[lane] = count;
                mm.clist[lane] = clist;
#pragma unroll
                for (int c = 0; c < 4; c++) mm.g[c][lane] = gs[c];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (q == 0u) {
                    float gmax = -__builtin_inff();
#pragma unroll
                    for (uint32_t qq = 0; qq < 4u; qq++) gmax = fmaxf(gmax, mm.runmax[n + 16u * qq]);
                    const float band = fmaxf(gmax - 2.0f * S2_DELTA, thr_lo);
                    uint32_t cnt = 0;
                    unsigned long long cl = 0ull;
#pragma unroll
                    for (uint32_t qq = 0; qq < 4u; qq++) {
                        const uint32_t lc = mm.count[n + 16u * qq];
                        const unsigned long long lcl = mm.clist[n + 16u * qq];
                        if (lc > (uint32_t)S2_K && mm.runmax[n + 16u * qq] >= band) cnt = (uint32_t)S2_K + 1u; // overflowed inside the band
#pragma unroll
                        for (uint32_t c = 0; c < (uint32_t)S2_K; c++) {
                            if (c < min(lc, (uint32_t)S2_K) && mm.g[c][n + 16u * qq] >= band) {
                                if (cnt < (uint32_t)S2_K) cl |= ((lcl >> (15u * c)) & 0x7FFFull) << (15u * cnt);
                                cnt = min(cnt + 1u, (uint32_t)S2_K + 1u);
                            }
                        }
                    }
                    st.runmax[pi] = gmax;
                    st.count[pi] = cnt;
                    st.clist[pi] = cl;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();

    // ---- exact re-evaluation of the contenders (one pixel per thread): search3_box_kernel's ----------------------
    unsigned long long word = 0ull;
    uint2 cell = make_uint2(CELL_NONE, 0x7FC00000u);
    uint32_t multi = 0, whole = 0, exact_evals = 0;
    if (has) {
        const uint32_t count = st.count[threadIdx.x];
        const unsigned long long clist = st.clist[threadIdx.x];
        const float runmax = st.runmax[threadIdx.x];
        if (count > (uint32_t)S2_K) {
            word = CW_WHOLE << 60;
            whole = 1;
        } else if (count > 0u) {
            multi = count > 1 ? 1u : 0u;
            bool have = false;
            float bcorr = 0.0f;
            uint32_t bxy = 0, bcode = 0;
            const float avg1 = ps.st1.x, sdev1 = ps.st1.y;
            const uint8_t *base1 = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (x - KERNEL_SIZE);
            uint32_t ecount = count;
            if (!p.need_scores && count == 1u && runmax >= p.threshold + S2_DELTA) {
                const uint32_t code = (uint32_t)clist & 0x7FFFu;
                const CandXY c = candidate_xy(e, r0 + (code & 0x7FFu), (int)(code >> 11) - cs);
                cell = make_uint2(c.x | (c.y << 16), __float_as_uint(runmax));
                ecount = 0u;
            }
            for (uint32_t jj = 0; jj < ecount; jj++) {
                const uint32_t code = (uint32_t)(clist >> (15u * jj)) & 0x7FFFu;
                const CandXY c = candidate_xy(e, r0 + (code & 0x7FFu), (int)(code >> 11) - cs);
                const uint2 is2 = istats2[(size_t)c.y * p.w2 + c.x];
                const float avg2 = (float)(is2.x & 0x7FFFFFFFu) / (float)KERNEL_POINT_COUNT; // == compute_point_avg
                const float sdev2 = __uint_as_float(is2.y);
                const uint8_t *base2 = img2 + (size_t)(c.y - KERNEL_SIZE) * p.w2 + (c.x - KERNEL_SIZE);
                float corr = 0.0f;
#pragma unroll 1
                for (int rb = 0; rb < 12; rb += 4) {
                    Row12 av[4], bv[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int rr = min(rb + r, KERNEL_WIDTH - 1);
                        av[r] = load_row12(base1 + (size_t)rr * p.w1);
                        bv[r] = load_row12(base2 + (size_t)rr * p.w2);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (rb + r < KERNEL_WIDTH) corr = row_corr_acc(corr, av[r], bv[r], avg1, avg2);
                }
                corr /= sdev1 * sdev2 * (float)KERNEL_POINT_COUNT; // mod.rs:454
                exact_evals++;
                if (corr >= p.threshold && (!have || corr > bcorr || (corr == bcorr && code < bcode))) { // mod.rs:456-464
                    have = true;
                    bcorr = corr;
                    bcode = code;
                    bxy = c.x | (c.y << 16);
                }
            }
            if (have) cell = make_uint2(bxy, __float_as_uint(bcorr));
        }
    }
    if (is_out && !whole) store_cell(p, out, out_score, pix, cell);
    if (counters) {
        uint32_t v0 = evaluated, v1 = exact_evals, v2 = multi, v3 = whole;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            v0 += __shfl_down(v0, sft, 64);
            v1 += __shfl_down(v1, sft, 64);
            v2 += __shfl_down(v2, sft, 64);
            v3 += __shfl_down(v3, sft, 64);
        }
        if (lane == 0) {
            if (v0) atomicAdd(&counters[0], (unsigned long long)v0);
            if (v1) atomicAdd(&counters[1], (unsigned long long)v1);
            if (v2) atomicAdd(&counters[2], (unsigned long long)v2);
            if (v3) atomicAdd(&counters[3], (unsigned long long)v3);
        }
    }
    {
        const int any_whole = __syncthreads_or(whole ? 1 : 0);
        if (any_whole && is_out) contenders[pix] = word;
        if (threadIdx.x == 0 && any_whole) worklist_push(j.whole, (uint32_t)U0 | (tid.y << 16) | 0x80000000u);
    }
}

// ---- kernel B: exact re-evaluation of the contenders (mod.rs:442-464), in corridor order --------------
__device__ __forceinline__ void search2_exact_tile(const CorrParams &p, const uint8_t *__restrict__ img1,
                                                   const uint8_t *__restrict__ img2, const uint2 *__restrict__ stats1,
                                                   const uint2 *__restrict__ istats2, const uint32_t *__restrict__ range,
                                                   const unsigned long long *__restrict__ contenders,
                                                   uint32_t *__restrict__ out, float *__restrict__ out_score,
                                                   unsigned long long *__restrict__ counters, PixTile tl)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x, y;
    const bool in_image = tile_pixel(tl, x, y) && x < p.w1 && y < p.row1;
    // Only pixels the filter kernel marked CW_WHOLE are handled here (the filter kernel settles everything
    // else itself); a wave without such a pixel leaves after reading its contender words.
    const bool interior =
        in_image && x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < p.w1 && y + KERNEL_SIZE < p.h1;
    const unsigned long long word = in_image ? contenders[(size_t)y * p.w1 + x] : 0ull;
    if (!__any((uint32_t)(word >> 60) == (uint32_t)CW_WHOLE)) return;
    const bool mine = interior && (uint32_t)(word >> 60) == (uint32_t)CW_WHOLE;
    float2 st1v = make_float2(0.0f, 0.0f);
    uint32_t rg = 0;
    Row12 arow[KERNEL_WIDTH];
    if (mine) {
        st1v = stats_of(stats1[(size_t)y * p.w1 + x]);
        if (!p.first_pass) rg = range[(size_t)y * p.w1 + x];
        const uint8_t *base = img1 + (size_t)(y - KERNEL_SIZE) * p.w1 + (x - KERNEL_SIZE);
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) arow[r] = load_row12(base + (size_t)r * p.w1);
    } else {
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) arow[r].a = arow[r].b = arow[r].c = 0u;
    }
    const uint32_t count = (uint32_t)(word >> 60);
    uint32_t evaluated = 0, exact_evals = 0;
    uint2 cell = make_uint2(CELL_NONE, 0x7FC00000u);
    PixelSetup ps;
    bool go = false;
    if (interior && count == (uint32_t)CW_WHOLE) { // pixel_setup (mod.rs:321-364) on the values loaded above
        ps.st1 = st1v;
        ps.e = epipolar_line(p, x, y);
        go = finite_f32(ps.st1.y) && !(fabsf(ps.st1.y) < p.min_stdev) && line_finite(ps.e);
        ps.r0 = KERNEL_SIZE;
        ps.r1 = corridor_end_of(p, ps.e);
        if (!p.first_pass) {
            go = go && rg != RANGE_NONE;
            ps.r0 = rg & 0xFFFFu;
            ps.r1 = rg >> 16;
        }
        go = go && ps.r0 < ps.r1;
    }
    if (go) {
        const int cs = p.corridor_size;
        const uint32_t len = ps.r1 - ps.r0;
        // compute_point_data deltas (mod.rs:727-731); avg identical to stats1.x
        float d1[KERNEL_POINT_COUNT];
#pragma unroll
        for (int r = 0; r < KERNEL_WIDTH; r++) {
#pragma unroll
            for (int c = 0; c < KERNEL_WIDTH; c++) {
                const uint32_t wv = c < 4 ? arow[r].a : (c < 8 ? arow[r].b : arow[r].c);
                d1[r * KERNEL_WIDTH + c] = byte_f32(wv, c & 3) - ps.st1.x;
            }
        }
        bool have = false;
        float bcorr = 0.0f;
        uint32_t bx = 0, by = 0;
        auto exact_candidate = [&](uint32_t sidx, uint32_t di, bool count_it) {
            const CandXY c = candidate_xy(ps.e, ps.r0 + di, (int)sidx - cs);
            if (!candidate_in_bounds(p, c)) return;
            const uint2 is2 = istats2[(size_t)c.y * p.w2 + c.x];
            if (!(is2.x & 0x80000000u)) return; // stdev2 non-finite or < min_stdev (mod.rs:437-441)
            if (count_it) evaluated++;
            exact_evals++;
            const float avg2 = (float)(is2.x & 0x7FFFFFFFu) / (float)KERNEL_POINT_COUNT; // == compute_point_avg
            const float stdev2 = __uint_as_float(is2.y);
            float corr = 0.0f;
            const uint8_t *base = img2 + (size_t)(c.y - KERNEL_SIZE) * p.w2 + (c.x - KERNEL_SIZE);
#pragma unroll
            for (int r = 0; r < KERNEL_WIDTH; r++) {
                const Row12 row = load_row12(base + (size_t)r * p.w2);
#pragma unroll
                for (int cc = 0; cc < KERNEL_WIDTH; cc++) {
                    const uint32_t wv = cc < 4 ? row.a : (cc < 8 ? row.b : row.c);
                    const float delta2 = byte_f32(wv, cc & 3) - avg2;
                    corr += d1[r * KERNEL_WIDTH + cc] * delta2;
                }
            }
            corr /= ps.st1.y * stdev2 * (float)KERNEL_POINT_COUNT; // mod.rs:454
            if (corr >= p.threshold && (!have || corr > bcorr)) {  // mod.rs:456-464
                have = true;
                bcorr = corr;
                bx = c.x;
                by = c.y;
            }
        };
        if (count >= (uint32_t)CW_WHOLE) {
            for (uint32_t sidx = 0; sidx < (uint32_t)(2 * cs + 1); sidx++)
                for (uint32_t di = 0; di < len; di++) exact_candidate(sidx, di, true);
        } else {
            for (uint32_t j = 0; j < count; j++) {
                const uint32_t code = (uint32_t)(word >> (15u * j)) & 0x7FFFu;
                exact_candidate(code >> 11, code & 0x7FFu, false);
            }
        }
        if (have) cell = make_uint2(bx | (by << 16), __float_as_uint(bcorr));
    }
    if (in_image && count == (uint32_t)CW_WHOLE) store_cell(p, out, out_score, (size_t)y * p.w1 + x, cell); // everything else was settled by the filter kernel
    if (counters) {
        uint32_t v0 = evaluated, v1 = exact_evals;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            v0 += __shfl_down(v0, sft, 64);
            v1 += __shfl_down(v1, sft, 64);
        }
        if (lane == 0) {
            if (v0) atomicAdd(&counters[0], (unsigned long long)v0);
            if (v1) atomicAdd(&counters[1], (unsigned long long)v1);
        }
    }
}

// Search version 3, everything the box kernel left behind, in ONE persistent grid over its two work lists:
// the tiles it declined go through the candidate filter and then, for their own > 4-contender pixels, straight
// through the whole-corridor evaluation (each lane reads back only the contender word it wrote itself); the tiles
// where the box kernel found such pixels only need the latter.
template <bool COUNT>
__global__ __launch_bounds__(256, 2) void search3_fallback_kernel(SearchJob ja, SearchJob jb, int skip_exact, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    const SearchJob &j = this_job();
    const CorrParams &p = j.p;
    const uint32_t nd = *j.declined.count, nw = *j.whole.count;
    for (uint32_t t = blockIdx.x; t < nd; t += gridDim.x) {
        const PixTile tl = tile_of_entry(j.declined.items[t], p.row0);
        const bool any_whole = search2_filter_tile<COUNT>(p, j.img1, j.img2, j.stats1, j.stats1, j.stats2, j.range, j.contenders,
                                                          j.out, j.out_score, j.counters, 1, tl, WorkList{nullptr, nullptr}, dyn_lds, lds_bytes);
        __syncthreads(); // the tile's LDS is reused by the next one
        if (!skip_exact && any_whole)
            search2_exact_tile(p, j.img1, j.img2, j.stats1, j.stats2, j.range, j.contenders, j.out, j.out_score, j.counters, tl);
    }
    if (skip_exact) return;
    for (uint32_t t = blockIdx.x; t < nw; t += gridDim.x)
        search2_exact_tile(p, j.img1, j.img2, j.stats1, j.stats2, j.range, j.contenders, j.out, j.out_score, j.counters,
                           tile_of_entry(j.whole.items[t], p.row0));
}

// LDS per workgroup of the candidate filter.  A 64x4 tile's candidate box is as tall as the lines are steep: for
// an affine F with row-major lines of slope <= 0.25 the 40 KB budget (3 workgroups per CU) always fits; column-major
// lines, steeper ones and perspective F (per-pixel lines) get 64 KB, without which most of their tiles would fall
// through to the whole-corridor kernel.
static uint32_t search2_lds_bytes(const CorrParams &p)
{
    const double *F = p.F;
    const bool affine_form = F[0] == 0.0 && F[1] == 0.0 && F[3] == 0.0 && F[4] == 0.0;
    const bool shallow = affine_form && std::fabs(F[2]) <= 0.25 * std::fabs(F[5]);
    return shallow ? (uint32_t)S2_LDS_BYTES : (uint32_t)S2_LDS_BYTES_STEEP;
}

constexpr int LIST_GRID = 512; // persistent workgroups of the work-list kernels, all jobs of a launch together

// The candidate filter over every tile; tiles with whole-corridor pixels queue themselves on the job's whole list
// for search3_fallback_kernel (launched behind it; its declined list stays empty).
void launch_search2_filter(const SearchJob *jobs, int n, hipStream_t s)
{
    uint32_t gx = 0, gy = 0, lds = 0;
    for (int i = 0; i < n; i++) {
        const CorrParams &p = jobs[i].p;
        if (!job_active(jobs[i])) continue;
        gx = std::max(gx, (p.w1 + 63) / 64);
        gy = std::max(gy, (p.row1 - p.row0 + 3) / 4);
        lds = std::max(lds, search2_lds_bytes(p));
    }
    if (!gx || !gy) return;
    // the first pass, one workgroup per (tile, stripe) where every job has the scratch for it (search2_split_words)
    bool split = true;
    for (int i = 0; i < n; i++)
        split = split && job_active(jobs[i]) && jobs[i].p.first_pass && jobs[i].split != nullptr &&
                jobs[i].p.corridor_size == jobs[0].p.corridor_size && (jobs[i].p.w1 + 63) / 64 == gx && (jobs[i].p.row1 - jobs[i].p.row0 + 3) / 4 == gy;
    if (split) {
        const uint32_t NS = (uint32_t)(2 * jobs[0].p.corridor_size + 1);
        const dim3 wgrid(gx * NS, gy, (unsigned)n), mgrid(gx, gy, (unsigned)n);
        if (jobs[0].counters) {
            hipLaunchKernelGGL((search2_filter_split_kernel<true, 1>), wgrid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds, gx);
            hipLaunchKernelGGL((search2_filter_split_kernel<true, 2>), mgrid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds, gx);
        } else {
            hipLaunchKernelGGL((search2_filter_split_kernel<false, 1>), wgrid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds, gx);
            hipLaunchKernelGGL((search2_filter_split_kernel<false, 2>), mgrid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds, gx);
        }
        return;
    }
    const dim3 grid(gx, gy, (unsigned)n);
    if (jobs[0].counters)
        hipLaunchKernelGGL(search2_filter_kernel<true>, grid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds);
    else
        hipLaunchKernelGGL(search2_filter_kernel<false>, grid, dim3(256), lds, s, jobs[0], jobs[n - 1], lds);
}

size_t search2_split_words(uint32_t w, uint32_t rows, uint32_t stripes)
{
    const size_t tiles = (size_t)((w + 63) / 64) * ((rows + 3) / 4);
    return tiles * stripes * 256u * 4u; // one 32-byte record per (tile, stripe, thread)
}

void launch_search3_fallback(const SearchJob *jobs, int n, bool skip_exact, hipStream_t s)
{
    uint32_t lds = 0;
    bool any = false;
    size_t entries = 0; // the most entries one of the lists can hold (search3_worklist_capacity, for this pass's rows)
    for (int i = 0; i < n; i++) {
        if (!job_active(jobs[i])) continue;
        any = true;
        lds = std::max(lds, search2_lds_bytes(jobs[i].p));
        entries = std::max(entries, search3_worklist_capacity(jobs[i].p.w1, jobs[i].p.row1 - jobs[i].p.row0));
    }
    if (!any) return;
    // As many workgroups as the chip holds at once (two per CU: 40 - 64 KB of LDS each), over both jobs; they loop over the
    // lists.  More only queue behind those - and an EMPTY list, the usual case behind the box walk of a rectified pair, costs
    // the dispatch of the grid: 4.5 us for 2 x 34 workgroups, 6 us for 2 x 256, 11.5 us for the 2 x 768 of round 4, per level.
    // (The small levels' lists cannot hold even that many tiles: 64^2 .. 256^2 launch 2 x 34 .. 2 x 325 at most.)
    const size_t want = (size_t)LIST_GRID / (n == 2 ? 2 : 1);
    const dim3 grid((unsigned)std::min<size_t>(want, entries), 1, (unsigned)n);
    if (jobs[0].counters)
        hipLaunchKernelGGL(search3_fallback_kernel<true>, grid, dim3(256), lds, s, jobs[0], jobs[n - 1], skip_exact ? 1 : 0, lds);
    else
        hipLaunchKernelGGL(search3_fallback_kernel<false>, grid, dim3(256), lds, s, jobs[0], jobs[n - 1], skip_exact ? 1 : 0, lds);
}

// fallback_skip_exact >= 0: where the two directions go out on two streams (the stepped instantiations with a side
// stream), each direction's fallback kernel is launched here too, behind its own box kernel - the first direction's few,
// slow tiles (a declined tile takes ~0.1 ms of one workgroup) then run under the second direction's box walk instead of
// after it.  Returns whether the fallback kernels went out (else the caller launches them).
bool launch_search3_box(const SearchJob *jobs, int n, bool stepped_lines, bool transposed, int form, hipStream_t s,
                        hipStream_t side, hipEvent_t fork, hipEvent_t join, int fallback_skip_exact)
{
    bool fallbacks_out = false;
    // form (rectified affine pairs only - lines that never step, row-major): 0 = the box walk, one column per lane; 1 = the
    // filter on the matrix pipe (search version 5); 2 = the box walk with two columns per lane (search3_box2_kernel)
    const bool mfma = form == 1;
    if (form == 2 && !stepped_lines && !transposed) {
        uint32_t gx = 0, gy = 0;
        for (int i = 0; i < n; i++) {
            const CorrParams &p = jobs[i].p;
            if (!job_active(jobs[i])) continue;
            gx = std::max(gx, (p.w1 + P2_OUT - 1) / P2_OUT);
            gy = std::max(gy, (p.row1 - p.row0 + 3) / 4);
        }
        if (!gx || !gy) return false;
        const dim3 grid(gx, gy, (unsigned)n);
        if (jobs[0].counters)
            hipLaunchKernelGGL(search3_box2_kernel<true>, grid, dim3(256), 0, s, jobs[0], jobs[n - 1]);
        else
            hipLaunchKernelGGL(search3_box2_kernel<false>, grid, dim3(256), 0, s, jobs[0], jobs[n - 1]);
        return false;
    }
    // rectified affine pairs under search version 5: the filter on the matrix pipe (search4_mfma_kernel); a pass that counts
    // candidates (profiling) takes the box kernel - the matrix-pipe walk does not know, value by value, which pixel ends up
    // re-evaluating its whole corridor
    if (mfma && !stepped_lines && !transposed && !jobs[0].counters) {
        uint32_t gx = 0, gy = 0;
        for (int i = 0; i < n; i++) {
            const CorrParams &p = jobs[i].p;
            if (!job_active(jobs[i])) continue;
            gx = std::max(gx, (p.w1 + S4_TW - 1) / S4_TW);
            gy = std::max(gy, (p.row1 - p.row0 + S4_TH - 1) / S4_TH);
        }
        if (!gx || !gy) return false;
        const dim3 grid(gx, gy, (unsigned)n);
        hipLaunchKernelGGL(search4_mfma_kernel<false>, grid, dim3(256), 0, s, jobs[0], jobs[n - 1]);
        return false;
    }
    // lanes along x, 4 rows per workgroup - or, transposed, lanes along y and 4 columns per workgroup
    uint32_t gx = 0, gy = 0;
    for (int i = 0; i < n; i++) {
        const CorrParams &p = jobs[i].p;
        if (!job_active(jobs[i])) continue;
        gx = std::max(gx, transposed ? (p.row1 - p.row0 + S3_OUT - 1) / S3_OUT : (p.w1 + S3_OUT - 1) / S3_OUT);
        gy = std::max(gy, transposed ? (p.w1 + 3) / 4 : (p.row1 - p.row0 + 3) / 4);
    }
    if (!gx || !gy) return false;
    const dim3 grid(gx, gy, (unsigned)n);
    size_t lean_pad = 0;
#ifdef CVHIP_ABLATIONS
    // (occupancy experiments: CVHIP_LEAN_LDS_PAD bytes of unused dynamic LDS per workgroup of the lean instantiations)
    if (const char *e = getenv("CVHIP_LEAN_LDS_PAD")) lean_pad = (size_t)atoi(e);
#endif
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(256), lean_pad, s, jobs[0], jobs[n - 1]); };
    auto launch_each = [&](auto kernel) { // the stepped instantiations: one launch per job (see search3_box_single_kernel)
        const bool forked = side && fork && join && n == 2 && job_active(jobs[0]) && job_active(jobs[1]) &&
                            hipEventRecord(fork, s) == hipSuccess && hipStreamWaitEvent(side, fork, 0) == hipSuccess;
        for (int i = 0; i < n; i++)
            if (job_active(jobs[i])) {
                hipLaunchKernelGGL(kernel, dim3(gx, gy, 1), dim3(256),
                                   search3_step_lds_bytes(jobs[i].p.box_pd, jobs[i].p.box_sh, jobs[i].p.box_wide != 0),
                                   forked && i == 1 ? side : s, jobs[i].p,
                                   jobs[i].img1, jobs[i].img2, jobs[i].stats1, jobs[i].stats1, jobs[i].stats2,
                                   (const uint32_t *)jobs[i].range, jobs[i].contenders, jobs[i].out, jobs[i].out_score, jobs[i].counters,
                                   jobs[i].declined, jobs[i].whole);
                if (forked && fallback_skip_exact >= 0) {
                    launch_search3_fallback(&jobs[i], 1, fallback_skip_exact != 0, i == 1 ? side : s);
                    fallbacks_out = true;
                }
            }
        if (forked) { // (an error here surfaces at the caller's hipGetLastError / the next synchronisation)
            (void)hipEventRecord(join, side);
            (void)hipStreamWaitEvent(s, join, 0);
        }
    };
    const bool wide = jobs[0].p.box_wide != 0;
    const int variant = (jobs[0].counters ? 4 : 0) | (stepped_lines ? 2 : 0) | (transposed ? 1 : 0);
    switch (variant) {
    case 0: launch(search3_box_kernel<false, false, false>); break;
    case 1: launch(search3_box_kernel<false, false, true>); break;
    case 2: wide ? launch_each(search3_box_single_kernel<false, true, false, true>) : launch_each(search3_box_single_kernel<false, true, false, false>); break;
    case 3: wide ? launch_each(search3_box_single_kernel<false, true, true, true>) : launch_each(search3_box_single_kernel<false, true, true, false>); break;
    case 4: launch(search3_box_kernel<true, false, false>); break;
    case 5: launch(search3_box_kernel<true, false, true>); break;
    case 6: wide ? launch_each(search3_box_single_kernel<true, true, false, true>) : launch_each(search3_box_single_kernel<true, true, false, false>); break;
    default: wide ? launch_each(search3_box_single_kernel<true, true, true, true>) : launch_each(search3_box_single_kernel<true, true, true, false>); break;
    }
    return fallbacks_out;
}

size_t search3_worklist_capacity(uint32_t max_w, uint32_t max_h)
{
    // every box tile can be declined once and report CW_WHOLE once; every 64-wide tile can report CW_WHOLE once
    const size_t along_x = (size_t)((max_w + S3_OUT - 1) / S3_OUT) * ((max_h + 3) / 4 + 1);
    const size_t along_y = (size_t)((max_h + S3_OUT - 1) / S3_OUT) * ((max_w + 3) / 4 + 1); // transposed tiles
    return along_x > along_y ? along_x : along_y;
}

// ---------------------------------------------------------------------------------------------
// cross_check: cross_check_filter / cross_check_point (mod.rs:552-624) in level coordinates.
// search_area = 4 * round(1/scale) full-res cells == 4 level cells on both grids, and a level
// match (x2, y2) is the full-res match (x2, y2) << k, so the window test reduces to +-4 level
// cells.  Each thread owns one cell of `own` and only reads `other`.
// ---------------------------------------------------------------------------------------------
constexpr int CC_ROWS = 1; // rows per thread (more than one only lengthens the chain of dependent round trips)

struct CrossJob { // one direction's cross-check: `own` is filtered against `other` (match planes)
    uint32_t *own;
    const uint32_t *other;
    uint32_t ow, oh, rw, rh, row0; // oh = end of the row range handled, row0 its start
};

// One or both directions of a level in one launch (blockIdx.z).  The two filters may run side by side: a
// match's supporters are exactly the matches it supports, so neither ever removes a cell the other one needs and
// each decision depends on the UNFILTERED other grid only.
// zero_words (optional): eight u32 cleared by the first threads - the work-list counts of the NEXT level's search passes
// (cvhip_ctx_set_stats_ahead: the statistics kernel that otherwise clears them runs on another stream)
__global__ __launch_bounds__(256) void cross_check_kernel(CrossJob ja, CrossJob jb, uint32_t *__restrict__ zero_words)
{
    if (zero_words && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) zero_words[threadIdx.x] = 0u;
    const CrossJob &job = blockIdx.z == 0 ? ja : jb;
    uint32_t *__restrict__ own = job.own;
    const uint32_t *__restrict__ other = job.other;
    const uint32_t ow = job.ow, oh = job.oh, rw = job.rw, rh = job.rh, row0 = job.row0;
    // oh = end of the row range handled by this launch, row0 its start
    const TileId tid = xcd_tile();
    static_assert(CC_ROWS == 1, "one cell per thread");
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t x = tid.x * 64 + lane;
    const uint32_t y = row0 + tid.y * 4 + (threadIdx.x >> 6); // (uniform in the wave)
    const uint32_t sa = CROSS_CHECK_SEARCH_AREA;
    const bool in = x < ow && y < oh;
    const uint32_t cell = in ? own[(size_t)y * ow + x] : CELL_NONE;
    // The result is an existence test (mod.rs:613-623 returns true at the first hit), so the scan
    // order is free: probe the window centre first — a consistent pair of matches points straight
    // back — and fall back to the full scan only when that fails.
    const uint32_t pmx = cell & 0xFFFFu, pmy = cell >> 16;
    const uint32_t probe = (cell != CELL_NONE && pmx < rw && pmy < rh) ? other[(size_t)pmy * rw + pmx] : CELL_NONE;
    // does reverse match rm lie within +-sa of own cell (cx, cy)?
    auto points_back = [&](uint32_t rm, uint32_t cx, uint32_t cy) {
        const uint32_t rx = rm & 0xFFFFu, ry = rm >> 16;
        return rm != CELL_NONE && rx >= sat_sub_u32(cx, sa) && rx < cx + sa + 1 && ry >= sat_sub_u32(cy, sa) && ry < cy + sa + 1;
    };
    bool found = cell != CELL_NONE && points_back(probe, x, y);
    // The cells whose probe failed: their (2 sa + 1)^2 windows of the other grid are scanned by the WHOLE wave, one cell
    // of the window per lane (81 cells: two loads per lane), instead of by the failing lane alone while the others wait -
    // a wave with a single failing lane used to issue 81 load instructions for it, and nearly every wave has one; the
    // kernel's time was those instructions (it sat at the texture addresser's rate, VALU busy 0.18).  Two failing cells
    // per round, so that four loads are in flight.
    constexpr uint32_t CCW = 2 * CROSS_CHECK_SEARCH_AREA + 1, CCN = CCW * CCW;
    static_assert(CCN > 64 && CCN <= 128, "two window cells per lane");
    const uint32_t wy0 = lane / CCW, wx0 = lane - wy0 * CCW, t1 = lane + 64u, wy1 = t1 / CCW, wx1 = t1 - wy1 * CCW;
    unsigned long long todo = __ballot(cell != CELL_NONE && !found);
    while (todo) {
        const int s0 = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const bool two = todo != 0ull;
        const int s1 = two ? (int)__builtin_ctzll(todo) : s0;
        if (two) todo &= todo - 1ull;
        uint32_t rm[2][2], cxs[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int src = q ? s1 : s0;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cell, src);
            const uint32_t mx = c & 0xFFFFu, my = c >> 16;
            const uint32_t min_x = min(sat_sub_u32(mx, sa), rw), max_x = min(mx + sa + 1, rw);
            const uint32_t min_y = min(sat_sub_u32(my, sa), rh), max_y = min(my + sa + 1, rh);
            cxs[q] = tid.x * 64 + (uint32_t)src;
            const uint32_t ax = min_x + wx0, ay = min_y + wy0, bx = min_x + wx1, by = min_y + wy1;
            rm[q][0] = (ax < max_x && ay < max_y) ? other[(size_t)ay * rw + ax] : CELL_NONE;
            rm[q][1] = (t1 < CCN && bx < max_x && by < max_y) ? other[(size_t)by * rw + bx] : CELL_NONE;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const bool any = __ballot(points_back(rm[q][0], cxs[q], y) || points_back(rm[q][1], cxs[q], y)) != 0ull;
            if ((int)lane == (q ? s1 : s0)) found = any;
        }
    }
    if (cell != CELL_NONE && !found) own[(size_t)y * ow + x] = CELL_NONE; // (the score of a None cell is never looked at)
}

void launch_cross_check(uint32_t *own, const uint32_t *other, uint32_t ow, uint32_t oh, uint32_t rw, uint32_t rh,
                        uint32_t row0, uint32_t row1, hipStream_t s)
{
    row1 = min(row1, oh);
    if (row1 <= row0) return;
    dim3 grid((ow + 63) / 64, (row1 - row0 + 4 * CC_ROWS - 1) / (4 * CC_ROWS), 1);
    const CrossJob j{own, other, ow, row1, rw, rh, row0};
    hipLaunchKernelGGL(cross_check_kernel, grid, dim3(256), 0, s, j, j, (uint32_t *)nullptr);
}

// forward and reverse cross-check of a level in one launch
void launch_cross_check_pair(uint32_t *fwd, uint32_t *rev, uint32_t fw, uint32_t fh, uint32_t rw, uint32_t rh, uint32_t f_row0,
                             uint32_t f_row1, uint32_t r_row0, uint32_t r_row1, hipStream_t s, uint32_t *zero_words)
{
    f_row1 = f_row1 < fh ? f_row1 : fh;
    r_row1 = r_row1 < rh ? r_row1 : rh;
    const uint32_t rows_f = f_row1 > f_row0 ? f_row1 - f_row0 : 0u, rows_r = r_row1 > r_row0 ? r_row1 - r_row0 : 0u;
    const uint32_t rows_max = rows_f > rows_r ? rows_f : rows_r;
    if (rows_max == 0) {
        if (zero_words) (void)hipMemsetAsync(zero_words, 0, 8 * sizeof(uint32_t), s);
        return;
    }
    // an empty range is expressed as oh = row0 (every thread of that slice exits)
    const CrossJob jf{fwd, rev, fw, rows_f ? f_row1 : f_row0, rw, rh, f_row0};
    const CrossJob jr{rev, fwd, rw, rows_r ? r_row1 : r_row0, fw, fh, r_row0};
    dim3 grid(((fw > rw ? fw : rw) + 63) / 64, (rows_max + 4 * CC_ROWS - 1) / (4 * CC_ROWS), 2);
    hipLaunchKernelGGL(cross_check_kernel, grid, dim3(256), 0, s, jf, jr, zero_words);
}

// ---------------------------------------------------------------------------------------------
// expand_grid: the scatter of mod.rs:311-316 plus the Match position of mod.rs:459-462, applied
// once at complete(): full-res cell (x << k, y << k) = level cell (x, y) with the match scaled
// back by round(x2 / scale) = x2 << k.  All other full-res cells are None.
// ---------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void expand_grid_kernel(const uint32_t *__restrict__ cells, const float *__restrict__ scores,
                                                           uint32_t lw, uint32_t lh,
                                                           uint32_t k, uint32_t gw, uint32_t gy0, uint32_t gy1,
                                                           int32_t *__restrict__ out_xy, float *__restrict__ out_corr)
{
    const uint32_t gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t gy = gy0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= gw || gy >= gy1) return;
    int32_t ox = -1, oy = -1;
    float oc = __builtin_nanf("");
    const uint32_t mask = (1u << k) - 1u;
    if ((gx & mask) == 0 && (gy & mask) == 0) {
        const uint32_t lx = gx >> k, ly = gy >> k;
        if (lx < lw && ly < lh) {
            const uint32_t c = cells[(size_t)ly * lw + lx];
            if (c != CELL_NONE) {
                ox = (int32_t)((c & 0xFFFFu) << k);
                oy = (int32_t)((c >> 16) << k);
                if (scores) oc = scores[(size_t)ly * lw + lx];
            }
        }
    }
    const size_t o = (size_t)gy * gw + gx;
    if (PACKED) // one word per cell: y << 16 | x, all ones = None (the coordinates are below 65536: check_level_args)
        reinterpret_cast<uint32_t *>(out_xy)[o] = ox < 0 ? CELL_NONE : ((uint32_t)oy << 16 | (uint32_t)ox);
    else
        reinterpret_cast<int2 *>(out_xy)[o] = make_int2(ox, oy);
    if (out_corr) out_corr[o] = oc;
}

// rows [gy0, min(gy1, gh)) of the full-resolution grid
void launch_expand_grid(const uint32_t *cells, const float *scores, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                        int32_t *out_xy, float *out_corr, hipStream_t s, uint32_t gy0, uint32_t gy1, bool packed)
{
    gy1 = std::min(gy1, gh);
    if (gy1 <= gy0) return;
    dim3 grid((gw + 63) / 64, (gy1 - gy0 + 3) / 4);
    if (packed)
        hipLaunchKernelGGL(expand_grid_kernel<true>, grid, dim3(256), 0, s, cells, scores, lw, lh, k, gw, gy0, gy1, out_xy, out_corr);
    else
        hipLaunchKernelGGL(expand_grid_kernel<false>, grid, dim3(256), 0, s, cells, scores, lw, lh, k, gw, gy0, gy1, out_xy, out_corr);
}

// ---------------------------------------------------------------------------------------------
// triangulate_affine: AffineTriangulation::triangulate + triangulate_point (triangulation.rs:268-330)
// on the device-resident forward grid: count Some cells per 256-cell block (scan order), exclusive
// scan of the block counts, then an ordered write of (x, y, sqrt(dx^2 + dy^2)) per track.
// dx^2 + dy^2 is an exact integer in f64; sqrt is the correctly rounded f64 square root.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tri_count_kernel(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                         uint32_t k, uint32_t gw, uint32_t gh,
                                                         uint32_t *__restrict__ block_counts)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t mx, my;
    const bool f = i < (size_t)gw * gh && full_res_match(cells, lw, lh, k, (uint32_t)(i % gw), (uint32_t)(i / gw), mx, my);
    __shared__ uint32_t wsum[4];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-block exclusive scan over n values (in place); total written to *total
__global__ __launch_bounds__(1024) void tri_scan_kernel(uint32_t *__restrict__ data, uint32_t n,
                                                         uint32_t *__restrict__ total)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? data[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t t = __shfl_up(incl, s, 64);
            if ((int)(threadIdx.x & 63) >= s) incl += t;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) woff += wtot[w];
        const uint32_t carry = carry_s;
        if (i < n) data[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void tri_write_kernel(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh,
                                                         uint32_t k, uint32_t gw, uint32_t gh,
                                                         const uint32_t *__restrict__ block_offsets,
                                                         unsigned long long cap, double *__restrict__ out_points3d,
                                                         uint32_t *__restrict__ out_p2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t mx = 0, my = 0;
    const uint32_t gx = (uint32_t)(i % gw), gy = (uint32_t)(i / gw);
    const bool f = i < (size_t)gw * gh && full_res_match(cells, lw, lh, k, gx, gy, mx, my);
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
            const double dx = (double)gx - (double)mx, dy = (double)gy - (double)my;
            out_points3d[3 * off + 0] = (double)gx;
            out_points3d[3 * off + 1] = (double)gy;
            out_points3d[3 * off + 2] = sqrt(dx * dx + dy * dy);
            if (out_p2) {
                out_p2[2 * off + 0] = mx;
                out_p2[2 * off + 1] = my;
            }
        }
    }
}

void launch_triangulate_affine(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                               uint32_t *block_counts, uint32_t *total, double *out_points3d, uint32_t *out_p2,
                               unsigned long long cap, hipStream_t s)
{
    const uint32_t nblocks = (uint32_t)(((size_t)gw * gh + 255) / 256);
    hipLaunchKernelGGL(tri_count_kernel, dim3(nblocks), dim3(256), 0, s, cells, lw, lh, k, gw, gh, block_counts);
    hipLaunchKernelGGL(tri_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nblocks, total);
    if (cap)
        hipLaunchKernelGGL(tri_write_kernel, dim3(nblocks), dim3(256), 0, s, cells, lw, lh, k, gw, gh, block_counts, cap,
                           out_points3d, out_p2);
}

void launch_scan_u32(uint32_t *data, uint32_t n, uint32_t *total, hipStream_t s)
{
    hipLaunchKernelGGL(tri_scan_kernel, dim3(1), dim3(1024), 0, s, data, n, total);
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
