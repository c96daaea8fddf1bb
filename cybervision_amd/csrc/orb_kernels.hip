// orb_kernels.hip — ORB keypoint extraction and brute-force descriptor matching for gfx950.
//
// Replaces orb::extract_points (zlogic/cybervision src/orb.rs:50-84) and
// KeypointMatching::match_points (src/pointmatching.rs:43-77).  Stage by stage:
//
//   minmax + contrast    adjust_contrast            orb.rs:455-472   (u8, one f32 multiply + round)
//   fast_score           find_fast_keypoints        orb.rs:86-135    FAST-9 + score, closed form
//   nms + compaction     find_fast_keypoints        orb.rs:138-187   3x3 NMS, scan-ordered list
//   harris               harris_response            orb.rs:230-269   f64, incl. the 7-wide Sobel quirk
//   sort (stable, desc)  extract_points             orb.rs:76-81     radix sort on ordered f64 bits
//   blur                 gaussian_blur<11>          orb.rs:271-314   separable f64, serial tap order
//   moments              get_brief_orientation      orb.rs:316-344   integer moments (exact)
//   (host)               atan2 / sin / cos          orb.rs:341,365   libm, exactly as the reference
//   brief                extract_brief_descriptors  orb.rs:346-405   f64 rotation, ballot-packed bits
//
// Integer stages are exact by construction; every f64 stage keeps the reference's serial
// operation order with contraction off, so keypoint coordinates AND descriptors are
// bit-identical to --mode=cpu.  Transcendentals stay on the host because device libm is not
// bit-identical to glibc.
#include "cvhip_internal.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <cmath>
#include <system_error>
#include <thread>
#include <vector>

#include "orb_pattern.inc"

namespace cvhip {

constexpr int FAST_KERNEL_SIZE = 3;
constexpr int FAST_THRESHOLD = 15;
constexpr int HARRIS_KERNEL_SIZE = 3;
constexpr int HARRIS_KERNEL_WIDTH = 7;
constexpr double HARRIS_K = 0.04;
constexpr int ORB_GAUSS_KERNEL_WIDTH = 11;
constexpr int ORB_PATCH_WIDTH = 31;
constexpr int ORB_PATCH_SIZE = 15;
constexpr uint32_t MAX_KEYPOINTS = 10000;

struct Taps7 {
    double k[7];
};
struct Taps11 {
    double k[11];
};

// ---------------------------------------------------------------------------------------------
// adjust_contrast (orb.rs:455-472)
// ---------------------------------------------------------------------------------------------
// img is the library's own 4-byte-aligned copy: dword loads, four pixels per lane and load; one pair of atomics per
// workgroup (a pair per wave from 2048 workgroups - 16 k atomics on two addresses - cost 120 us per image).
__device__ __forceinline__ void minmax_body(const uint8_t *__restrict__ img, size_t n, uint32_t *__restrict__ mm)
{
    __shared__ uint32_t wlo[4], whi[4];
    uint32_t lo = 255, hi = 0;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *__restrict__ words = reinterpret_cast<const uint32_t *>(img);
    for (size_t i = gid; i < n4; i += stride) {
        const uint32_t v = words[i];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t px = (v >> (8 * b)) & 0xFFu;
            lo = min(lo, px);
            hi = max(hi, px);
        }
    }
    for (size_t i = (n4 << 2) + gid; i < n; i += stride) {
        const uint32_t px = img[i];
        lo = min(lo, px);
        hi = max(hi, px);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        lo = min(lo, (uint32_t)__shfl_down(lo, s, 64));
        hi = max(hi, (uint32_t)__shfl_down(hi, s, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        wlo[threadIdx.x >> 6] = lo;
        whi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&mm[0], min(min(wlo[0], wlo[1]), min(wlo[2], wlo[3])));
        atomicMax(&mm[1], max(max(whi[0], whi[1]), max(whi[2], whi[3])));
    }
}

__device__ __forceinline__ void contrast_body(const uint8_t *__restrict__ img, size_t n, const uint32_t *__restrict__ mm,
                                uint8_t *__restrict__ out)
{
    const uint32_t lo = mm[0], hi = mm[1];
    const bool identity = lo >= hi;
    const float coeff = identity ? 1.0f : 255.0f / (float)(hi - lo);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = img[i];
        uint32_t o = v;
        if (!identity) {
            const float r = roundf(coeff * (float)(v - lo));
            o = r >= 255.0f ? 255u : (r > 0.0f ? (uint32_t)r : 0u); // `as u8` saturates
        }
        out[i] = (uint8_t)o;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9 score (orb.rs:86-135, is_keypoint :424-453).
// is_keypoint(t) <=> some circular run of >= 9 ring pixels is brighter than v + t, or darker
// than v - t.  With d_i = c_i - v:  brighter run at t  <=>  max_s min_{j<9} d_{s+j} > t, and
// symmetrically for darker.  The reference's bisection (orb.rs:122-133) returns the largest t in
// [15, 254] for which is_keypoint holds (is_keypoint is monotone in t), i.e. M - 1 with
// M = max(best_brighter, best_darker).  score plane: 0 = not a corner, else that threshold.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ring_best_min9(const int (&d)[16])
{
    int m1[16], m2[16], m4[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m1[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m2[i] = min(m1[i], m1[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 4) & 15]);
    int best = -1000;
#pragma unroll
    for (int i = 0; i < 16; i++) best = max(best, min(m4[i], d[(i + 8) & 15]));
    return best;
}

__device__ __forceinline__ void fast_score_body(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                          uint8_t *__restrict__ score)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t out = 0;
    if (x >= FAST_KERNEL_SIZE && y >= FAST_KERNEL_SIZE && x + FAST_KERNEL_SIZE < w && y + FAST_KERNEL_SIZE < h) {
        const int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        const int oy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
        const int v = img[(size_t)y * w + x];
        int d[16], nd[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            d[i] = (int)img[(size_t)(y + oy[i]) * w + (x + ox[i])] - v;
            nd[i] = -d[i];
        }
        const int m = max(ring_best_min9(d), ring_best_min9(nd));
        if (m - 1 >= FAST_THRESHOLD) out = (uint32_t)min(m - 1, 254);
    }
    score[(size_t)y * w + x] = (uint8_t)out;
}

// 3x3 non-maximum suppression (orb.rs:138-187): a corner is dropped when any of its eight
// neighbours is a corner with score >= its own.
__device__ __forceinline__ bool nms_survives(const uint8_t *__restrict__ score, uint32_t w, uint32_t h, uint32_t x,
                                             uint32_t y)
{
    const uint32_t s = score[(size_t)y * w + x];
    if (s == 0) return false;
    // corners only exist for 3 <= x < w-3, 3 <= y < h-3, so the 3x3 neighbourhood is in bounds
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            if (dx == 0 && dy == 0) continue;
            if (score[(size_t)(y + dy) * w + (x + dx)] >= s) return false;
        }
    return true;
}

// ---------------------------------------------------------------------------------------------
// ordered (scan-order) stream compaction helpers: block counts -> exclusive scan -> write
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void nms_count_body(const uint8_t *__restrict__ score, uint32_t w, uint32_t h,
                                                         uint32_t *__restrict__ block_counts)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)w * h;
    bool f = false;
    if (i < n) f = nms_survives(score, w, h, (uint32_t)(i % w), (uint32_t)(i / w));
    __shared__ uint32_t wsum[4];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-block exclusive scan over n values (in place); total written to *total
__device__ __forceinline__ void exclusive_scan_body(uint32_t *__restrict__ data, uint32_t n,
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
        for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (i < n) data[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__device__ __forceinline__ void nms_write_body(const uint8_t *__restrict__ score, uint32_t w, uint32_t h,
                                                         const uint32_t *__restrict__ block_offsets, uint32_t cap,
                                                         uint32_t *__restrict__ out_xy)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)w * h;
    bool f = false;
    uint32_t x = 0, y = 0;
    if (i < n) {
        x = (uint32_t)(i % w);
        y = (uint32_t)(i / w);
        f = nms_survives(score, w, h, x, y);
    }
    __shared__ uint32_t wsum[4];
    const unsigned long long b = __ballot(f);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
    __syncthreads();
    if (f) {
        uint32_t off = block_offsets[blockIdx.x];
        for (uint32_t k = 0; k < wv; k++) off += wsum[k];
        off += (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (off < cap) {
            out_xy[2 * (size_t)off] = x;
            out_xy[2 * (size_t)off + 1] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Harris response (orb.rs:204-269), one thread per FAST corner, on the ORIGINAL image.
// Keeps the reference's quirk: convolve_kernel::<7, 9> indexes the nine Sobel taps with width 7,
// so they sample (x-3..x+3, y-3), (x-3, y-2), (x-2, y-2).  key = order-preserving u64 of the f64
// response (for a descending stable radix sort); None (too close to the edge) -> key 0, which
// sorts after every real response.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_order_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ void harris_body(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                     const uint32_t *__restrict__ kp_xy,
                                                     const uint32_t *__restrict__ n_ptr, uint32_t cap, Taps7 kg,
                                                     unsigned long long *__restrict__ keys,
                                                     uint32_t *__restrict__ idx, uint32_t tag = 0u, uint32_t none = 0xFFFFFFFFu)
{
    // tag / none: the batched form sorts the corners of several images together - the value carries the image in its top
    // bits (tag), and "None" is the all-ones INDEX under that tag
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const uint32_t n = min(*n_ptr, cap);
    if (i >= cap) return;
    if (i >= n) {
        keys[i] = 0ull;
        idx[i] = none;
        return;
    }
    const double sobel_x[9] = {-1.0, 0.0, 1.0, -2.0, 0.0, 2.0, -1.0, 0.0, 1.0};
    const double sobel_y[9] = {-1.0, -2.0, -1.0, 0.0, 0.0, 0.0, 1.0, 2.0, 1.0};
    const uint32_t x = kp_xy[2 * (size_t)i], y = kp_xy[2 * (size_t)i + 1];
    const uint32_t ks = HARRIS_KERNEL_WIDTH / 2;
    // every sample point (x + k_x - 3, y + k_y - 3) must itself pass convolve_kernel's bounds check
    bool ok = x >= 2 * ks && y >= 2 * ks && x + 2 * ks < w && y + 2 * ks < h;
    unsigned long long key = 0ull;
    if (ok) {
        double g_dx2 = 0.0, g_dy2 = 0.0, g_dx_dy = 0.0;
        for (int k_y = 0; k_y < HARRIS_KERNEL_WIDTH; k_y++) {
            for (int k_x = 0; k_x < HARRIS_KERNEL_WIDTH; k_x++) {
                const uint32_t px = x + k_x - HARRIS_KERNEL_SIZE, py = y + k_y - HARRIS_KERNEL_SIZE;
                double dx = 0.0, dy = 0.0;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const int tx = t % HARRIS_KERNEL_WIDTH, ty = t / HARRIS_KERNEL_WIDTH;
                    const double v = (double)img[(size_t)(py + ty - ks) * w + (px + tx - ks)];
                    dx += sobel_x[t] * v / 255.0;
                    dy += sobel_y[t] * v / 255.0;
                }
                const double gauss_mul = kg.k[k_x] * kg.k[k_y];
                g_dx2 += dx * dx * gauss_mul;
                g_dy2 += dy * dy * gauss_mul;
                g_dx_dy += dx * dy * gauss_mul;
            }
        }
        const double det = g_dx2 * g_dy2 - g_dx_dy * g_dx_dy;
        const double trace = g_dx2 + g_dy2;
        key = f64_order_key(det - HARRIS_K * (trace * trace));
    }
    keys[i] = key;
    idx[i] = ok ? (i | tag) : none;
}

// ---------------------------------------------------------------------------------------------
// gaussian_blur<11> (orb.rs:271-314): horizontal then vertical, f64, taps added in index order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void blur_h_body(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                      Taps11 kg, double *__restrict__ out)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint32_t ks = ORB_GAUSS_KERNEL_WIDTH / 2;
    double sum = 0.0;
    if (x >= ks && x + ks < w && y >= ks && y + ks < h) {
#pragma unroll
        for (int i = 0; i < ORB_GAUSS_KERNEL_WIDTH; i++) sum += kg.k[i] * (double)img[(size_t)y * w + (x + i - ks)];
    }
    out[(size_t)y * w + x] = sum;
}
__device__ __forceinline__ void blur_v_body(const double *__restrict__ in, uint32_t w, uint32_t h, Taps11 kg,
                                                      double *__restrict__ out)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint32_t ks = ORB_GAUSS_KERNEL_WIDTH / 2;
    double sum = 0.0;
    if (x >= ks && x + ks < w && y >= 2 * ks && y + 2 * ks < h) {
#pragma unroll
        for (int i = 0; i < ORB_GAUSS_KERNEL_WIDTH; i++) sum += kg.k[i] * in[(size_t)(y + i - ks) * w + x];
    }
    out[(size_t)y * w + x] = sum;
}
// Some(...) cells of the blurred grid.  QUIRK (orb.rs:293): the second grid is allocated
// width x width, so rows >= width do not exist.
__device__ __forceinline__ bool blur_valid(uint32_t w, uint32_t h, uint32_t x, uint32_t y)
{
    const uint32_t ks = ORB_GAUSS_KERNEL_WIDTH / 2;
    return x >= ks && x + ks < w && y >= 2 * ks && y + 2 * ks < h && y < w;
}

// ---------------------------------------------------------------------------------------------
// patch moments (orb.rs:316-339): one wave per ranked keypoint.  Integer sums are exact, so the
// wave-parallel reduction equals the reference's serial loop.  out = (m00, m10, m01, valid).
constexpr uint32_t ORB_BAD_INDEX = 0x80000000u; // in the describe step's open-count word: a sorted corner index was out of range
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void moments_body(const double *__restrict__ blur, uint32_t w, uint32_t h,
                                                      const uint32_t *__restrict__ kp_xy,
                                                      const uint32_t *__restrict__ sorted_idx, uint32_t count, uint32_t n_kp,
                                                      unsigned long long *__restrict__ out, double *__restrict__ sincos)
{
    const uint32_t r = blockIdx.x;
    if (r >= count) return;
    const uint32_t src = sorted_idx[r];
    const uint32_t lane = threadIdx.x;
    unsigned long long m00 = 0, m10 = 0, m01 = 0;
    // (an index beyond the corner list - a sort that went wrong, see scripts/micro/rocprim_partial_bits_sort.hip - must not
    // become an address: the keypoint is dropped here and brief_body reports it)
    bool ok = src != 0xFFFFFFFFu && src < n_kp;
    uint32_t x = 0, y = 0;
    if (ok) {
        x = kp_xy[2 * (size_t)src];
        y = kp_xy[2 * (size_t)src + 1];
        const uint32_t bh = w; // blurred grid height (quirk)
        ok = x >= ORB_PATCH_SIZE && y >= ORB_PATCH_SIZE && x + ORB_PATCH_SIZE < w && y + ORB_PATCH_SIZE < bh;
    }
    if (ok) {
        bool all_some = true;
        for (uint32_t c = lane; c < ORB_PATCH_WIDTH * ORB_PATCH_WIDTH; c += 64) {
            const uint32_t m_x = c % ORB_PATCH_WIDTH, m_y = c / ORB_PATCH_WIDTH;
            const uint32_t s_x = x + m_x - ORB_PATCH_SIZE, s_y = y + m_y - ORB_PATCH_SIZE;
            if (!blur_valid(w, h, s_x, s_y)) {
                all_some = false;
                continue;
            }
            const double v = blur[(size_t)s_y * w + s_x];
            const double cl = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
            const unsigned long long val = (unsigned long long)cl; // truncation, `as usize`
            m00 += val;
            m10 += (unsigned long long)s_x * val;
            m01 += (unsigned long long)s_y * val;
        }
        ok = __all(all_some);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            m00 += __shfl_down(m00, s, 64);
            m10 += __shfl_down(m10, s, 64);
            m01 += __shfl_down(m01, s, 64);
        }
    }
    if (lane == 0) { // five words per rank: the moments, the validity flag, the keypoint itself (x | y << 32)
        out[5 * (size_t)r + 0] = m00;
        out[5 * (size_t)r + 1] = m10;
        out[5 * (size_t)r + 2] = m01;
        out[5 * (size_t)r + 3] = ok ? 1ull : 0ull;
        out[5 * (size_t)r + 4] = (unsigned long long)x | ((unsigned long long)y << 32);
        // The orientation (orb.rs:337-341, 365-366) with the DEVICE's atan2 / sin / cos: the same IEEE divisions and
        // subtractions as the reference, then library functions that agree with glibc's to a few ulp.  brief_kernel
        // uses them only where that cannot change a rounded sample offset (its `guard`); a keypoint where it could is
        // redone with the host's libm.
        double sn = 0.0, cs = 0.0;
        if (ok) {
            const double m00d = (double)m00;
            const double centroid_x = (double)m10 / m00d, centroid_y = (double)m01 / m00d;
            const double angle = atan2(centroid_y - (double)y, centroid_x - (double)x);
            sn = sin(angle);
            cs = cos(angle);
        }
        sincos[3 * (size_t)r + 0] = sn;
        sincos[3 * (size_t)r + 1] = cs;
        sincos[3 * (size_t)r + 2] = ok ? 1.0 : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// rotated BRIEF (orb.rs:363-402): one wave per ranked keypoint, four pair tests per lane; the
// 64-bit ballot of test j*64+lane yields descriptor words 2j and 2j+1 directly.
// ---------------------------------------------------------------------------------------------
struct Pattern {
    signed char v[1024];
};
__device__ __forceinline__ long long f64_to_i64_sat(double v)
{
    if (v != v) return 0;
    if (v >= 9.2e18) return 0x7FFFFFFFFFFFFFFFll;
    if (v <= -9.2e18) return (long long)0x8000000000000000ull;
    return (long long)v;
}
__device__ __forceinline__ unsigned long long sat_add_signed(unsigned long long a, long long b)
{
    if (b >= 0) {
        const unsigned long long r = a + (unsigned long long)b;
        return r < a ? ~0ull : r;
    }
    const unsigned long long nb = (unsigned long long)(-(b + 1)) + 1ull;
    return a > nb ? a - nb : 0ull;
}

__device__ __forceinline__ void brief_body(const double *__restrict__ blur, uint32_t w, uint32_t h,
                                                    const uint32_t *__restrict__ kp_xy,
                                                    const uint32_t *__restrict__ sorted_idx, uint32_t count, uint32_t n_kp,
                                                    const double *__restrict__ sincos,
                                                    const signed char *__restrict__ pattern,
                                                    uint32_t *__restrict__ desc, uint32_t *__restrict__ flags, double guard,
                                                    uint32_t *__restrict__ open_count)
{
    // guard > 0: sin / cos come from the device's math library (moments_kernel).  They - and the angle under them - may
    // differ from glibc's by a few ulp, which moves a rotated offset o_y cos - o_x sin (|o| <= 15) by less than 1e-12:
    // its rounding is the reference's unless the value lies within `guard` (1e-9) of a half-integer.  A keypoint with
    // such a sample is counted in *open_count and the whole image is redone with host-computed orientations.
    const uint32_t r = blockIdx.x;
    if (r >= count) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t src = sorted_idx[r];
    const double angle_sin = sincos[3 * (size_t)r + 0], angle_cos = sincos[3 * (size_t)r + 1];
    const bool in_list = src == 0xFFFFFFFFu || src < n_kp;
    if (!in_list && lane == 0 && open_count) atomicOr(open_count, ORB_BAD_INDEX); // the host turns this into CVHIP_ERR_DEVICE
    bool ok = src != 0xFFFFFFFFu && in_list && sincos[3 * (size_t)r + 2] != 0.0;
    uint32_t cx = 0, cy = 0;
    if (ok) {
        cx = kp_xy[2 * (size_t)src];
        cy = kp_xy[2 * (size_t)src + 1];
    }
    const uint32_t bh = w; // blurred grid height (quirk)
    bool fail = false, open = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = j * 64 + (int)lane;
        const double o1x = (double)pattern[4 * i + 0], o1y = (double)pattern[4 * i + 1];
        const double o2x = (double)pattern[4 * i + 2], o2y = (double)pattern[4 * i + 3];
        // x/y roles as in the reference (orb.rs:371-378)
        const double v1x = o1y * angle_cos - o1x * angle_sin, v1y = o1y * angle_sin + o1x * angle_cos;
        const double v2x = o2y * angle_cos - o2x * angle_sin, v2y = o2y * angle_sin + o2x * angle_cos;
        if (guard > 0.0) {
            const auto near_half = [guard](double v) { return fabs((v - floor(v)) - 0.5) < guard; };
            open = open || near_half(v1x) || near_half(v1y) || near_half(v2x) || near_half(v2y);
        }
        const long long off1x = f64_to_i64_sat(round(v1x));
        const long long off1y = f64_to_i64_sat(round(v1y));
        const long long off2x = f64_to_i64_sat(round(v2x));
        const long long off2y = f64_to_i64_sat(round(v2y));
        const unsigned long long p1x = sat_add_signed(cx, off1x), p1y = sat_add_signed(cy, off1y);
        const unsigned long long p2x = sat_add_signed(cx, off2x), p2y = sat_add_signed(cy, off2y);
        bool bad = p1x == 0 || p2x == 0 || p1x + 1 >= w || p2x + 1 >= w || p1y + 1 >= bh || p2y + 1 >= bh;
        bool tau = false;
        if (ok && !bad) {
            if (!blur_valid(w, h, (uint32_t)p1x, (uint32_t)p1y) || !blur_valid(w, h, (uint32_t)p2x, (uint32_t)p2y)) {
                bad = true;
            } else {
                const double v1 = blur[(size_t)p1y * w + p1x], v2 = blur[(size_t)p2y * w + p2x];
                tau = v1 < v2;
            }
        }
        fail = fail || bad;
        const unsigned long long bits = __ballot(tau);
        if (lane == 0) {
            desc[8 * (size_t)r + 2 * j] = (uint32_t)(bits & 0xFFFFFFFFull);
            desc[8 * (size_t)r + 2 * j + 1] = (uint32_t)(bits >> 32);
        }
    }
    const bool any_fail = __any(fail), any_open = __any(ok && open);
    if (lane == 0) {
        flags[r] = (ok && !any_fail) ? 1u : 0u;
        if (any_open && open_count) atomicAdd(open_count, 1u);
    }
}

// ordered compaction of <= 10240 ranked keypoints by flag (single block)
__device__ __forceinline__ void final_compact_body(const uint32_t *__restrict__ flags,
                                                              const uint32_t *__restrict__ kp_xy,
                                                              const uint32_t *__restrict__ sorted_idx,
                                                              const uint32_t *__restrict__ desc, uint32_t count,
                                                              uint32_t cap, uint32_t *__restrict__ out_xy,
                                                              uint32_t *__restrict__ out_desc,
                                                              uint32_t *__restrict__ out_n)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < count; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const bool f = i < count && flags[i] != 0;
        const unsigned long long b = __ballot(f);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wtot[wv] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t k = 0; k < wv; k++) off += wtot[k];
        off += (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (f && off < cap) {
            const uint32_t src = sorted_idx[i];
            out_xy[2 * (size_t)off] = kp_xy[2 * (size_t)src];
            out_xy[2 * (size_t)off + 1] = kp_xy[2 * (size_t)src + 1];
#pragma unroll
            for (int k = 0; k < 8; k++) out_desc[8 * (size_t)off + k] = desc[8 * (size_t)i + k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int k = 0; k < 16; k++) t += wtot[k];
            carry_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_n = min(carry_s, cap);
}


// ---- launchable forms: one image per launch (the redo path, single images), and one launch for ALL images of a batch
// (blockIdx.z = the image: the bodies only ever look at blockIdx.x / .y) ----
__global__ __launch_bounds__(256) void minmax_kernel(const uint8_t *__restrict__ img, size_t n, uint32_t *__restrict__ mm)
{
    minmax_body(img, n, mm);
}
__global__ void contrast_kernel(const uint8_t *__restrict__ img, size_t n, const uint32_t *__restrict__ mm,
                                uint8_t *__restrict__ out)
{
    contrast_body(img, n, mm, out);
}
__global__ __launch_bounds__(256) void fast_score_kernel(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                          uint8_t *__restrict__ score)
{
    fast_score_body(img, w, h, score);
}
__global__ __launch_bounds__(256) void nms_count_kernel(const uint8_t *__restrict__ score, uint32_t w, uint32_t h,
                                                         uint32_t *__restrict__ block_counts)
{
    nms_count_body(score, w, h, block_counts);
}
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(uint32_t *__restrict__ data, uint32_t n,
                                                               uint32_t *__restrict__ total)
{
    exclusive_scan_body(data, n, total);
}
__global__ __launch_bounds__(256) void nms_write_kernel(const uint8_t *__restrict__ score, uint32_t w, uint32_t h,
                                                         const uint32_t *__restrict__ block_offsets, uint32_t cap,
                                                         uint32_t *__restrict__ out_xy)
{
    nms_write_body(score, w, h, block_offsets, cap, out_xy);
}
__global__ __launch_bounds__(64) void harris_kernel(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                     const uint32_t *__restrict__ kp_xy,
                                                     const uint32_t *__restrict__ n_ptr, uint32_t cap, Taps7 kg,
                                                     unsigned long long *__restrict__ keys,
                                                     uint32_t *__restrict__ idx)
{
    harris_body(img, w, h, kp_xy, n_ptr, cap, kg, keys, idx);
}
__global__ __launch_bounds__(256) void blur_h_kernel(const uint8_t *__restrict__ img, uint32_t w, uint32_t h,
                                                      Taps11 kg, double *__restrict__ out)
{
    blur_h_body(img, w, h, kg, out);
}
__global__ __launch_bounds__(256) void blur_v_kernel(const double *__restrict__ in, uint32_t w, uint32_t h, Taps11 kg,
                                                      double *__restrict__ out)
{
    blur_v_body(in, w, h, kg, out);
}
__global__ __launch_bounds__(64) void moments_kernel(const double *__restrict__ blur, uint32_t w, uint32_t h,
                                                      const uint32_t *__restrict__ kp_xy,
                                                      const uint32_t *__restrict__ sorted_idx, uint32_t count, uint32_t n_kp,
                                                      unsigned long long *__restrict__ out, double *__restrict__ sincos)
{
    moments_body(blur, w, h, kp_xy, sorted_idx, count, n_kp, out, sincos);
}
__global__ __launch_bounds__(64) void brief_kernel(const double *__restrict__ blur, uint32_t w, uint32_t h,
                                                    const uint32_t *__restrict__ kp_xy,
                                                    const uint32_t *__restrict__ sorted_idx, uint32_t count, uint32_t n_kp,
                                                    const double *__restrict__ sincos,
                                                    const signed char *__restrict__ pattern,
                                                    uint32_t *__restrict__ desc, uint32_t *__restrict__ flags, double guard,
                                                    uint32_t *__restrict__ open_count)
{
    brief_body(blur, w, h, kp_xy, sorted_idx, count, n_kp, sincos, pattern, desc, flags, guard, open_count);
}
__global__ __launch_bounds__(1024) void final_compact_kernel(const uint32_t *__restrict__ flags,
                                                              const uint32_t *__restrict__ kp_xy,
                                                              const uint32_t *__restrict__ sorted_idx,
                                                              const uint32_t *__restrict__ desc, uint32_t count,
                                                              uint32_t cap, uint32_t *__restrict__ out_xy,
                                                              uint32_t *__restrict__ out_desc,
                                                              uint32_t *__restrict__ out_n)
{
    final_compact_body(flags, kp_xy, sorted_idx, desc, count, cap, out_xy, out_desc, out_n);
}

// One image of a batch as the batched launches see it (a table of these in device memory, blockIdx.z picks the row).
struct OrbJobDev {
    const uint8_t *img;       // the library's padded copy
    uint8_t *adj, *score;
    uint32_t *mm, *counts, *total;
    uint32_t w, h, nblocks, rblocks, g2x, g2y;
    unsigned long long n;
    // stage B
    uint32_t n_fast, count, out_cap, key_off;
    uint32_t *kp, *idx_sorted, *desc, *flags, *pack, *out_xy, *out_desc;
    unsigned long long *keys, *mom;
    uint32_t *idx;
    double *blur_h, *blur, *sc;
};
constexpr uint32_t ORB_TAG_SHIFT = 28, ORB_TAG_MASK = 0x0FFFFFFFu; // (image << 28 | corner index) as the sorts' value

__global__ __launch_bounds__(256) void orb_stage_a_init_jobs(const OrbJobDev *jobs, uint32_t n_jobs)
{
    for (uint32_t i = threadIdx.x; i < n_jobs; i += 256) {
        jobs[i].mm[0] = 255u; // {min, max} = {255, 0}
        jobs[i].mm[1] = 0u;
    }
}
__global__ __launch_bounds__(256) void minmax_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    minmax_body(j.img, (size_t)j.n, j.mm); // (a grid-stride loop: every workgroup of the launch takes part)
}
__global__ __launch_bounds__(256) void contrast_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    contrast_body(j.img, (size_t)j.n, j.mm, j.adj);
}
__global__ __launch_bounds__(256) void fast_score_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.g2x || blockIdx.y >= j.g2y) return;
    fast_score_body(j.adj, j.w, j.h, j.score);
}
__global__ __launch_bounds__(256) void nms_count_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.nblocks) return;
    nms_count_body(j.score, j.w, j.h, j.counts);
}
__global__ __launch_bounds__(1024) void exclusive_scan_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    exclusive_scan_body(j.counts, j.nblocks, j.total);
}
__global__ __launch_bounds__(256) void nms_write_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.nblocks || j.n_fast == 0) return;
    if (blockIdx.x == 0 && threadIdx.x < 2) j.pack[threadIdx.x] = 0u; // {n_out, open count} of the describe step
    nms_write_body(j.score, j.w, j.h, j.counts, j.n_fast, j.kp);
}
__global__ __launch_bounds__(64) void harris_jobs(const OrbJobDev *jobs, Taps7 kg)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= (j.n_fast + 63u) / 64u) return;
    const uint32_t tag = blockIdx.z << ORB_TAG_SHIFT;
    harris_body(j.img, j.w, j.h, j.kp, j.total, j.n_fast, kg, j.keys, j.idx, tag, tag | ORB_TAG_MASK);
}
// Grouping the globally sorted corners by image, keeping their order: rank r of the first sort gets the 32-bit key
// (image << 28 | r), a full-width sort of those keys brings every image's ranks together in ascending r, and the
// gather below turns them back into plain per-image index lists (None = ~0, as the single-image sort leaves it).
__global__ __launch_bounds__(256) void orb_rank_key_kernel(const uint32_t *__restrict__ tagged, uint32_t n, uint32_t *__restrict__ key)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    key[i] = (tagged[i] & ~ORB_TAG_MASK) | i;
}
__global__ __launch_bounds__(256) void orb_untag_kernel(const uint32_t *__restrict__ tagged, const uint32_t *__restrict__ key_sorted,
                                                        uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = tagged[key_sorted[i] & ORB_TAG_MASK] & ORB_TAG_MASK;
    out[i] = v == ORB_TAG_MASK ? 0xFFFFFFFFu : v;
}
__global__ __launch_bounds__(256) void blur_h_jobs(const OrbJobDev *jobs, Taps11 kg)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.g2x || blockIdx.y >= j.g2y || j.n_fast == 0) return;
    blur_h_body(j.img, j.w, j.h, kg, j.blur_h);
}
__global__ __launch_bounds__(256) void blur_v_jobs(const OrbJobDev *jobs, Taps11 kg)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.g2x || blockIdx.y >= j.g2y || j.n_fast == 0) return;
    blur_v_body(j.blur_h, j.w, j.h, kg, j.blur);
}
__global__ __launch_bounds__(64) void moments_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.count) return;
    moments_body(j.blur, j.w, j.h, j.kp, j.idx_sorted, j.count, j.n_fast, j.mom, j.sc);
}
__global__ __launch_bounds__(64) void brief_jobs(const OrbJobDev *jobs, const signed char *__restrict__ pattern, double guard)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (blockIdx.x >= j.count) return;
    brief_body(j.blur, j.w, j.h, j.kp, j.idx_sorted, j.count, j.n_fast, j.sc, pattern, j.desc, j.flags, guard, j.pack + 1);
}
__global__ __launch_bounds__(1024) void final_compact_jobs(const OrbJobDev *jobs)
{
    const OrbJobDev &j = jobs[blockIdx.z];
    if (j.n_fast == 0) return;
    final_compact_body(j.flags, j.kp, j.idx_sorted, j.desc, j.count, j.out_cap, j.out_xy, j.out_desc, j.pack);
}

// ---------------------------------------------------------------------------------------------
// matcher (pointmatching.rs:43-77): one thread per query keypoint, the train descriptors stream
// through LDS in tiles read at a wave-uniform address (broadcast).  First minimum wins
// (Iterator::min_by keeps the first of equal elements).
// ---------------------------------------------------------------------------------------------
constexpr int MATCH_TILE = 512;
// One thread per query descriptor, the candidates streamed through LDS (every lane reads the same candidate:
// broadcast).  blockIdx.y splits the CANDIDATE list, so that 30 000 queries are 118 x S workgroups instead of 118 (less
// than half a wave per SIMD: 2.5 ms); the splits meet in one 64-bit atomicMin per query on (distance << 32 | j) -
// the smallest distance, and among equal distances the smallest j: exactly what the serial scan's strict `<` keeps.
__global__ __launch_bounds__(256) void match_kernel(const uint32_t *__restrict__ desc1, uint32_t n1,
                                                     const uint32_t *__restrict__ desc2, uint32_t n2, uint32_t chunk,
                                                     uint32_t threshold, unsigned long long *__restrict__ best)
{
    __shared__ uint4 tile[MATCH_TILE * 2];
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (q < n1) {
        a0 = reinterpret_cast<const uint4 *>(desc1)[2 * (size_t)q];
        a1 = reinterpret_cast<const uint4 *>(desc1)[2 * (size_t)q + 1];
    }
    uint32_t bd = 0xFFFFFFFFu, bj = 0xFFFFFFFFu;
    const uint32_t first = blockIdx.y * chunk, last = min(n2, first + chunk);
    for (uint32_t base = first; base < last; base += MATCH_TILE) {
        const uint32_t n = min((uint32_t)MATCH_TILE, last - base);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 2 * n; i += 256)
            tile[i] = reinterpret_cast<const uint4 *>(desc2)[2 * (size_t)base + i];
        __syncthreads();
        for (uint32_t j = 0; j < n; j++) {
            const uint4 b0 = tile[2 * j], b1 = tile[2 * j + 1];
            const uint32_t d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                               __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            if (d <= threshold && d < bd) {
                bd = d;
                bj = base + j;
            }
        }
    }
    if (q < n1 && bj != 0xFFFFFFFFu) atomicMin(&best[q], ((unsigned long long)bd << 32) | bj);
}
// ---- the same minimum on the matrix pipe ------------------------------------------------------------------------
// Hamming distance of 256-bit descriptors is an inner product: d(a, b) = |a| + |b| - 2 a.b with the bits as 0 / 1.
// All pairs of 30 000 x 30 000 descriptors are then a [n1 x 256] x [256 x n2] integer matrix product, which
// v_mfma_i32_32x32x32_i8 does exactly (8 instructions per 32 x 32 pairs against ~19 vector instructions PER PAIR of the
// kernel above, which is bound by vector issue: 0.63 ms for config 5's 30 000 x 29 000).  The descriptors are expanded
// once per call to one byte per bit (queries 0 / 1, candidates 0 / -1, so the product comes out as -a.b), with their
// popcounts next to them; a wave keeps the 32 x 256 bytes of its 32 queries in registers, the candidate tiles stream
// through LDS (shared by the workgroup's four waves), and the epilogue per product is one shift-add and one minimum
// on the key  (|b| + 256 + 2 (-a.b)) << 16 | j  - |a| is added once at the end, the minimum over j commutes with it.
// Both operands use the same lane -> byte map (lane l = row / column l & 31, bytes 16 (l >> 5) .. +15 of the 32-byte
// K slab), so the instruction's internal k order is irrelevant: a sum over all 256 positions either way.
// First minimum wins as above: the key's low half is j, and the splits of the candidate list meet in the same
// 64-bit atomicMin.
typedef int match_i32x4 __attribute__((ext_vector_type(4)));
typedef int match_i32x16 __attribute__((ext_vector_type(16)));
constexpr uint32_t MM_ROWS = 32;       // queries per wave, candidates per tile
constexpr uint32_t MM_PAD_COUNT = 0x4000; // popcount given to the padding rows: no threshold admits them

// bits -> bytes (ones = 1 for queries, 0xFF for candidates), popcounts; rows n .. n_pad - 1 are padding
__global__ __launch_bounds__(256) void match_expand_kernel(const uint32_t *__restrict__ desc, uint32_t n, uint32_t n_pad,
                                                            uint32_t one, uint32_t *__restrict__ bytes, uint32_t *__restrict__ pop)
{
    // one thread per (row, 32-bit word): 32 bytes out
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, row = t >> 3, wd = t & 7u;
    if (row >= n_pad) return;
    const uint32_t v = row < n ? desc[(size_t)row * 8 + wd] : 0u;
    uint4 *out = reinterpret_cast<uint4 *>(bytes + (size_t)row * 64 + wd * 8);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t nib = (v >> (16 * q + 4 * k)) & 0xFu;
            o[k] = ((nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21)) * one;
        }
        out[q] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (wd == 0) {
        uint32_t c = MM_PAD_COUNT;
        if (row < n) {
            c = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) c += __popc(desc[(size_t)row * 8 + k]);
        }
        pop[row] = c;
    }
}

__global__ __launch_bounds__(256) void match_mfma_kernel(const uint32_t *__restrict__ qbytes, const uint32_t *__restrict__ qpop,
                                                          uint32_t n1, const uint32_t *__restrict__ cbytes,
                                                          const uint32_t *__restrict__ cpop, uint32_t n2_pad, uint32_t chunk,
                                                          uint32_t threshold, unsigned long long *__restrict__ best)
{
    constexpr uint32_t PITCH = 17;          // uint4 per candidate row: 16 + 1, so that the 32 rows a ds_read_b128 touches spread over the banks
    __shared__ uint4 tile[MM_ROWS * PITCH]; // 32 candidates x 256 bytes
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, r = lane & 31u, h = lane >> 5;
    const uint32_t q0 = (blockIdx.x * 4 + wave) * MM_ROWS; // this wave's queries (rows past n1 are padding: they exist)
    // A: query q0 + r, K slab s = bytes 32 s + 16 h .. + 15
    match_i32x4 a[8];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(qbytes + (size_t)(q0 + r) * 64);
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            const uint4 v = src[2 * s8 + h];
            a[s8] = match_i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
    }
    uint32_t lowest[16];
#pragma unroll
    for (int i = 0; i < 16; i++) lowest[i] = 0xFFFFFFFFu;
    const uint32_t first = blockIdx.y * chunk, last = min(n2_pad, first + chunk); // multiples of MM_ROWS
    for (uint32_t base = first; base < last; base += MM_ROWS) {
        __syncthreads();
        // the tile: 32 x 256 bytes = 512 uint4, two per thread
        tile[(threadIdx.x >> 4) * PITCH + (threadIdx.x & 15u)] = reinterpret_cast<const uint4 *>(cbytes + (size_t)base * 64)[threadIdx.x];
        tile[((threadIdx.x >> 4) + 16u) * PITCH + (threadIdx.x & 15u)] = reinterpret_cast<const uint4 *>(cbytes + (size_t)base * 64)[threadIdx.x + 256];
        // this lane's candidate (column r of the tile): its popcount and index in the key's format.  |b| - 2 a.b alone can
        // be negative (down to -|b|): a bias of 256 keeps the key's high half an unsigned number until |a| is added
        const uint32_t kb = ((cpop[base + r] + 256u) << 16) | (base - first + r);
        __syncthreads();
        match_i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            const uint4 v = tile[r * PITCH + 2 * s8 + h];
            const match_i32x4 b = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s8], b, acc, 0, 0, 0);
        }
        // acc[i] = -(common ones) of query row (i & 3) + 8 (i >> 2) + 4 h and candidate column r
#pragma unroll
        for (int i = 0; i < 16; i++) lowest[i] = min(lowest[i], ((uint32_t)acc[i] << 17) + kb);
    }
    // per query: + its own popcount, the threshold, the minimum over the 32 candidate columns (lanes of one half)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t q = q0 + (uint32_t)((i & 3) + 8 * (i >> 2)) + 4u * h;
        uint32_t key = lowest[i];
#pragma unroll
        for (int sft = 16; sft > 0; sft >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, sft, 64)); // (stays inside a half of 32)
        if (r == 0 && q < n1 && key != 0xFFFFFFFFu) {
            const uint32_t d = (key >> 16) + qpop[q] - 256u; // |a| + |b| - 2 a.b
            if (d <= threshold) atomicMin(&best[q], ((unsigned long long)d << 32) | (unsigned long long)(first + (key & 0xFFFFu)));
        }
    }
}

// best[] -> the sort's keys (distance; 0xFFFFFFFF = no match, sorts last), the matched index and the query index
__global__ __launch_bounds__(256) void match_unpack_kernel(const unsigned long long *__restrict__ best, uint32_t n1,
                                                            uint32_t *__restrict__ best_j, uint32_t *__restrict__ best_d,
                                                            uint32_t *__restrict__ q_index)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n1) return;
    const unsigned long long k = best[q];
    best_j[q] = (uint32_t)k;
    best_d[q] = (uint32_t)(k >> 32);
    q_index[q] = q;
}

__global__ void match_gather_kernel(const uint32_t *__restrict__ sorted_q, const uint32_t *__restrict__ sorted_d,
                                    const uint32_t *__restrict__ best_j, const uint32_t *__restrict__ xy1,
                                    const uint32_t *__restrict__ xy2, uint32_t n1, uint32_t *__restrict__ out_matches,
                                    uint32_t *__restrict__ out_dist, uint32_t *__restrict__ out_n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const uint32_t d = sorted_d[i];
    if (d == 0xFFFFFFFFu) {
        // first unmatched entry marks the count (distances are sorted ascending)
        if (i == 0 || sorted_d[i - 1] != 0xFFFFFFFFu) *out_n = i;
        return;
    }
    if (i == n1 - 1) *out_n = n1;
    const uint32_t q = sorted_q[i], j = best_j[q];
    out_matches[4 * (size_t)i + 0] = xy1[2 * (size_t)q];
    out_matches[4 * (size_t)i + 1] = xy1[2 * (size_t)q + 1];
    out_matches[4 * (size_t)i + 2] = xy2[2 * (size_t)j];
    out_matches[4 * (size_t)i + 3] = xy2[2 * (size_t)j + 1];
    if (out_dist) out_dist[i] = d;
}

// gaussian_kernel (orb.rs:190-202), host side with libm exp like the reference
static void gaussian_kernel_host(int width, double *kernel)
{
    const double sigma = (double)(width - 1) / 6.0;
    const double sigma_2 = sigma * sigma;
    const double divider = std::sqrt(2.0 * M_PI) * sigma;
    const double center = (double)(width / 2);
    for (int i = 0; i < width; i++) {
        const double d = (double)i - center;
        kernel[i] = std::exp(-(d * d) / (2.0 * sigma_2)) / divider;
    }
}

static bool dev_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

} // namespace cvhip

using namespace cvhip;

// 2x2 box-filter pyramid step (documented substitute for the upstream Lanczos3 resize)
__global__ __launch_bounds__(256) void downsample_box_kernel(const uint8_t *__restrict__ src, uint32_t w, uint32_t dw,
                                                             uint32_t dh, uint8_t *__restrict__ dst)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const uint8_t *r0 = src + (size_t)(2 * y) * w + 2 * x, *r1 = r0 + w;
    dst[(size_t)y * dw + x] = (uint8_t)(((uint32_t)r0[0] + r0[1] + r1[0] + r1[1] + 2u) >> 2);
}

extern "C" int cvhip_downsample_box(cvhip_device *dev, const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst)
{
    if (!dev || !src || !dst) return fail(CVHIP_ERR_INVALID, "null argument");
    const uint32_t dw = w / 2, dh = h / 2;
    if (dw == 0 || dh == 0) return fail(CVHIP_ERR_INVALID, "image too small to halve");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    const bool s_dev = dev_ptr(src), d_dev = dev_ptr(dst);
    const uint8_t *d_src = src;
    uint8_t *d_dst = dst, *tmp = nullptr;
    if (!s_dev) {
        CVHIP_TRY_HIP(mem.alloc(&tmp, (size_t)w * h));
        CVHIP_TRY_HIP(hipMemcpyAsync(tmp, src, (size_t)w * h, hipMemcpyHostToDevice, s));
        d_src = tmp;
    }
    if (!d_dev) CVHIP_TRY_HIP(mem.alloc(&d_dst, (size_t)dw * dh));
    hipLaunchKernelGGL(downsample_box_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, d_src, w, dw, dh,
                       d_dst);
    CVHIP_TRY_HIP(hipGetLastError());
    if (!d_dev) CVHIP_TRY_HIP(hipMemcpyAsync(dst, d_dst, (size_t)dw * dh, hipMemcpyDeviceToHost, s));
    if (!s_dev || !d_dev) CVHIP_TRY_HIP(hipStreamSynchronize(s));
    return CVHIP_OK;
}

// orb::extract_points (orb.rs:50-84) for a BATCH of images in two stages, every stage enqueued for all images before
// the host waits once: (A) contrast stretch, FAST, non-maximum suppression -> corner counts to the host (they size
// everything that follows); (B) corner lists, Harris, ranking, blur, patch moments, orientation, descriptors, ordered
// compaction -> results.  The reference computes atan2 / sin / cos with libm (orb.rs:337-341, 365-366), and a rotated
// sample offset is round(o_y cos - o_x sin): the device's own f64 atan2 / sin / cos decide it identically unless the
// value sits within 1e-9 of a half-integer (they agree with glibc's to a few ulp, |o| <= 15).  A keypoint where it does
// is counted, and an image with such a keypoint takes the host path - moments out, libm on the host, orientations in,
// descriptors again - so the result is always the libm result (test hook: cvhip_orb_set_orientation_guard widens the
// band, or switches the device orientation off).  The reference extracts level after level and image after image
// (reconstruction.rs:418-458); the levels of an image - or all images of a set - are independent, so one batch pays
// the two host round trips once instead of once per extraction.
namespace {
struct OrbJob {
    const uint8_t *img = nullptr;
    uint32_t w = 0, h = 0, cap = 0;
    uint32_t *out_xy = nullptr, *out_desc = nullptr, *out_n = nullptr;
    size_t n = 0;
    uint32_t nblocks = 0, n_fast = 0, count = 0, out_cap = 0;
    uint8_t *d_img = nullptr, *d_adj = nullptr, *d_score = nullptr;
    uint32_t *d_mm = nullptr, *d_counts = nullptr, *d_total = nullptr, *d_kp = nullptr, *d_idx_sorted = nullptr;
    double *d_blur = nullptr, *d_sc = nullptr;
    unsigned long long *d_mom = nullptr;
    uint32_t *d_pack = nullptr, *d_desc = nullptr, *d_flags = nullptr;
    bool xy_dev = false, desc_dev = false;
    size_t mom_off = 0, sc_off = 0, pack_off = 0, mom_bytes = 0, sc_bytes = 0, pack_bytes = 0; // in the pinned staging
};
} // namespace

extern "C" int cvhip_orb_extract_batch(cvhip_device *dev, uint32_t n_images, const uint8_t *const *imgs, const uint32_t *ws,
                                       const uint32_t *hs, uint32_t cap, uint32_t *const *out_xy, uint32_t *const *out_desc,
                                       uint32_t *out_n, cvhip_progress_fn progress, void *user)
{
    if (!dev || !imgs || !ws || !hs || !out_xy || !out_desc || !out_n) return fail(CVHIP_ERR_INVALID, "null argument");
    if (n_images == 0) return CVHIP_OK;
    if (n_images > 256) return fail(CVHIP_ERR_INVALID, "more than 256 images in one batch");
    for (uint32_t i = 0; i < n_images; i++) {
        if (!imgs[i] || !out_xy[i] || !out_desc[i]) return fail(CVHIP_ERR_INVALID, "null argument");
        if (ws[i] < 2 * FAST_KERNEL_SIZE + 1 || hs[i] < 2 * FAST_KERNEL_SIZE + 1)
            return fail(CVHIP_ERR_INVALID, "image smaller than the FAST ring");
        if (ws[i] > 65535 || hs[i] > 65535) return fail(CVHIP_ERR_UNSUPPORTED, "image dimension above 65535");
    }
    // ProgressListener::report_status at the reference's stage boundaries (orb.rs:60-66, 93-100, 112-118, 138-146, 358-363)
    const auto report = [&](float pos) {
        if (progress) progress(user, pos);
    };
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    // The images of a batch are independent and every one of them is a chain of ~25 small launches (2 - 25 us each: a
    // 2048^2 view's four pyramid levels took 1.1 ms, almost all of it launch latency): the chains go round-robin onto the
    // handle's stream and its two side streams, forked and joined with events around each stage.
    constexpr uint32_t LANES = 3;
    hipStream_t lane_stream[LANES] = {s, nullptr, nullptr};
    // Three chains only run side by side if the driver gave the three streams different hardware queues - it does not
    // always (config 5's twelve extractions: 2.3 ms, or 3.7 ms in about every other process; 3.4 ms on one stream).  So the
    // handle times three, two and one chain(s) at a time on the batches after the first of a shape and keeps the fastest.
    Device::OrbLanes &ol = dev->d.orb_lanes;
    size_t shape = n_images;
    for (uint32_t i = 0; i < n_images; i++) shape = shape * 1000003u + (size_t)ws[i] * 65537u + hs[i];
    // Up to 16 images go out as ONE launch per kernel (a table of the images in device memory, blockIdx.z the image):
    // config 5's twelve image levels were ~300 launches of 2 - 25 us, i.e. launch latency; as a batch they are ~30.
    // (16 images of fewer than 2^28 pixels together: image index and rank share one 32-bit word in stage B's sorts.)  Larger
    // batches go image by image, their chains round-robin over the handle's streams:
    size_t batch_px = 0;
    for (uint32_t i = 0; i < n_images; i++) batch_px += (size_t)ws[i] * hs[i];
    const bool batched = n_images <= 16 && batch_px < ((size_t)1 << ORB_TAG_SHIFT);
    uint32_t lanes = n_images > 1 && !batched ? LANES : 1;
    int measuring = 0; // 1 + the number of lanes this call is the sample for
    if (n_images > 1 && !batched) {
        if (ol.shape != shape) { // (the first batch of a shape also pays for streams, events and staging memory: not a sample)
            ol = Device::OrbLanes{};
            ol.shape = shape;
        } else if (ol.decided) {
            lanes = ol.use_lanes;
        } else {
            lanes = LANES - ol.samples; // three, then two, then one
            measuring = 1 + (int)lanes;
        }
    }
    const auto t_begin = std::chrono::steady_clock::now();
    for (uint32_t l = 1; l < lanes; l++) CVHIP_TRY_HIP(aux_stream(dev->d, (int)l - 1, &lane_stream[l]));
    for (uint32_t l = 0; l < lanes; l++)
        if (!dev->d.orb_ev[l]) CVHIP_TRY_HIP(hipEventCreateWithFlags(&dev->d.orb_ev[l], hipEventDisableTiming));
    const auto fork = [&]() -> int { // the side streams continue from where the handle's stream is
        if (lanes == 1) return CVHIP_OK;
        CVHIP_TRY_HIP(hipEventRecord(dev->d.orb_ev[0], s));
        for (uint32_t l = 1; l < lanes; l++) CVHIP_TRY_HIP(hipStreamWaitEvent(lane_stream[l], dev->d.orb_ev[0], 0));
        return CVHIP_OK;
    };
    const auto join = [&]() -> int { // ... and the handle's stream from where they all are
        for (uint32_t l = 1; l < lanes; l++) {
            CVHIP_TRY_HIP(hipEventRecord(dev->d.orb_ev[l], lane_stream[l]));
            CVHIP_TRY_HIP(hipStreamWaitEvent(s, dev->d.orb_ev[l], 0));
        }
        return CVHIP_OK;
    };
    try {
        std::vector<OrbJob> jobs(n_images);

        // ---- stage A: contrast stretch, FAST score, NMS count -> corner totals
        char *pinned = static_cast<char *>(pinned_scratch(dev->d, 4096 + (size_t)n_images * sizeof(OrbJobDev)));
        if (!pinned) return fail(CVHIP_ERR_NOMEM, "cvhip_orb_extract: out of page-locked host memory");
        uint32_t *h_counts = reinterpret_cast<uint32_t *>(pinned);
        OrbJobDev *h_tbl = reinterpret_cast<OrbJobDev *>(pinned + 4096), *d_tbl = nullptr;
        uint32_t *d_totals = nullptr;
        if (batched) {
            CVHIP_TRY_HIP(mem.alloc(&d_tbl, n_images));
            CVHIP_TRY_HIP(mem.alloc(&d_totals, n_images));
            uint32_t gx_contrast = 1, g2x = 1, g2y = 1, gx_nms = 1;
            for (uint32_t i = 0; i < n_images; i++) {
                OrbJob &j = jobs[i];
                j.img = imgs[i];
                j.w = ws[i];
                j.h = hs[i];
                j.cap = cap;
                j.out_xy = out_xy[i];
                j.out_desc = out_desc[i];
                j.out_n = &out_n[i];
                j.n = (size_t)j.w * j.h;
                j.nblocks = (uint32_t)((j.n + 255) / 256);
                CVHIP_TRY_HIP(mem.alloc(&j.d_img, j.n + IMG_PAD));
                CVHIP_TRY_HIP(mem.alloc(&j.d_adj, j.n + IMG_PAD));
                CVHIP_TRY_HIP(mem.alloc(&j.d_score, j.n));
                CVHIP_TRY_HIP(mem.alloc(&j.d_mm, 2));
                CVHIP_TRY_HIP(mem.alloc(&j.d_counts, j.nblocks));
                j.d_total = d_totals + i;
                CVHIP_TRY_HIP(hipMemcpyAsync(j.d_img, j.img, j.n, dev_ptr(j.img) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
                OrbJobDev &t = h_tbl[i];
                std::memset(&t, 0, sizeof(t));
                t.img = j.d_img;
                t.adj = j.d_adj;
                t.score = j.d_score;
                t.mm = j.d_mm;
                t.counts = j.d_counts;
                t.total = j.d_total;
                t.w = j.w;
                t.h = j.h;
                t.nblocks = j.nblocks;
                t.rblocks = (uint32_t)std::min<size_t>(2048, (j.n + 255) / 256);
                t.g2x = (j.w + 63) / 64;
                t.g2y = (j.h + 3) / 4;
                t.n = j.n;
                gx_contrast = std::max(gx_contrast, t.rblocks);
                g2x = std::max(g2x, t.g2x);
                g2y = std::max(g2y, t.g2y);
                gx_nms = std::max(gx_nms, t.nblocks);
            }
            CVHIP_TRY_HIP(hipMemcpyAsync(d_tbl, h_tbl, (size_t)n_images * sizeof(OrbJobDev), hipMemcpyHostToDevice, s));
            const OrbJobDev *tbl = d_tbl;
            hipLaunchKernelGGL(orb_stage_a_init_jobs, dim3(1), dim3(256), 0, s, tbl, n_images);
            hipLaunchKernelGGL(minmax_jobs, dim3(std::min(gx_contrast, 512u), 1, n_images), dim3(256), 0, s, tbl);
            hipLaunchKernelGGL(contrast_jobs, dim3(gx_contrast, 1, n_images), dim3(256), 0, s, tbl);
            hipLaunchKernelGGL(fast_score_jobs, dim3(g2x, g2y, n_images), dim3(256), 0, s, tbl);
            hipLaunchKernelGGL(nms_count_jobs, dim3(gx_nms, 1, n_images), dim3(256), 0, s, tbl);
            hipLaunchKernelGGL(exclusive_scan_jobs, dim3(1, 1, n_images), dim3(1024), 0, s, tbl);
            CVHIP_TRY_HIP(hipMemcpyAsync(h_counts, d_totals, (size_t)n_images * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        } else {
            CVHIP_TRY(fork());
            for (uint32_t i = 0; i < n_images; i++) {
                OrbJob &j = jobs[i];
                hipStream_t s = lane_stream[i % lanes]; // (this image's chain)
                j.img = imgs[i];
                j.w = ws[i];
                j.h = hs[i];
                j.cap = cap;
                j.out_xy = out_xy[i];
                j.out_desc = out_desc[i];
                j.out_n = &out_n[i];
                j.n = (size_t)j.w * j.h;
                j.nblocks = (uint32_t)((j.n + 255) / 256);
                CVHIP_TRY_HIP(mem.alloc(&j.d_img, j.n + IMG_PAD));
                CVHIP_TRY_HIP(mem.alloc(&j.d_adj, j.n + IMG_PAD));
                CVHIP_TRY_HIP(mem.alloc(&j.d_score, j.n));
                CVHIP_TRY_HIP(mem.alloc(&j.d_mm, 2));
                CVHIP_TRY_HIP(mem.alloc(&j.d_counts, j.nblocks));
                CVHIP_TRY_HIP(mem.alloc(&j.d_total, 1));
                CVHIP_TRY_HIP(hipMemcpyAsync(j.d_img, j.img, j.n, dev_ptr(j.img) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
                CVHIP_TRY_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(j.d_mm), 255, 1, s)); // {min, max} = {255, 0}
                CVHIP_TRY_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(j.d_mm + 1), 0, 1, s));
                const unsigned rblocks = (unsigned)std::min<size_t>(2048, (j.n + 255) / 256);
                hipLaunchKernelGGL(minmax_kernel, dim3(std::min(rblocks, 512u)), dim3(256), 0, s, j.d_img, j.n, j.d_mm);
                hipLaunchKernelGGL(contrast_kernel, dim3(rblocks), dim3(256), 0, s, j.d_img, j.n, j.d_mm, j.d_adj);
                dim3 grid2d((j.w + 63) / 64, (j.h + 3) / 4);
                hipLaunchKernelGGL(fast_score_kernel, grid2d, dim3(256), 0, s, j.d_adj, j.w, j.h, j.d_score);
                hipLaunchKernelGGL(nms_count_kernel, dim3(j.nblocks), dim3(256), 0, s, j.d_score, j.w, j.h, j.d_counts);
                hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, s, j.d_counts, j.nblocks, j.d_total);
                CVHIP_TRY_HIP(hipMemcpyAsync(h_counts + i, j.d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            }
            CVHIP_TRY(join());
        }
        report(0.20f);
        CVHIP_TRY_HIP(hipStreamSynchronize(s));
        CVHIP_TRY_HIP(hipGetLastError());
        report(0.25f);

        // ---- stage B: corner lists, Harris on the original image, stable descending sort, top MAX_KEYPOINTS, blur of
        // the original image, patch moments and the orientation from the device's math library, descriptors, ordered
        // compaction -> results, and the number of keypoints whose descriptor could depend on the last bits of
        // sin / cos (brief_kernel's guard).  No host round trip in between.
        Taps7 k7;
        gaussian_kernel_host(HARRIS_KERNEL_WIDTH, k7.k);
        Taps11 k11;
        gaussian_kernel_host(ORB_GAUSS_KERNEL_WIDTH, k11.k);
        size_t stage_bytes = 0, packs_bytes = 0;
        std::vector<uint32_t> n_fast_of(n_images);
        for (uint32_t i = 0; i < n_images; i++) n_fast_of[i] = h_counts[i]; // (before the staging block may move)
        for (uint32_t i = 0; i < n_images; i++) { // the packs of all images first - one contiguous block on both sides
            OrbJob &j = jobs[i];
            j.n_fast = n_fast_of[i];
            j.count = std::min(j.n_fast, MAX_KEYPOINTS); // entries past the Some(...) ones carry idx = ~0
            j.out_cap = std::min(j.cap, j.count);
            j.pack_bytes = (256 + (size_t)j.out_cap * 10 * sizeof(uint32_t) + 255) / 256 * 256; // [n_out, open count + outputs]
            j.pack_off = packs_bytes;
            packs_bytes += j.pack_bytes;
        }
        stage_bytes = packs_bytes;
        for (uint32_t i = 0; i < n_images; i++) {
            OrbJob &j = jobs[i];
            // page-locked staging of the host-orientation path: [moments + keypoint, 5 u64 per rank] [sin, cos, valid: 3 f64 per rank]
            j.mom_bytes = (size_t)j.count * 5 * sizeof(unsigned long long);
            j.sc_bytes = (size_t)j.count * 3 * sizeof(double);
            j.mom_off = stage_bytes;
            j.sc_off = j.mom_off + j.mom_bytes;
            stage_bytes = (j.sc_off + j.sc_bytes + 255) / 256 * 256;
        }
        // (pinned_scratch may move the block: h_counts / h_tbl of stage A are not used past this point)
        char *stage = static_cast<char *>(pinned_scratch(dev->d, stage_bytes + 512 + (size_t)n_images * sizeof(OrbJobDev)));
        if (!stage) return fail(CVHIP_ERR_NOMEM, "cvhip_orb_extract: out of page-locked host memory");
        h_tbl = reinterpret_cast<OrbJobDev *>(stage + ((stage_bytes + 256 + 15) & ~(size_t)15));
        if (!dev->d.orb_pattern) {
            CVHIP_TRY_HIP(hipMalloc(&dev->d.orb_pattern, 1024));
            CVHIP_TRY_HIP(hipMemcpyAsync(dev->d.orb_pattern, CVHIP_ORB_PATTERN, 1024, hipMemcpyHostToDevice, s));
        }
        const double guard = dev->d.orb_guard;
        // descriptors of image j from the orientations in j.d_sc, results into the image's staging block
        const auto describe = [&](OrbJob &j, double g, hipStream_t s) -> int {
            uint32_t *d_out_n = j.d_pack, *d_out_xy = j.xy_dev ? j.out_xy : j.d_pack + 64,
                     *d_out_desc = j.desc_dev ? j.out_desc : j.d_pack + 64 + (size_t)j.out_cap * 2;
            CVHIP_TRY_HIP(hipMemsetAsync(j.d_pack, 0, 8, s)); // {n_out, open count}
            hipLaunchKernelGGL(brief_kernel, dim3(j.count), dim3(64), 0, s, j.d_blur, j.w, j.h, j.d_kp, j.d_idx_sorted, j.count, j.n_fast, j.d_sc,
                               (const signed char *)dev->d.orb_pattern, j.d_desc, j.d_flags, g, j.d_pack + 1);
            hipLaunchKernelGGL(final_compact_kernel, dim3(1), dim3(1024), 0, s, j.d_flags, j.d_kp, j.d_idx_sorted, j.d_desc, j.count,
                               j.out_cap, d_out_xy, d_out_desc, d_out_n);
            const size_t back = (j.xy_dev && j.desc_dev) ? 256 : j.pack_bytes;
            CVHIP_TRY_HIP(hipMemcpyAsync(stage + j.pack_off, j.d_pack, back, hipMemcpyDeviceToHost, s));
            return CVHIP_OK;
        };
        if (batched) {
            // every image's buffers; the corner keys of all images in one array (image i at key_off), their packs in one block
            uint32_t total_keys = 0, gx_nms = 1, gx_harris = 1, g2x = 1, g2y = 1, gx_count = 1;
            uint32_t *d_packs = nullptr;
            CVHIP_TRY_HIP(mem.alloc(&d_packs, packs_bytes / sizeof(uint32_t) + 64));
            std::vector<uint32_t> key_off(n_images);
            for (uint32_t i = 0; i < n_images; i++) {
                key_off[i] = total_keys;
                total_keys += jobs[i].n_fast;
            }
            unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
            uint32_t *d_idx = nullptr, *d_idx_s1 = nullptr, *d_idx_s2 = nullptr, *d_idx_plain = nullptr, *d_rank = nullptr;
            CVHIP_TRY_HIP(mem.alloc(&d_keys, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_keys_sorted, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_idx, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_idx_s1, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_idx_s2, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_idx_plain, std::max(total_keys, 1u)));
            CVHIP_TRY_HIP(mem.alloc(&d_rank, std::max(total_keys, 1u)));
            for (uint32_t i = 0; i < n_images; i++) {
                OrbJob &j = jobs[i];
                OrbJobDev &t = h_tbl[i];
                std::memset(&t, 0, sizeof(t));
                j.d_pack = d_packs + j.pack_off / sizeof(uint32_t);
                j.xy_dev = dev_ptr(j.out_xy);
                j.desc_dev = dev_ptr(j.out_desc);
                j.d_idx_sorted = d_idx_plain + key_off[i];
                double *d_blur_h = nullptr;
                if (j.n_fast) {
                    CVHIP_TRY_HIP(mem.alloc(&j.d_kp, (size_t)j.n_fast * 2));
                    CVHIP_TRY_HIP(mem.alloc(&d_blur_h, j.n));
                    CVHIP_TRY_HIP(mem.alloc(&j.d_blur, j.n));
                    CVHIP_TRY_HIP(mem.alloc(&j.d_mom, (size_t)j.count * 5));
                    CVHIP_TRY_HIP(mem.alloc(&j.d_sc, (size_t)j.count * 3));
                    CVHIP_TRY_HIP(mem.alloc(&j.d_desc, (size_t)j.count * 8));
                    CVHIP_TRY_HIP(mem.alloc(&j.d_flags, j.count));
                }
                t.img = j.d_img;
                t.score = j.d_score;
                t.counts = j.d_counts;
                t.total = j.d_total;
                t.w = j.w;
                t.h = j.h;
                t.nblocks = j.nblocks;
                t.g2x = (j.w + 63) / 64;
                t.g2y = (j.h + 3) / 4;
                t.n = j.n;
                t.n_fast = j.n_fast;
                t.count = j.count;
                t.out_cap = j.out_cap;
                t.key_off = key_off[i];
                t.kp = j.d_kp;
                t.idx_sorted = j.d_idx_sorted;
                t.desc = j.d_desc;
                t.flags = j.d_flags;
                t.pack = j.d_pack;
                t.out_xy = j.xy_dev ? j.out_xy : j.d_pack + 64;
                t.out_desc = j.desc_dev ? j.out_desc : j.d_pack + 64 + (size_t)j.out_cap * 2;
                t.keys = d_keys + key_off[i];
                t.idx = d_idx + key_off[i];
                t.mom = j.d_mom;
                t.blur_h = d_blur_h;
                t.blur = j.d_blur;
                t.sc = j.d_sc;
                if (j.n_fast) {
                    gx_nms = std::max(gx_nms, j.nblocks);
                    gx_harris = std::max(gx_harris, (j.n_fast + 63) / 64);
                    g2x = std::max(g2x, t.g2x);
                    g2y = std::max(g2y, t.g2y);
                    gx_count = std::max(gx_count, j.count);
                }
            }
            CVHIP_TRY_HIP(hipMemcpyAsync(d_tbl, h_tbl, (size_t)n_images * sizeof(OrbJobDev), hipMemcpyHostToDevice, s));
            const OrbJobDev *tbl = d_tbl;
            if (total_keys) {
                // The blur depends on the images only: on a side stream it runs under the corner chain, whose two sorts are
                // ~40 launches of 5 us each.  (Streams that share a hardware queue run it in line, as before.)
                hipStream_t side = nullptr;
                CVHIP_TRY_HIP(aux_stream(dev->d, 0, &side));
                for (int l = 0; l < 2; l++)
                    if (!dev->d.orb_ev[l]) CVHIP_TRY_HIP(hipEventCreateWithFlags(&dev->d.orb_ev[l], hipEventDisableTiming));
                CVHIP_TRY_HIP(hipEventRecord(dev->d.orb_ev[0], s));
                CVHIP_TRY_HIP(hipStreamWaitEvent(side, dev->d.orb_ev[0], 0));
                hipLaunchKernelGGL(blur_h_jobs, dim3(g2x, g2y, n_images), dim3(256), 0, side, tbl, k11);
                hipLaunchKernelGGL(blur_v_jobs, dim3(g2x, g2y, n_images), dim3(256), 0, side, tbl, k11);
                CVHIP_TRY_HIP(hipEventRecord(dev->d.orb_ev[1], side));
                hipLaunchKernelGGL(nms_write_jobs, dim3(gx_nms, 1, n_images), dim3(256), 0, s, tbl);
                hipLaunchKernelGGL(harris_jobs, dim3(gx_harris, 1, n_images), dim3(64), 0, s, tbl, k7);
                // All images' corners in two sorts: descending by the Harris key (stable: equal keys keep their scan order),
                // then the ranks of that order by (image, rank) - which leaves every image's corners together, in descending
                // key order: what the per-image sort produced.
                size_t tmp1 = 0, tmp2 = 0;
                CVHIP_TRY_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp1, d_keys, d_keys_sorted, d_idx, d_idx_s1, (size_t)total_keys, 0u, 64u, s));
                CVHIP_TRY_HIP(rocprim::radix_sort_keys(nullptr, tmp2, d_rank, d_idx_s2, (size_t)total_keys, 0u, 32u, s));
                uint8_t *d_tmp = nullptr, *d_tmp2 = nullptr;
                CVHIP_TRY_HIP(mem.alloc(&d_tmp, tmp1));
                CVHIP_TRY_HIP(mem.alloc(&d_tmp2, tmp2));
                const dim3 gk((total_keys + 255) / 256);
                CVHIP_TRY_HIP(rocprim::radix_sort_pairs_desc(d_tmp, tmp1, d_keys, d_keys_sorted, d_idx, d_idx_s1, (size_t)total_keys, 0u, 64u, s));
                hipLaunchKernelGGL(orb_rank_key_kernel, gk, dim3(256), 0, s, (const uint32_t *)d_idx_s1, total_keys, d_rank);
                CVHIP_TRY_HIP(rocprim::radix_sort_keys(d_tmp2, tmp2, d_rank, d_idx_s2, (size_t)total_keys, 0u, 32u, s));
                hipLaunchKernelGGL(orb_untag_kernel, gk, dim3(256), 0, s, (const uint32_t *)d_idx_s1, (const uint32_t *)d_idx_s2, total_keys, d_idx_plain);
                CVHIP_TRY_HIP(hipStreamWaitEvent(s, dev->d.orb_ev[1], 0));
                hipLaunchKernelGGL(moments_jobs, dim3(gx_count, 1, n_images), dim3(64), 0, s, tbl);
                report(0.35f);
                if (guard > 0.0) {
                    hipLaunchKernelGGL(brief_jobs, dim3(gx_count, 1, n_images), dim3(64), 0, s, tbl, (const signed char *)dev->d.orb_pattern, guard);
                    hipLaunchKernelGGL(final_compact_jobs, dim3(1, 1, n_images), dim3(1024), 0, s, tbl);
                    CVHIP_TRY_HIP(hipMemcpyAsync(stage, d_packs, packs_bytes, hipMemcpyDeviceToHost, s)); // every image's results
                } else { // device orientations switched off (cvhip_orb_set_orientation_guard(dev, 0)): the host path for all
                    for (OrbJob &j : jobs)
                        if (j.n_fast) CVHIP_TRY_HIP(hipMemcpyAsync(stage + j.mom_off, j.d_mom, j.mom_bytes, hipMemcpyDeviceToHost, s));
                }
            }
        } else {
        CVHIP_TRY(fork()); // (behind the pattern upload)
        for (uint32_t i = 0; i < n_images; i++) {
            OrbJob &j = jobs[i];
            if (j.n_fast == 0) continue;
            hipStream_t s = lane_stream[i % lanes];
            CVHIP_TRY_HIP(mem.alloc(&j.d_kp, (size_t)j.n_fast * 2));
            hipLaunchKernelGGL(nms_write_kernel, dim3(j.nblocks), dim3(256), 0, s, j.d_score, j.w, j.h, j.d_counts, j.n_fast, j.d_kp);
            unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
            uint32_t *d_idx = nullptr;
            CVHIP_TRY_HIP(mem.alloc(&d_keys, j.n_fast));
            CVHIP_TRY_HIP(mem.alloc(&d_keys_sorted, j.n_fast));
            CVHIP_TRY_HIP(mem.alloc(&d_idx, j.n_fast));
            CVHIP_TRY_HIP(mem.alloc(&j.d_idx_sorted, j.n_fast));
            hipLaunchKernelGGL(harris_kernel, dim3((j.n_fast + 63) / 64), dim3(64), 0, s, j.d_img, j.w, j.h, j.d_kp, j.d_total, j.n_fast,
                               k7, d_keys, d_idx);
            size_t tmp_bytes = 0;
            CVHIP_TRY_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_idx, j.d_idx_sorted,
                                                         (size_t)j.n_fast, 0u, 64u, s));
            uint8_t *d_tmp = nullptr;
            CVHIP_TRY_HIP(mem.alloc(&d_tmp, tmp_bytes));
            CVHIP_TRY_HIP(rocprim::radix_sort_pairs_desc(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_idx, j.d_idx_sorted,
                                                         (size_t)j.n_fast, 0u, 64u, s));
            double *d_blur_h = nullptr;
            CVHIP_TRY_HIP(mem.alloc(&d_blur_h, j.n));
            CVHIP_TRY_HIP(mem.alloc(&j.d_blur, j.n));
            dim3 grid2d((j.w + 63) / 64, (j.h + 3) / 4);
            hipLaunchKernelGGL(blur_h_kernel, grid2d, dim3(256), 0, s, j.d_img, j.w, j.h, k11, d_blur_h);
            hipLaunchKernelGGL(blur_v_kernel, grid2d, dim3(256), 0, s, d_blur_h, j.w, j.h, k11, j.d_blur);
            CVHIP_TRY_HIP(mem.alloc(&j.d_mom, (size_t)j.count * 5));
            CVHIP_TRY_HIP(mem.alloc(&j.d_sc, (size_t)j.count * 3));
            CVHIP_TRY_HIP(mem.alloc(&j.d_desc, (size_t)j.count * 8));
            CVHIP_TRY_HIP(mem.alloc(&j.d_flags, j.count));
            CVHIP_TRY_HIP(mem.alloc(&j.d_pack, j.pack_bytes / sizeof(uint32_t)));
            j.xy_dev = dev_ptr(j.out_xy);
            j.desc_dev = dev_ptr(j.out_desc);
            hipLaunchKernelGGL(moments_kernel, dim3(j.count), dim3(64), 0, s, j.d_blur, j.w, j.h, j.d_kp, j.d_idx_sorted, j.count, j.n_fast, j.d_mom,
                               j.d_sc);
            if (i == 0) report(0.35f);
            if (guard > 0.0) {
                CVHIP_TRY(describe(j, guard, s));
            } else { // device orientations switched off (cvhip_orb_set_orientation_guard(dev, 0)): the host path for all
                CVHIP_TRY_HIP(hipMemcpyAsync(stage + j.mom_off, j.d_mom, j.mom_bytes, hipMemcpyDeviceToHost, s));
            }
        }
        CVHIP_TRY(join());
        }
        report(0.70f);
        CVHIP_TRY_HIP(hipStreamSynchronize(s));
        CVHIP_TRY_HIP(hipGetLastError());

        // ---- the images with a keypoint inside the guard band (rare), or all of them with the guard off: orientation
        // on the host with libm, exactly as the reference (orb.rs:337-341, 365-366), and their descriptors again
        std::vector<OrbJob *> redo;
        // (the describe step's status word: open-count | ORB_BAD_INDEX; written by every describe(), first pass or redo)
        const auto status_of = [&](const OrbJob &j) { return reinterpret_cast<const uint32_t *>(stage + j.pack_off)[1]; };
        const char *const bad_index = "orb_extract: a sorted corner index is outside the corner list (device sort failed)";
        for (OrbJob &j : jobs)
            if (j.n_fast && guard > 0.0 && (status_of(j) & ORB_BAD_INDEX)) return fail(CVHIP_ERR_DEVICE, bad_index);
        for (OrbJob &j : jobs)
            if (j.n_fast && (guard <= 0.0 || (status_of(j) & ~ORB_BAD_INDEX) != 0u)) redo.push_back(&j);
        if (!redo.empty()) {
            if (guard > 0.0) {
                for (OrbJob *j : redo) CVHIP_TRY_HIP(hipMemcpyAsync(stage + j->mom_off, j->d_mom, j->mom_bytes, hipMemcpyDeviceToHost, s));
                CVHIP_TRY_HIP(hipStreamSynchronize(s));
            }
            // (glibc's atan2 / sin / cos are ~70 ns each; the keypoints are independent, so a few host threads share them)
            struct Span {
                const unsigned long long *mom;
                double *sc;
                uint32_t r0, r1;
            };
            std::vector<Span> spans;
            uint32_t total = 0;
            for (const OrbJob *j : redo) total += j->count;
            const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
            const uint32_t nthreads = std::min({8u, hw, total / 2048u + 1u});
            const uint32_t per = std::max((total + nthreads - 1) / std::max(nthreads, 1u), 1u);
            for (const OrbJob *j : redo)
                for (uint32_t r0 = 0; r0 < j->count; r0 += per)
                    spans.push_back(Span{reinterpret_cast<const unsigned long long *>(stage + j->mom_off),
                                         reinterpret_cast<double *>(stage + j->sc_off), r0, std::min(j->count, r0 + per)});
            const auto orient = [](const Span &sp) {
                for (uint32_t r = sp.r0; r < sp.r1; r++) {
                    const unsigned long long *m = sp.mom + 5 * (size_t)r;
                    double *o = sp.sc + 3 * (size_t)r;
                    o[0] = o[1] = o[2] = 0.0;
                    if (!m[3]) continue; // (an entry past the Some(...) ones has no valid patch either)
                    const double x = (double)(uint32_t)m[4], y = (double)(uint32_t)(m[4] >> 32);
                    const double m00 = (double)m[0];
                    const double centroid_x = (double)m[1] / m00;
                    const double centroid_y = (double)m[2] / m00;
                    const double angle = std::atan2(centroid_y - y, centroid_x - x);
                    o[0] = std::sin(angle);
                    o[1] = std::cos(angle);
                    o[2] = 1.0;
                }
            };
            std::vector<std::thread> pool;
            size_t next = 1;
            try {
                for (; next < spans.size(); next++) pool.emplace_back(orient, spans[next]);
            } catch (const std::system_error &) { // no more threads: the caller's thread does what was not handed out
                for (size_t k = next; k < spans.size(); k++) orient(spans[k]);
            }
            if (!spans.empty()) orient(spans[0]);
            for (auto &th : pool) th.join();
            for (OrbJob *j : redo) {
                CVHIP_TRY_HIP(hipMemcpyAsync(j->d_sc, stage + j->sc_off, j->sc_bytes, hipMemcpyHostToDevice, s));
                CVHIP_TRY(describe(*j, 0.0, s));
            }
            CVHIP_TRY_HIP(hipStreamSynchronize(s));
            CVHIP_TRY_HIP(hipGetLastError());
            // the redone descriptions report a bad index the same way - with the guard off they are the only ones
            for (const OrbJob *j : redo)
                if (status_of(*j) & ORB_BAD_INDEX) return fail(CVHIP_ERR_DEVICE, bad_index);
        }
        for (uint32_t i = 0; i < n_images; i++) {
            OrbJob &j = jobs[i];
            if (j.n_fast == 0) {
                *j.out_n = 0;
                continue;
            }
            const char *h_pack = stage + j.pack_off;
            const uint32_t n_out = *reinterpret_cast<const uint32_t *>(h_pack);
            if (!j.xy_dev && n_out) std::memcpy(j.out_xy, h_pack + 256, (size_t)n_out * 2 * sizeof(uint32_t));
            if (!j.desc_dev && n_out)
                std::memcpy(j.out_desc, h_pack + 256 + (size_t)j.out_cap * 2 * sizeof(uint32_t), (size_t)n_out * 8 * sizeof(uint32_t));
            *j.out_n = n_out;
        }
        report(1.0f);
        if (measuring) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            if (ol.samples == 0 || ms < 0.95 * ol.best_ms) { // (fewer lanes only where they are clearly faster)
                ol.best_ms = ms;
                ol.use_lanes = lanes;
            }
            ol.decided = ++ol.samples == LANES;
        }
        return CVHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(CVHIP_ERR_NOMEM, "cvhip_orb_extract: out of host memory");
    }
}

extern "C" int cvhip_orb_set_orientation_guard(cvhip_device *dev, double guard)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "dev is null");
    if (!(guard >= 0.0) || guard > 1.0) return fail(CVHIP_ERR_INVALID, "guard must be in [0, 1]");
    dev->d.orb_guard = guard;
    return CVHIP_OK;
}

extern "C" int cvhip_orb_extract(cvhip_device *dev, const uint8_t *img, uint32_t w, uint32_t h, uint32_t cap,
                                 uint32_t *out_xy, uint32_t *out_desc, uint32_t *out_n, cvhip_progress_fn progress, void *user)
{
    if (!dev || !img || !out_xy || !out_desc || !out_n) return fail(CVHIP_ERR_INVALID, "null argument");
    return cvhip_orb_extract_batch(dev, 1, &img, &w, &h, cap, &out_xy, &out_desc, out_n, progress, user);
}

extern "C" int cvhip_match_points(cvhip_device *dev, const uint32_t *xy1, const uint32_t *desc1, uint32_t n1,
                                  const uint32_t *xy2, const uint32_t *desc2, uint32_t n2, uint32_t threshold,
                                  uint32_t *out_matches, uint32_t *out_dist, uint32_t *out_n)
{
    if (!dev || !out_matches || !out_n) return fail(CVHIP_ERR_INVALID, "null argument");
    *out_n = 0;
    if (n1 == 0 || n2 == 0) return CVHIP_OK;
    if (!xy1 || !desc1 || !xy2 || !desc2) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    uint32_t *d_xy1, *d_xy2, *d_desc1, *d_desc2, *d_bj, *d_bd, *d_bd_sorted, *d_q, *d_q_sorted, *d_om, *d_od, *d_n;
    CVHIP_TRY_HIP(mem.alloc(&d_xy1, (size_t)n1 * 2));
    CVHIP_TRY_HIP(mem.alloc(&d_xy2, (size_t)n2 * 2));
    CVHIP_TRY_HIP(mem.alloc(&d_desc1, (size_t)n1 * 8));
    CVHIP_TRY_HIP(mem.alloc(&d_desc2, (size_t)n2 * 8));
    CVHIP_TRY_HIP(mem.alloc(&d_bj, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_bd, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_bd_sorted, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_q, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_q_sorted, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_om, (size_t)n1 * 4));
    CVHIP_TRY_HIP(mem.alloc(&d_od, n1));
    CVHIP_TRY_HIP(mem.alloc(&d_n, 1));
    auto kind = [](const void *p) { return dev_ptr(p) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice; };
    CVHIP_TRY_HIP(hipMemcpyAsync(d_xy1, xy1, (size_t)n1 * 8, kind(xy1), s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_xy2, xy2, (size_t)n2 * 8, kind(xy2), s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_desc1, desc1, (size_t)n1 * 32, kind(desc1), s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_desc2, desc2, (size_t)n2 * 32, kind(desc2), s));
    CVHIP_TRY_HIP(hipMemsetAsync(d_n, 0, sizeof(uint32_t), s));
    unsigned long long *d_key = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_key, n1));
    CVHIP_TRY_HIP(hipMemsetAsync(d_key, 0xFF, (size_t)n1 * sizeof(unsigned long long), s)); // no match
    const uint32_t qblocks = (n1 + 255) / 256;
    if ((size_t)n1 * n2 >= (size_t)1 << 22 && threshold < MM_PAD_COUNT) {
        // the matrix-pipe form (see match_mfma_kernel): descriptors as bytes, 128 queries per workgroup, the candidate list
        // split so that the launch has ~4 workgroups per CU (a split covers at most 65 536 candidates: 16 bits of the key)
        const uint32_t n1_pad = (n1 + 127u) / 128u * 128u, n2_pad = (n2 + MM_ROWS - 1) / MM_ROWS * MM_ROWS;
        uint32_t *d_b1 = nullptr, *d_b2 = nullptr, *d_p1 = nullptr, *d_p2 = nullptr;
        CVHIP_TRY_HIP(mem.alloc(&d_b1, (size_t)n1_pad * 64));
        CVHIP_TRY_HIP(mem.alloc(&d_b2, (size_t)n2_pad * 64));
        CVHIP_TRY_HIP(mem.alloc(&d_p1, n1_pad));
        CVHIP_TRY_HIP(mem.alloc(&d_p2, n2_pad));
        hipLaunchKernelGGL(match_expand_kernel, dim3((n1_pad * 8 + 255) / 256), dim3(256), 0, s, d_desc1, n1, n1_pad, 1u, d_b1, d_p1);
        hipLaunchKernelGGL(match_expand_kernel, dim3((n2_pad * 8 + 255) / 256), dim3(256), 0, s, d_desc2, n2, n2_pad, 0xFFu, d_b2, d_p2);
        const uint32_t wgs = n1_pad / 128u, tiles = n2_pad / MM_ROWS;
        uint32_t splits = std::max(1u, std::min(tiles, (1024u + wgs - 1) / wgs));
        uint32_t chunk = (tiles + splits - 1) / splits * MM_ROWS;
        chunk = std::min(chunk, 65536u);
        hipLaunchKernelGGL(match_mfma_kernel, dim3(wgs, (n2_pad + chunk - 1) / chunk), dim3(256), 0, s, (const uint32_t *)d_b1,
                           (const uint32_t *)d_p1, n1, (const uint32_t *)d_b2, (const uint32_t *)d_p2, n2_pad, chunk, threshold, d_key);
    } else {
        const uint32_t tiles = (n2 + MATCH_TILE - 1) / MATCH_TILE;
        const uint32_t splits = std::max(1u, std::min(tiles, (2048u + qblocks - 1) / qblocks)); // >= ~2 workgroups per SIMD
        const uint32_t chunk = (tiles + splits - 1) / splits * MATCH_TILE;
        hipLaunchKernelGGL(match_kernel, dim3(qblocks, (n2 + chunk - 1) / chunk), dim3(256), 0, s, d_desc1, n1, d_desc2, n2, chunk,
                           threshold, d_key);
    }
    hipLaunchKernelGGL(match_unpack_kernel, dim3(qblocks), dim3(256), 0, s, (const unsigned long long *)d_key, n1, d_bj, d_bd,
                       d_q);
    // stable ascending sort by distance (sort_by_key, pointmatching.rs:74); unmatched = ~0 go last
    size_t tmp_bytes = 0;
    CVHIP_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_bd, d_bd_sorted, d_q, d_q_sorted, (size_t)n1, 0u, 32u, s));
    uint8_t *d_tmp = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_tmp, tmp_bytes));
    CVHIP_TRY_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_bd, d_bd_sorted, d_q, d_q_sorted, (size_t)n1, 0u, 32u, s));
    hipLaunchKernelGGL(match_gather_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, d_q_sorted, d_bd_sorted, d_bj, d_xy1,
                       d_xy2, n1, d_om, d_od, d_n);
    uint32_t n = 0;
    CVHIP_TRY_HIP(hipMemcpyAsync(&n, d_n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipStreamSynchronize(s));
    CVHIP_TRY_HIP(hipGetLastError());
    if (n) {
        CVHIP_TRY_HIP(hipMemcpy(out_matches, d_om, (size_t)n * 16,
                                dev_ptr(out_matches) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
        if (out_dist)
            CVHIP_TRY_HIP(hipMemcpy(out_dist, d_od, (size_t)n * 4,
                                    dev_ptr(out_dist) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    }
    *out_n = n;
    return CVHIP_OK;
}
