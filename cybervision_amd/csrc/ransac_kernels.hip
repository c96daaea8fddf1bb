// ransac_kernels.hip — RANSAC hypothesis scoring for gfx950.
//
// Replaces the all-matches fold of FundamentalMatrix::validate_f
// (zlogic/cybervision src/fundamentalmatrix.rs:210-216) with fits_model / reprojection_error
// (:452-471) for a whole batch of hypotheses.  One lane per hypothesis: its F sits in
// registers, the match list streams through LDS in tiles that every lane reads at the same
// address (LDS broadcast), and count/error-sum are folded serially in match order — the same
// order as the reference's iterator fold, so both outputs are bit-identical to --mode=cpu.
// f64 throughout, contraction off; expression order follows nalgebra 0.35's
// published gemv / dot algorithms (column-by-column axpy; 3-vector dot = (a0*b0 + a1*b1) + a2*b2).
#include "cvhip_internal.hpp"

#include <cstring>
#include <vector>

namespace cvhip {

constexpr int RANSAC_TILE = 1024; // matches per LDS tile (16 KiB as 4 x u32)

__device__ __forceinline__ double reprojection_error(const double (&F)[9], double p1x, double p1y, double p2x,
                                                     double p2y)
{
    // p2.tr_mul(f): element j = dot(p2, F[:, j]) = (p2x*F0j + p2y*F1j) + 1*F2j
    const double r0 = (p2x * F[0] + p2y * F[3]) + F[6];
    const double r1 = (p2x * F[1] + p2y * F[4]) + F[7];
    const double r2 = (p2x * F[2] + p2y * F[5]) + F[8];
    // (1x3) * p1, gemv order
    double n = r0 * p1x;
    n = r1 * p1y + n;
    n = r2 + n;
    // f * p1, rows 0 and 1
    double a0 = F[0] * p1x;
    a0 = F[1] * p1y + a0;
    a0 = F[2] + a0;
    double a1 = F[3] * p1x;
    a1 = F[4] * p1y + a1;
    a1 = F[5] + a1;
    // f.tr_mul(p2): element i = dot(F[:, i], p2), i = 0, 1
    const double b0 = (F[0] * p2x + F[3] * p2y) + F[6];
    const double b1 = (F[1] * p2x + F[4] * p2y) + F[7];
    const double nominator = n * n;
    const double denominator = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
    return nominator / denominator;
}

__global__ __launch_bounds__(64) void ransac_score_kernel(const double *__restrict__ F, uint32_t H,
                                                           const uint4 *__restrict__ matches, uint32_t N, double t,
                                                           uint32_t *__restrict__ out_count,
                                                           double *__restrict__ out_err_sum)
{
    __shared__ uint4 tile[RANSAC_TILE];
    const uint32_t h = blockIdx.x * 64 + threadIdx.x;
    const bool active = h < H;
    double f[9];
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = active ? F[(size_t)h * 9 + i] : 0.0;
    uint32_t count = 0;
    double sum = 0.0;
    for (uint32_t base = 0; base < N; base += RANSAC_TILE) {
        const uint32_t n = min((uint32_t)RANSAC_TILE, N - base);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 64) tile[i] = matches[base + i];
        __syncthreads();
        if (active) {
            for (uint32_t i = 0; i < n; i++) {
                const uint4 m = tile[i];
                const double err = reprojection_error(f, (double)m.x, (double)m.y, (double)m.z, (double)m.w);
                // fits_model: finite and |err| <= t (fundamentalmatrix.rs:452-458)
                if (fabs(err) < __builtin_inf() && !(fabs(err) > t)) {
                    count += 1;
                    sum += err;
                }
            }
        }
    }
    if (active) {
        out_count[h] = count;
        out_err_sum[h] = sum;
    }
}

void launch_ransac_score(const double *F, uint32_t H, const uint32_t *matches, uint32_t N, double t,
                         uint32_t *out_count, double *out_err_sum, hipStream_t s)
{
    if (!H) return;
    hipLaunchKernelGGL(ransac_score_kernel, dim3((H + 63) / 64), dim3(64), 0, s, F, H,
                       reinterpret_cast<const uint4 *>(matches), N, t, out_count, out_err_sum);
}


// ---------------------------------------------------------------------------------------------
// Whole affine RANSAC on the device: hypothesis generation moves next
// to the scoring kernel, so the 10^6-iteration loop of FundamentalMatrix::find_ransac
// (fundamentalmatrix.rs:103-147) never leaves the GPU except for one 4-byte early-exit check per
// 50 000-iteration round.  Per hypothesis (one thread): choose_inliers (:155-175, rejection
// sampling from the top 5000 matches, >= 10 px apart), calculate_model_affine (:260-286: mean-centred
// 4x4, right-singular vector of the smallest singular value — here via Jacobi on A^T A), validate_f's
// finiteness and sample-fit checks (:197-209).  The reference seeds its RNG from the OS and is not
// reproducible run to run, so parity for this row is statistical (tests compare with the known model).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ bool affine_model_from_sample(const uint4 (&sm)[4], double (&f)[9])
{
    double a[4][4], mean[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) { // rows (x2, y2, x1, y1), fundamentalmatrix.rs:262-268
        a[i][0] = (double)sm[i].z;
        a[i][1] = (double)sm[i].w;
        a[i][2] = (double)sm[i].x;
        a[i][3] = (double)sm[i].y;
#pragma unroll
        for (int j = 0; j < 4; j++) mean[j] += a[i][j] / 4.0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) a[i][j] -= mean[j];
    double m[4][4], v[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) acc += a[k][i] * a[k][j];
            m[i][j] = acc;
            v[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; sweep++) { // cyclic Jacobi, fixed pivot order (static indices)
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) off += m[p][q] * m[p][q];
        if (off < 1e-280) break;
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                if (fabs(m[p][q]) < 1e-300) continue;
                const double theta = (m[q][q] - m[p][p]) / (2.0 * m[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double mkp = m[k][p], mkq = m[k][q];
                    m[k][p] = c * mkp - sn * mkq;
                    m[k][q] = sn * mkp + c * mkq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double mpk = m[p][k], mqk = m[q][k];
                    m[p][k] = c * mpk - sn * mqk;
                    m[q][k] = sn * mpk + c * mqk;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    // smallest eigenvalue -> null vector; second largest singular value must be >= 1e-3 (:272-275)
    const double ev[4] = {m[0][0], m[1][1], m[2][2], m[3][3]};
    int last = 0;
    double lo = ev[0], hi1 = -1.0, hi2 = -1.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (ev[i] < lo) {
            lo = ev[i];
            last = i;
        }
        if (ev[i] > hi1) {
            hi2 = hi1;
            hi1 = ev[i];
        } else if (ev[i] > hi2) {
            hi2 = ev[i];
        }
    }
    if (sqrt(fmax(hi2, 0.0)) < 0.001) return false;
    double vt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) vt[k] = last == 0 ? v[k][0] : (last == 1 ? v[k][1] : (last == 2 ? v[k][2] : v[k][3]));
    const double e = vt[0] * mean[0] + vt[1] * mean[1] + vt[2] * mean[2] + vt[3] * mean[3];
    const double raw[9] = {0.0, 0.0, vt[0], 0.0, 0.0, vt[1], vt[2], vt[3], -e};
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = raw[i] / raw[8]; // f / f[(2, 2)], :285
    return true;
}

__global__ __launch_bounds__(64) void ransac_generate_affine_kernel(const uint4 *__restrict__ matches, uint32_t limit,
                                                                     double t, unsigned long long seed,
                                                                     uint32_t round, uint32_t H,
                                                                     double *__restrict__ F)
{
    const uint32_t h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    unsigned long long state = mix64(seed ^ mix64(((unsigned long long)round << 32) | h));
    uint4 sm[4];
    int have = 0;
    for (int tries = 0; tries < 256 && have < 4; tries++) { // choose_inliers, :155-175 (bounded here)
        state = mix64(state + 0x9E3779B97F4A7C15ull);
        const uint32_t idx = (uint32_t)(((state >> 32) * (unsigned long long)limit) >> 32);
        const uint4 nm = matches[idx];
        bool close = false;
        for (int i = 0; i < have; i++) {
            const uint4 c = sm[i];
            auto dist = [](uint32_t a, uint32_t b) { return a > b ? a - b : b - a; };
            close = close || dist(nm.x, c.x) < 10u || dist(nm.y, c.y) < 10u || dist(nm.z, c.z) < 10u || dist(nm.w, c.w) < 10u;
        }
        if (!close) {
            if (have == 0) sm[0] = nm;
            else if (have == 1) sm[1] = nm;
            else if (have == 2) sm[2] = nm;
            else sm[3] = nm;
            have++;
        }
    }
    double f[9];
    bool ok = have == 4 && affine_model_from_sample(sm, f);
    if (ok) {
#pragma unroll
        for (int i = 0; i < 9; i++) ok = ok && fabs(f[i]) < __builtin_inf(); // validate_f, :197-199
#pragma unroll
        for (int i = 0; i < 4; i++) { // all sample points must fit, :206-209
            const double err = reprojection_error(f, (double)sm[i].x, (double)sm[i].y, (double)sm[i].z, (double)sm[i].w);
            ok = ok && fabs(err) < __builtin_inf() && !(fabs(err) > t);
        }
    }
    const double nan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)h * 9 + i] = ok ? f[i] : nan; // NaN hypotheses score 0 inliers
}

struct RansacBest {
    double f[9];
    double best_error;
    uint32_t matches_count;
    uint32_t valid;
};

// Ord for RansacIterationResult (fundamentalmatrix.rs:623-649)
__device__ __forceinline__ bool ransac_better(uint32_t ca, double ea, uint32_t cb, double eb)
{
    if (ca != cb) return ca > cb;
    const bool af = fabs(ea) < __builtin_inf(), bf = fabs(eb) < __builtin_inf();
    if (af != bf) return af;
    if (!af) return false;
    return ea < eb;
}

__global__ __launch_bounds__(1024) void ransac_pick_best_kernel(const double *__restrict__ F,
                                                                 const uint32_t *__restrict__ counts,
                                                                 const double *__restrict__ err_sums, uint32_t H,
                                                                 uint32_t min_count, RansacBest *__restrict__ best)
{
    __shared__ uint32_t s_cnt[1024];
    __shared__ double s_err[1024];
    __shared__ uint32_t s_idx[1024];
    uint32_t bc = 0, bi = 0xFFFFFFFFu;
    double be = __builtin_inf();
    for (uint32_t h = threadIdx.x; h < H; h += 1024) {
        const uint32_t c = counts[h];
        if (c < min_count) continue; // :218-220
        const double e = err_sums[h] / (double)c;
        if (bi == 0xFFFFFFFFu || ransac_better(c, e, bc, be)) {
            bc = c;
            be = e;
            bi = h;
        }
    }
    s_cnt[threadIdx.x] = bc;
    s_err[threadIdx.x] = be;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (uint32_t s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const uint32_t o = threadIdx.x + s;
            if (s_idx[o] != 0xFFFFFFFFu &&
                (s_idx[threadIdx.x] == 0xFFFFFFFFu || ransac_better(s_cnt[o], s_err[o], s_cnt[threadIdx.x], s_err[threadIdx.x]))) {
                s_cnt[threadIdx.x] = s_cnt[o];
                s_err[threadIdx.x] = s_err[o];
                s_idx[threadIdx.x] = s_idx[o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_idx[0] != 0xFFFFFFFFu) {
        if (!best->valid || ransac_better(s_cnt[0], s_err[0], best->matches_count, best->best_error)) {
            for (int i = 0; i < 9; i++) best->f[i] = F[(size_t)s_idx[0] * 9 + i];
            best->matches_count = s_cnt[0];
            best->best_error = s_err[0];
            best->valid = 1;
        }
    }
}

__global__ void ransac_inlier_mask_kernel(const RansacBest *__restrict__ best, const uint4 *__restrict__ matches,
                                          uint32_t N, double t, uint8_t *__restrict__ mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = best->f[k];
    const uint4 m = matches[i];
    const double err = reprojection_error(f, (double)m.x, (double)m.y, (double)m.z, (double)m.w);
    mask[i] = (fabs(err) < __builtin_inf() && !(fabs(err) > t)) ? 1 : 0; // optimize_result, :231-239
}

} // namespace cvhip

using namespace cvhip;

namespace {
bool dev_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
} // namespace

extern "C" int cvhip_ransac_score(cvhip_device *dev, const double *F, uint32_t H, const uint32_t *matches,
                                  uint32_t N, double t, uint32_t *out_count, double *out_err_sum)
{
    if (!dev || (!F && H) || (!matches && N) || !out_count || !out_err_sum)
        return fail(CVHIP_ERR_INVALID, "null argument");
    if (H == 0) return CVHIP_OK;
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    // the kernel reads matches as 16-byte vectors: a device pointer that is not 16-byte aligned is copied too
    const bool f_dev = dev_ptr(F),
               m_dev = N ? (dev_ptr(matches) && (reinterpret_cast<uintptr_t>(matches) & 15u) == 0) : true,
               c_dev = dev_ptr(out_count),
               e_dev = dev_ptr(out_err_sum);
    double *d_F = const_cast<double *>(F), *d_err = out_err_sum;
    uint32_t *d_m = const_cast<uint32_t *>(matches), *d_cnt = out_count;
    hipError_t e = hipSuccess;
    if (!f_dev) {
        e = hipMalloc(&d_F, (size_t)H * 9 * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(d_F, F, (size_t)H * 9 * sizeof(double), hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && !m_dev) {
        e = hipMalloc(&d_m, (size_t)N * 4 * sizeof(uint32_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_m, matches, (size_t)N * 4 * sizeof(uint32_t),
                               dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && !c_dev) e = hipMalloc(&d_cnt, (size_t)H * sizeof(uint32_t));
    if (e == hipSuccess && !e_dev) e = hipMalloc(&d_err, (size_t)H * sizeof(double));
    if (e == hipSuccess) {
        launch_ransac_score(d_F, H, d_m, N, t, d_cnt, d_err, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !c_dev)
        e = hipMemcpyAsync(out_count, d_cnt, (size_t)H * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && !e_dev)
        e = hipMemcpyAsync(out_err_sum, d_err, (size_t)H * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && !(c_dev && e_dev && f_dev && m_dev)) e = hipStreamSynchronize(s);
    if (!f_dev && d_F != F) (void)hipFree(d_F);
    if (!m_dev && d_m != matches) (void)hipFree(d_m);
    if (!c_dev && d_cnt != out_count) (void)hipFree(d_cnt);
    if (!e_dev && d_err != out_err_sum) (void)hipFree(d_err);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("ransac_score: ") + hipGetErrorString(e));
    return CVHIP_OK;
}

extern "C" int cvhip_ransac_affine(cvhip_device *dev, const uint32_t *matches, uint32_t N, uint64_t seed,
                                   double *out_F, uint32_t *out_inlier_count, uint8_t *out_inlier_mask)
{
    // constants of the affine model, fundamentalmatrix.rs:16-30
    constexpr uint32_t RANSAC_K = 1000000, CHECK_INTERVAL = 50000, RANSAC_N = 4, RANSAC_D = 10, EARLY_EXIT = 1000,
                       TOP_INLIERS = 5000;
    constexpr double RANSAC_T = 0.1;
    if (!dev || !matches || !out_F) return fail(CVHIP_ERR_INVALID, "null argument");
    if (N < RANSAC_D + RANSAC_N) return fail(CVHIP_ERR_NO_MODEL, "Not enough matches"); // :107-109
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    uint32_t *d_m = nullptr, *d_cnt = nullptr;
    double *d_F = nullptr, *d_err = nullptr;
    RansacBest *d_best = nullptr;
    uint8_t *d_mask = nullptr;
    hipError_t e = hipMalloc(&d_m, (size_t)N * 16);
    if (e == hipSuccess) e = hipMalloc(&d_F, (size_t)CHECK_INTERVAL * 9 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (size_t)CHECK_INTERVAL * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_err, (size_t)CHECK_INTERVAL * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_best, sizeof(RansacBest));
    if (e == hipSuccess) e = hipMalloc(&d_mask, N);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_best, 0, sizeof(RansacBest), s);
    RansacBest h_best;
    std::memset(&h_best, 0, sizeof(h_best));
    const uint4 *m4 = reinterpret_cast<const uint4 *>(d_m);
    for (uint32_t round = 0; e == hipSuccess && round < RANSAC_K / CHECK_INTERVAL; round++) {
        hipLaunchKernelGGL(ransac_generate_affine_kernel, dim3((CHECK_INTERVAL + 63) / 64), dim3(64), 0, s, m4,
                           std::min(N, TOP_INLIERS), RANSAC_T, (unsigned long long)seed, round, CHECK_INTERVAL, d_F);
        launch_ransac_score(d_F, CHECK_INTERVAL, d_m, N, RANSAC_T, d_cnt, d_err, s);
        hipLaunchKernelGGL(ransac_pick_best_kernel, dim3(1), dim3(1024), 0, s, d_F, d_cnt, d_err, CHECK_INTERVAL,
                           RANSAC_D + RANSAC_N, d_best);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (h_best.valid && h_best.matches_count > EARLY_EXIT) break; // :135-141
    }
    int rc = CVHIP_OK;
    if (e == hipSuccess && !h_best.valid) rc = fail(CVHIP_ERR_NO_MODEL, "No reliable matches found"); // :145
    if (e == hipSuccess && rc == CVHIP_OK) {
        std::memcpy(out_F, h_best.f, sizeof(h_best.f));
        hipLaunchKernelGGL(ransac_inlier_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_best, m4, N, RANSAC_T,
                           d_mask);
        e = hipGetLastError();
        std::vector<uint8_t> h_mask(N);
        if (e == hipSuccess) e = hipMemcpyAsync(h_mask.data(), d_mask, N, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < N; i++) cnt += h_mask[i];
            if (out_inlier_count) *out_inlier_count = cnt;
            if (out_inlier_mask) std::memcpy(out_inlier_mask, h_mask.data(), N);
        }
    }
    (void)hipFree(d_m);
    (void)hipFree(d_F);
    (void)hipFree(d_cnt);
    (void)hipFree(d_err);
    (void)hipFree(d_best);
    (void)hipFree(d_mask);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("ransac_affine: ") + hipGetErrorString(e));
    return rc;
}
