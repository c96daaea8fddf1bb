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
    const bool f_dev = dev_ptr(F), m_dev = N ? dev_ptr(matches) : true, c_dev = dev_ptr(out_count),
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
            e = hipMemcpyAsync(d_m, matches, (size_t)N * 4 * sizeof(uint32_t), hipMemcpyHostToDevice, s);
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
