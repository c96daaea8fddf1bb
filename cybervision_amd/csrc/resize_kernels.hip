// resize_kernels.hip — pyramid levels on the device.
//
// Replaces SourceImage::resize (zlogic/cybervision src/reconstruction.rs:146-162):
//     image::imageops::resize(&img, (w as f32 * scale) as u32, (h as f32 * scale) as u32, FilterType::Lanczos3)
// on a Luma8 image.  The `image` crate (0.25.10, Cargo.lock:475) is not vendored in the reference tree, so this
// restates its PUBLISHED algorithm (imageops/sample.rs): a vertical pass into an f32 image, then a horizontal pass,
// each output sample a normalised weighted sum over the source samples within `support * max(ratio, 1)` of its
// centre, all in f32, weights w((i - (centre - 0.5)) / sratio) with the Lanczos3 window sinc(x) sinc(x / 3),
// accumulated in source order, clamped to [0, 255] and rounded to nearest after the second pass only.  Equal
// dimensions are a plain copy.  The weights go through glibc's sinf (Rust's f32::sin on linux-gnu); everything else
// is one IEEE f32 operation per step in the crate's order (-ffp-contract=off), so the output equals the oracle's
// restatement byte for byte (tests/test_resize.py, MAX_DIFF = 0).  Nothing in the reference pins the crate's output.
// The weight tables are built on the host (glibc sinf, like the reference's host code) - one row of taps per output
// row / column - and the two passes are plain streaming kernels (coalesced along x).
#include "cvhip_internal.hpp"

#include <cmath>
#include <string>
#include <vector>

namespace cvhip {

struct ResampleTable {
    std::vector<uint32_t> left;  // first source index per output index
    std::vector<uint32_t> count; // taps per output index
    std::vector<float> weights;  // max_taps per output index, normalised
    uint32_t max_taps = 0;
};

static float sinc_f32(float t)
{
    const float a = t * 3.14159265358979323846f;
    return t == 0.0f ? 1.0f : std::sin(a) / a;
}
static float lanczos3_kernel(float x) { return std::fabs(x) < 3.0f ? sinc_f32(x) * sinc_f32(x / 3.0f) : 0.0f; }

// the per-output-sample part of vertical_sample / horizontal_sample
static ResampleTable build_table(uint32_t in_size, uint32_t out_size)
{
    ResampleTable t;
    const float ratio = (float)in_size / (float)out_size;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 3.0f * sratio;
    t.left.resize(out_size);
    t.count.resize(out_size);
    std::vector<std::vector<float>> rows(out_size);
    for (uint32_t o = 0; o < out_size; o++) {
        float centre = ((float)o + 0.5f) * ratio;
        long long left = (long long)std::floor(centre - src_support);
        left = left < 0 ? 0 : (left > (long long)in_size - 1 ? (long long)in_size - 1 : left);
        long long right = (long long)std::ceil(centre + src_support);
        right = right < left + 1 ? left + 1 : (right > (long long)in_size ? (long long)in_size : right);
        centre = centre - 0.5f;
        float sum = 0.0f;
        std::vector<float> &ws = rows[o];
        for (long long i = left; i < right; i++) {
            const float w = lanczos3_kernel(((float)i - centre) / sratio);
            ws.push_back(w);
            sum += w;
        }
        for (float &w : ws) w /= sum;
        t.left[o] = (uint32_t)left;
        t.count[o] = (uint32_t)ws.size();
        if (ws.size() > t.max_taps) t.max_taps = (uint32_t)ws.size();
    }
    t.weights.assign((size_t)out_size * t.max_taps, 0.0f);
    for (uint32_t o = 0; o < out_size; o++)
        for (size_t i = 0; i < rows[o].size(); i++) t.weights[(size_t)o * t.max_taps + i] = rows[o][i];
    return t;
}

// u8 [h][w] -> f32 [nh][w]
__global__ __launch_bounds__(256) void lanczos_vertical_kernel(const uint8_t *__restrict__ src, uint32_t w,
                                                                const uint32_t *__restrict__ left,
                                                                const uint32_t *__restrict__ count,
                                                                const float *__restrict__ weights, uint32_t max_taps,
                                                                uint32_t nh, float *__restrict__ tmp)
{
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (x >= w || oy >= nh) return;
    const uint32_t l = left[oy], n = count[oy];
    const float *ws = weights + (size_t)oy * max_taps;
    float t = 0.0f;
    for (uint32_t i = 0; i < n; i++) t += (float)src[(size_t)(l + i) * w + x] * ws[i];
    tmp[(size_t)oy * w + x] = t;
}

// f32 [nh][w] -> u8 [nh][nw]
__global__ __launch_bounds__(256) void lanczos_horizontal_kernel(const float *__restrict__ tmp, uint32_t w,
                                                                  const uint32_t *__restrict__ left,
                                                                  const uint32_t *__restrict__ count,
                                                                  const float *__restrict__ weights, uint32_t max_taps,
                                                                  uint32_t nw, uint32_t nh, uint8_t *__restrict__ dst)
{
    const uint32_t ox = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (ox >= nw || y >= nh) return;
    const uint32_t l = left[ox], n = count[ox];
    const float *ws = weights + (size_t)ox * max_taps;
    const float *row = tmp + (size_t)y * w + l;
    float t = 0.0f;
    for (uint32_t i = 0; i < n; i++) t += row[i] * ws[i];
    t = fminf(fmaxf(t, 0.0f), 255.0f);        // clamp(t, min, max)
    dst[(size_t)y * nw + ox] = (uint8_t)roundf(t); // FloatNearest: round half away from zero
}

} // namespace cvhip

using namespace cvhip;

namespace {
bool resident(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
} // namespace

extern "C" int cvhip_resize_lanczos3(cvhip_device *dev, const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst,
                                     uint32_t nw, uint32_t nh)
{
    if (!dev || !src || !dst) return fail(CVHIP_ERR_INVALID, "null argument");
    if (!w || !h || !nw || !nh) return fail(CVHIP_ERR_INVALID, "empty image");
    if (w > 65535 || h > 65535 || nw > 65535 || nh > 65535) return fail(CVHIP_ERR_UNSUPPORTED, "image dimension above 65535");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    const bool src_dev = resident(src), dst_dev = resident(dst);
    try {
        if (nw == w && nh == h) { // "if the new dimensions are the same as the old, make a copy instead of resampling"
            const hipMemcpyKind kind = src_dev ? (dst_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost)
                                               : (dst_dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost);
            CVHIP_TRY_HIP(hipMemcpyAsync(dst, src, (size_t)w * h, kind, s));
            if (!src_dev || !dst_dev) CVHIP_TRY_HIP(hipStreamSynchronize(s));
            return CVHIP_OK;
        }
        // the tables of this (size, size) pair: built and uploaded once per device handle
        const auto table_of = [&](uint32_t in_size, uint32_t out_size, const Device::ResizeTable *&out) -> hipError_t {
            for (const auto &t : dev->d.resize_tables)
                if (t.in_size == in_size && t.out_size == out_size) {
                    out = &t;
                    return hipSuccess;
                }
            const ResampleTable host = build_table(in_size, out_size);
            Device::ResizeTable t;
            t.in_size = in_size;
            t.out_size = out_size;
            t.max_taps = host.max_taps;
            hipError_t e = hipMalloc(&t.idx, (size_t)2 * out_size * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(&t.weights, host.weights.size() * sizeof(float));
            if (e == hipSuccess) e = hipMemcpyAsync(t.idx, host.left.data(), (size_t)out_size * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(t.idx + out_size, host.count.data(), (size_t)out_size * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(t.weights, host.weights.data(), host.weights.size() * sizeof(float), hipMemcpyHostToDevice, s);
            // (the host vectors go away with this scope: the uploads must have happened)
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                if (t.idx) (void)hipFree(t.idx);
                if (t.weights) (void)hipFree(t.weights);
                return e;
            }
            dev->d.resize_tables.push_back(t);
            out = &dev->d.resize_tables.back();
            return hipSuccess;
        };
        const Device::ResizeTable *tv = nullptr, *th = nullptr;
        hipError_t e = table_of(h, nh, tv);
        if (e == hipSuccess) {
            const size_t at = tv - dev->d.resize_tables.data(); // (the second lookup may grow the vector)
            e = table_of(w, nw, th);
            tv = &dev->d.resize_tables[at];
        }
        uint8_t *d_src = const_cast<uint8_t *>(src), *d_dst = dst;
        if (e == hipSuccess && !src_dev) {
            e = hipMalloc(&d_src, (size_t)w * h);
            if (e == hipSuccess) e = hipMemcpyAsync(d_src, src, (size_t)w * h, hipMemcpyHostToDevice, s);
        }
        if (e == hipSuccess && !dst_dev) e = hipMalloc(&d_dst, (size_t)nw * nh);
        if (e == hipSuccess && (size_t)w * nh > dev->d.resize_tmp_floats) {
            // (the plane is reused by later calls in stream order; a larger one is needed: the old one's users must be through)
            if (dev->d.resize_tmp) {
                e = hipStreamSynchronize(s);
                (void)hipFree(dev->d.resize_tmp);
                dev->d.resize_tmp = nullptr;
                dev->d.resize_tmp_floats = 0;
            }
            if (e == hipSuccess) e = hipMalloc(&dev->d.resize_tmp, (size_t)w * nh * sizeof(float));
            if (e == hipSuccess) dev->d.resize_tmp_floats = (size_t)w * nh;
        }
        if (e == hipSuccess) {
            float *d_tmp = dev->d.resize_tmp;
            hipLaunchKernelGGL(lanczos_vertical_kernel, dim3((w + 255) / 256, nh), dim3(256), 0, s, d_src, w, tv->idx, tv->idx + nh, tv->weights,
                               tv->max_taps, nh, d_tmp);
            hipLaunchKernelGGL(lanczos_horizontal_kernel, dim3((nw + 255) / 256, nh), dim3(256), 0, s, d_tmp, w, th->idx, th->idx + nw,
                               th->weights, th->max_taps, nw, nh, d_dst);
            e = hipGetLastError();
        }
        if (e == hipSuccess && !dst_dev) e = hipMemcpyAsync(dst, d_dst, (size_t)nw * nh, hipMemcpyDeviceToHost, s);
        // device -> device: in stream order, no host synchronisation; host buffers are complete (and free again) on return
        if (e == hipSuccess && (!src_dev || !dst_dev)) e = hipStreamSynchronize(s);
        if (!src_dev && d_src) (void)hipFree(d_src);
        if (!dst_dev && d_dst) (void)hipFree(d_dst);
        if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("resize_lanczos3: ") + hipGetErrorString(e));
        return CVHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(CVHIP_ERR_NOMEM, "cvhip_resize_lanczos3: out of host memory");
    }
}
