// Does a matrix instruction run BESIDE vector instructions on a gfx950 SIMD?  Four kernels, same grid (one workgroup of 512
// threads = 2 waves per SIMD on every CU), timed with events:
//   valu    every wave issues N dependent-free v_fma_f32
//   mfma    every wave issues N/8 matrix instructions of the chosen kind
//   both    every wave issues both streams interleaved (one matrix instruction per 8 v_fma)
//   split   even waves issue only the vector stream, odd waves only the matrix stream (so a SIMD holds one of each)
//   split-v / split-m   the same with the other half of the waves idle: what each half takes alone
// (four independent accumulators per wave: the matrix stream is bound by the pipe, not by its own dependency chain)
// If the pipes overlap, t(split) approaches max(t(split-v), t(split-m)); if the matrix instruction takes the vector issue
// slots, it approaches their sum.   hipcc --offload-arch=gfx950 -O3 -o mfma_coexec mfma_coexec.hip && ./mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

template <int KIND> __device__ __forceinline__ void mfma_step(v16f &acc, v4f &acc4, float a32, float b32, v8h ah, v8h bh, v8s as, v8s bs)
{
    if constexpr (KIND == 0) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc4, 0, 0, 0);       // f32, 8 passes
    if constexpr (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);          // f16, 32x32x16
    if constexpr (KIND == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bs, acc, 0, 0, 0);         // bf16, 32x32x16
}

template <int KIND, int MODE> __global__ __launch_bounds__(512) void probe(float *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool do_valu = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 5) && (wave & 1) == 0);
    const bool do_mfma = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 4) && (wave & 1) == 1);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (float)threadIdx.x * 1e-3f + (float)i;
    const float m = 1.0000001f, c = 1e-7f;
    v16f acc[4] = {};
    v4f acc4[4] = {};
    v8h ah, bh;
    v8s as, bs;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        ah[i] = (_Float16)(0.001f * (float)(threadIdx.x & 7));
        bh[i] = (_Float16)(0.002f * (float)i);
        as[i] = (short)(0x3C00 + i);
        bs[i] = (short)(0x3C00 + (threadIdx.x & 3));
    }
    const float a32 = (float)(threadIdx.x & 15) * 1e-3f, b32 = 1e-3f;
    for (int it = 0; it < iters; it++) {
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], m, c);
                if (do_mfma && MODE == 2) mfma_step<KIND>(acc[r & 3], acc4[r & 3], a32, b32, ah, bh, as, bs);
            }
        }
        if (do_mfma && MODE != 2) {
#pragma unroll
            for (int r = 0; r < 8; r++) mfma_step<KIND>(acc[r & 3], acc4[r & 3], a32, b32, ah, bh, as, bs);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int MODE> float run(float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, MODE>), dim3(blocks), dim3(512), 0, 0, out, 16); // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, MODE>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND> void kind(const char *name, float *out, int blocks, int iters)
{
    const float tv = run<KIND, 0>(out, blocks, iters), tm = run<KIND, 1>(out, blocks, iters), tb = run<KIND, 2>(out, blocks, iters),
                ts = run<KIND, 3>(out, blocks, iters), tsm = run<KIND, 4>(out, blocks, iters), tsv = run<KIND, 5>(out, blocks, iters);
    // per SIMD: 2 waves; valu: each wave 64 v_fma per iteration; mfma: each wave 8 per iteration
    printf("%-26s valu %.3f  mfma %.3f  both-in-one-wave %.3f (sum %.3f)  |  split-v %.3f  split-m %.3f  split %.3f ms (sum %.3f, max %.3f)\n",
           name, tv, tm, tb, tv + tm, tsv, tsm, ts, tsv + tsm, tsv > tsm ? tsv : tsm);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount; // one 512-thread workgroup per CU: two waves per SIMD
    float *out;
    hipMalloc(&out, (size_t)blocks * 512 * sizeof(float));
    const int iters = 20000;
    printf("%s, %d CUs, %d iterations\n", prop.gcnArchName, blocks, iters);
    kind<0>("v_mfma_f32_16x16x4_f32", out, blocks, iters);
    kind<1>("v_mfma_f32_32x32x16_f16", out, blocks, iters);
    kind<2>("v_mfma_f32_32x32x16_bf16", out, blocks, iters);
    hipFree(out);
    return 0;
}
