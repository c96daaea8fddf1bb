// Micro-benchmark: issue cost of the cross-lane pieces of the box filter on gfx950:
// DPP prefix sums (6 dependent v_add_u32_dpp), ds_bpermute_b32, plain ds_read_b32, v_mul_u32_u24/v_cvt mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned scan(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    __shared__ unsigned lds[4096];
    const unsigned lane = threadIdx.x & 63;
    unsigned c[8];
    for (int j = 0; j < 8; j++) c[j] = threadIdx.x * (3 + j) + seed;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 0x01010101u + seed;
    __syncthreads();
    const int ihi = (int)min(lane + 5u, 63u) * 4, ilo = (int)(lane >= 6 ? lane - 6 : 0) * 4;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { // 5 independent scans: 30 DPP adds
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = scan(c[j]);
        } else if (OP == 1) { // 10 bpermutes
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const unsigned a = (unsigned)__builtin_amdgcn_ds_bpermute(ihi, (int)c[j]);
                const unsigned b = (unsigned)__builtin_amdgcn_ds_bpermute(ilo, (int)c[j]);
                c[j] = a - b + 1;
            }
        } else if (OP == 2) { // 10 ds_read_b32, consecutive lanes
#pragma unroll
            for (int j = 0; j < 10; j++) c[j & 7] += lds[(lane + j * 64 + (i & 63)) & 4095];
        } else if (OP == 3) { // 5 x (scan + 2 bpermute + sub)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const unsigned p = scan(c[j]);
                c[j] = (unsigned)__builtin_amdgcn_ds_bpermute(ihi, (int)p) - (unsigned)__builtin_amdgcn_ds_bpermute(ilo, (int)p);
            }
        } else if (OP == 4) { // 20 v_mul_u32_u24 + 20 cvt
#pragma unroll
            for (int j = 0; j < 5; j++) {
                c[j] = __umul24(c[j], 121u) - __umul24(c[j + 1], c[7]);
                c[j] = (unsigned)(float)(int)c[j] + __umul24(c[j], 3u);
                c[j] = __umul24(c[j], 5u) ^ (unsigned)(float)(int)(c[j] >> 3);
            }
        }
        asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
    }
    unsigned s = 0;
    for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int blocks_per_cu, double ops_per_iter)
{
    unsigned *out;
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 1);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)blocks_per_cu * iters;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.1f cycles per iteration per SIMD-wave, %.2f per instruction (2.4 GHz)\n", name,
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / groups, ms * 1e-3 * 2.4e9 / groups / ops_per_iter);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("5 scans (30 dpp)", w, 30);
        run<1>("10 bpermute (+10 valu)", w, 10);
        run<2>("10 ds_read_b32", w, 10);
        run<3>("5x(scan+2bperm+sub)", w, 45);
        run<4>("mul24/cvt mix (45)", w, 45);
    }
    return 0;
}
