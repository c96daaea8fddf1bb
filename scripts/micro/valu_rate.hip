// Micro-benchmark: per-CU issue rate of v_dot4_u32_u8 vs v_add_u32 vs ds_read_b64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    __shared__ unsigned long long lds[2048];
    unsigned a0 = threadIdx.x * 3 + seed, a1 = a0 * 5 + 1, a2 = a0 * 7 + 2, a3 = a0 * 11 + 3;
    unsigned b0 = seed ^ 0x01020304u, b1 = seed ^ 0x05060708u;
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i * 0x0101010101010101ull + seed;
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c0 = __builtin_amdgcn_udot4(a0, b0, c0, false);
                c1 = __builtin_amdgcn_udot4(a1, b1, c1, false);
                c2 = __builtin_amdgcn_udot4(a2, b0, c2, false);
                c3 = __builtin_amdgcn_udot4(a3, b1, c3, false);
                c4 = __builtin_amdgcn_udot4(a0, b1, c4, false);
                c5 = __builtin_amdgcn_udot4(a1, b0, c5, false);
                c6 = __builtin_amdgcn_udot4(a2, b1, c6, false);
                c7 = __builtin_amdgcn_udot4(a3, b0, c7, false);
            }
        } else if (OP == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c0 = (c0 + a0) ^ b0; c1 = (c1 + a1) ^ b1; c2 = (c2 + a2) ^ b0; c3 = (c3 + a3) ^ b1;
                c4 = (c4 + a0) ^ b1; c5 = (c5 + a1) ^ b0; c6 = (c6 + a2) ^ b1; c7 = (c7 + a3) ^ b0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned idx = (threadIdx.x + j * 64 + (c0 & 7)) & 2047;
                const unsigned long long v = lds[idx];
                c0 += (unsigned)v;
                c1 ^= (unsigned)(v >> 32);
            }
        }
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
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
    // wave-instructions per SIMD: blocks_per_cu waves per SIMD (4 waves/block, 4 SIMDs/CU)
    const double winst = (double)blocks_per_cu * iters * ops_per_iter;
    printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name,
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / winst);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("dot4", w, 64);
        run<1>("add+xor", w, 128);
        run<2>("ds_read64", w, 8);
    }
    return 0;
}
