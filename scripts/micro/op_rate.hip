// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU operations of the box kernel's
// walk, eight independent chains per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 op_rate.hip -o op_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned c[8], a = threadIdx.x * 2654435761u + seed, b = seed ^ 0x01020304u;
    float f[8], fa = (float)(threadIdx.x + 1) * 1.0001f, fb = 0.999f;
    for (int j = 0; j < 8; j++) {
        c[j] = a + j;
        f[j] = fa + j;
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (OP == 0) c[j] = __builtin_amdgcn_udot4(a, b, c[j], false);
                if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a));
                if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fb));
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(fb), "v"(fa));
                if (OP == 4) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c[j]));
                if (OP == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[j]));
                if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c[j]) : "v"(a));
                if (OP == 7) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
                if (OP == 8) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[j]) : "v"(c[j]));
                if (OP == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(fb), "v"(fa));
                if (OP == 10) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(c[j]) : "v"(a));
                if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[j]) : "v"(a));
                if (OP == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a));
                if (OP == 13) asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(c[j]) : "v"(c[(j + 1) & 7]));
                if (OP == 14) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[j]) : "v"(c[j]));
                if (OP == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fb));
            }
        }
    }
    unsigned s = 0;
    for (int j = 0; j < 8; j++) s += c[j] + (unsigned)f[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int waves_per_simd)
{
    unsigned *out;
    const int blocks = 256 * waves_per_simd, iters = 2000; // one 256-thread block = one wave per SIMD of a CU
    hipMalloc(&out, (size_t)blocks * 256 * 4);
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
    const double instr_per_wave = (double)iters * 32.0, cycles = ms * 1e-3 * 2.4e9;
    printf("%-22s waves/SIMD=%d  %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           cycles / (instr_per_wave * waves_per_simd));
    hipFree(out);
}

int main()
{
    for (int w : {1, 4}) {
        run<0>("v_dot4_u32_u8", w);
        run<1>("v_add_u32", w);
        run<12>("v_sub_u32", w);
        run<2>("v_add_f32", w);
        run<15>("v_mul_f32", w);
        run<3>("v_fma_f32", w);
        run<4>("v_add_u32 dpp row_shr", w);
        run<5>("v_add_f32 dpp row_shr", w);
        run<13>("v_mov_b32 dpp bcast15", w);
        run<6>("v_mul_u32_u24", w);
        run<7>("v_mad_i32_i24", w);
        run<8>("v_cvt_f32_u32", w);
        run<14>("v_cvt_f32_ubyte0", w);
        run<9>("v_max3_f32", w);
        run<10>("v_alignbyte_b32", w);
        run<11>("v_cndmask_b32", w);
    }
    return 0;
}
