// Two facts the matrix-product filter rests on, checked with exact integer data:
//  1. v_mfma_i32_16x16x64_i8 with A: lane l = row l & 15, K block l >> 4 (16 bytes), B: lane l = column l & 15, same K
//     block, C: column l & 15, rows 4 (l >> 4) + i - and the result does not depend on the K order inside a block as long
//     as both operands use the same one;
//  2. a ds_read_b128 at an arbitrary BYTE address returns the sixteen bytes at that address (gfx950, unaligned access
//     mode), and what it costs against an aligned one.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_i8 scripts/micro/mfma_i8_unaligned_lds.hip && /tmp/mfma_i8
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_check(const signed char *A /*[16][64]*/, const signed char *B /*[64][16], row = k*/, int *C /*[16][16]*/, int shift)
{
    __shared__ __attribute__((aligned(16))) signed char la[16 * 80 + 64], lb[16 * 80 + 64]; // rows of 64 bytes at pitch 80, base shifted by `shift`
    const int l = threadIdx.x;
    for (int i = l; i < 16 * 64; i += 64) {
        la[shift + (i >> 6) * 80 + (i & 63)] = A[i];                       // A row-major: [row][k]
        lb[shift + (i & 15) * 80 + (i >> 4)] = B[i];                       // B stored per COLUMN: [col][k]
    }
    __syncthreads();
    i32x4 a, b, c = {0, 0, 0, 0};
    __builtin_memcpy(&a, la + shift + (l & 15) * 80 + 16 * (l >> 4), 16);
    __builtin_memcpy(&b, lb + shift + (l & 15) * 80 + 16 * (l >> 4), 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) C[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}

__global__ void lds_speed(unsigned *out, int off, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = (unsigned char)(i * 7);
    __syncthreads();
    uint4 acc = make_uint4(0, 0, 0, 0);
    // off 0: 16-byte aligned; 1 .. 3: every lane at its own odd byte address (stride 17); 4, 8, 12: dword aligned, stride 16;
    // 20: stride 20 (dword aligned, every lane another alignment class)
    int p = off >= 1 && off <= 3 ? (threadIdx.x & 63) * 17 + off : (off == 20 ? (threadIdx.x & 63) * 20 : (threadIdx.x & 63) * 16 + off);
    for (int it = 0; it < iters; it++) {
        uint4 v;
        __builtin_memcpy(&v, buf + (p & 8191), 16);
        acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
        p += 1040;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    std::vector<signed char> A(16 * 64), B(64 * 16);
    srand(5);
    for (auto &v : A) v = (signed char)(rand() % 255 - 127);
    for (auto &v : B) v = (signed char)(rand() % 255 - 127);
    signed char *dA, *dB;
    int *dC;
    hipMalloc(&dA, A.size());
    hipMalloc(&dB, B.size());
    hipMalloc(&dC, 256 * sizeof(int));
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int shift = 0; shift < 8; shift++) {
        hipLaunchKernelGGL(mfma_check, dim3(1), dim3(64), 0, 0, dA, dB, dC, shift);
        std::vector<int> C(256);
        hipMemcpy(C.data(), dC, 256 * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                int want = 0;
                for (int k = 0; k < 64; k++) want += (int)A[r * 64 + k] * (int)B[k * 16 + c];
                bad += want != C[r * 16 + c];
            }
        printf("shift %d: %d of 256 outputs wrong\n", shift, bad);
        bad_total += bad;
    }
    unsigned *dout;
    hipMalloc(&dout, 256 * 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int offs[] = {0, 1, 2, 3, 4, 8, 12, 20};
    for (int off : offs) {
        hipLaunchKernelGGL(lds_speed, dim3(1024), dim3(256), 0, 0, dout, off, 2000);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lds_speed, dim3(1024), dim3(256), 0, 0, dout, off, 2000);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("ds_read_b128, offset class %d: %.3f ms for 1024 x 256 threads x 2000 reads\n", off, ms);
    }
    return bad_total != 0;
}
