// Micro-benchmark for VERDICT r4 item 6: the box walk's 5-plane group with ONE image column per lane (today's form: 64
// columns, 53 searched pixels per wave) against TWO columns per lane (128 columns, 118 pixels per wave: the 6-step DPP prefix
// sum then scans pair sums, the 11-column halo is paid once per 118 pixels, dot4 and the per-pixel epilogue double).
// Both kernels run the walk's loop alone - per step: the target lines' 20 bytes from the transposed LDS copy, the candidate
// statistics, column products (v_dot4_u32_u8 on pre-shifted copies of the lane's column), wave prefix sum, window sums via
// ds_bpermute, N = 121 S12 - s1 s2, the margin fma, max3 and the (never taken) hit branch - on LDS contents that do not
// matter, at the LDS footprint and launch bound each form would have in the kernel.  Reported: ns per wave-step and
// ps per (pixel, displacement plane) with the chip full.
//   hipcc --offload-arch=gfx950 -O3 -o box_group box_group.hip && ./box_group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

constexpr int STEPS = 24;

// ---- one column per lane (box_body.inc's lean group) ------------------------------------------------------------------
constexpr int COLS1 = 128, LDS1 = 4 * COLS1 * 16 + 4 * COLS1 * 4 + 8 * COLS1 * 8; // 22.5 KB: lines, tails, statistics
__global__ __launch_bounds__(256, 6) void group_one(const uint32_t *__restrict__ seed, float *__restrict__ out, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS1];
    for (int i = threadIdx.x; i < LDS1 / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = seed[(i + blockIdx.x) & 4095];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t a[4][4];
    for (int q = 0; q < 4; q++)
        for (int k = 0; k < 4; k++) a[q][k] = seed[(lane + 7 * q + k) & 4095] & 0x3F3F3F3Fu;
    const uint32_t s1 = seed[lane] & 0x7FFF;
    float limk = __uint_as_float(0x7F000000u), acc = 0.0f; // (a limit nothing reaches: the hit branch is never taken)
    const int idx_lo = (int)(lane >= 11u ? lane - 11u : 0u) * 4;
    const uint8_t *const bB16 = lds + (w * COLS1 + lane) * 16u, *const bB4 = lds + 4 * COLS1 * 16 + (w * COLS1 + lane) * 4u;
    const uint8_t *const bIS = lds + 4 * COLS1 * 16 + 4 * COLS1 * 4 + lane * 8u;
    for (int r = 0; r < reps; r++) {
        for (int step = 0; step < STEPS; step++) {
            const uint4 r4 = *reinterpret_cast<const uint4 *>(bB16 + step * 16);
            const uint32_t raw[5] = {r4.x, r4.y, r4.z, r4.w, *reinterpret_cast<const uint32_t *>(bB4 + step * 4)};
            const uint8_t *pIS = bIS + step * 8;
            int num[5];
            float sd[5], mg[5];
#pragma unroll
            for (int s = 0; s < 5; s++) {
                const int sh = s & 3, o = s >> 2;
                const uint2 is2 = *reinterpret_cast<const uint2 *>(pIS + s * (COLS1 * 8));
                uint32_t c = __builtin_amdgcn_udot4(a[sh][0], raw[o], 0u, false);
                c = __builtin_amdgcn_udot4(a[sh][1], raw[o + 1], c, false);
                c = __builtin_amdgcn_udot4(a[sh][2], raw[o + 2], c, false);
                if (sh >= 2) c = __builtin_amdgcn_udot4(a[sh][3], raw[o + 3 < 5 ? o + 3 : 4], c, false);
                const uint32_t pre = wave_prefix_sum(c);
                const uint32_t s12 = pre - (uint32_t)__builtin_amdgcn_ds_bpermute(idx_lo, (int)pre);
                num[s] = __mul24((int)s12, 121) + __mul24((int)s1, (int)is2.x);
                sd[s] = __uint_as_float(is2.y & 0x3FFFFFFFu);
            }
#pragma unroll
            for (int s = 0; s < 5; s++) mg[s] = __builtin_fmaf(-limk, sd[s], (float)num[s]);
            float margin = mg[0];
#pragma unroll
            for (int s = 1; s < 5; s++) margin = fmaxf(margin, mg[s]);
            if (margin >= 0.0f) { // never
                acc += margin;
                limk *= 1.5f;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + limk;
}

// ---- two columns per lane ---------------------------------------------------------------------------------------------
// lane l owns columns 2l and 2l + 1 and the two searched pixels whose windows END in them.  With C0, C1 the lane's column
// products, Q = wave prefix sum of C0 + C1 (= P(2l + 1)) and R = Q - C1 (= P(2l)):
//   S12(window ending at 2l + 1) = Q(l) - R(l - 5),   S12(window ending at 2l) = R(l) - Q(l - 6)
constexpr int COLS2 = 192, LDS2 = 4 * COLS2 * 16 + 4 * COLS2 * 4 + 8 * COLS2 * 8; // 33.8 KB
// WAVES: the occupancy the kernel would have with the real kernel's registers (~130: three waves per SIMD) - enforced here with LDS
// ballast (the loop alone needs 81 registers and would run at five)
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void group_two(const uint32_t *__restrict__ seed, float *__restrict__ out, int reps)
{
    constexpr int BALLAST = WAVES == 3 ? 52 * 1024 : (WAVES == 4 ? 39 * 1024 : LDS2); // 160 KB / CU: 3, 4 or 5 workgroups
    __shared__ __attribute__((aligned(16))) uint8_t lds[BALLAST > LDS2 ? BALLAST : LDS2];
    for (int i = threadIdx.x; i < LDS2 / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = seed[(i + blockIdx.x) & 4095];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t a0[4][4], a1[4][4];
    for (int q = 0; q < 4; q++)
        for (int k = 0; k < 4; k++) {
            a0[q][k] = seed[(lane + 7 * q + k) & 4095] & 0x3F3F3F3Fu;
            a1[q][k] = seed[(lane + 5 * q + k + 99) & 4095] & 0x3F3F3F3Fu;
        }
    const uint32_t s1e = seed[lane] & 0x7FFF, s1o = seed[lane + 64] & 0x7FFF;
    float limk_e = __uint_as_float(0x7F000000u), limk_o = limk_e, acc = 0.0f;
    const int idx5 = (int)(lane >= 5u ? lane - 5u : 0u) * 4, idx6 = (int)(lane >= 6u ? lane - 6u : 0u) * 4;
    // (even and odd columns in separate halves of a copy, so that the lane's two 16-byte reads are unit-stride across the wave:
    // at step t the lane's columns are 2l + t and 2l + t + 1, i.e. entries l + (t >> 1) [+ (t & 1)] of the two halves)
    const uint8_t *const bEven = lds + (w * COLS2 + lane) * 16u, *const bOdd = bEven + (COLS2 / 2) * 16u;
    const uint8_t *const bB4 = lds + 4 * COLS2 * 16 + (w * COLS2 + 2 * lane) * 4u;
    const uint8_t *const bIS = lds + 4 * COLS2 * 16 + 4 * COLS2 * 4 + 2 * lane * 8u;
    for (int r = 0; r < reps; r++) {
        for (int step = 0; step < STEPS; step++) {
            // (a step moves the target columns by one: the two columns' lines are two consecutive 16-byte entries)
            const int half = step >> 1, odd = step & 1;
            const uint4 ra = *reinterpret_cast<const uint4 *>((odd ? bOdd : bEven) + half * 16);
            const uint4 rb = *reinterpret_cast<const uint4 *>((odd ? bEven + 16 : bOdd) + half * 16);
            const uint32_t *tl = reinterpret_cast<const uint32_t *>(bB4 + step * 4); // (two dwords, 4-byte aligned: ds_read2_b32)
            const uint32_t raw0[5] = {ra.x, ra.y, ra.z, ra.w, tl[0]}, raw1[5] = {rb.x, rb.y, rb.z, rb.w, tl[1]};
            const uint8_t *pIS = bIS + step * 8;
            int ne[5], no[5];
            float sde[5], sdo[5], mge[5], mgo[5];
#pragma unroll
            for (int s = 0; s < 5; s++) {
                const int sh = s & 3, o = s >> 2;
                // both pixels' {-s2, sd2}: two cells, 8-byte aligned at any step (ds_read2_b64; a 16-byte read is misaligned at every odd step)
                const uint2 ise = *reinterpret_cast<const uint2 *>(pIS + s * (COLS2 * 8)), iso = *reinterpret_cast<const uint2 *>(pIS + s * (COLS2 * 8) + 8);
                const uint4 is4 = make_uint4(ise.x, ise.y, iso.x, iso.y);
                uint32_t c1 = __builtin_amdgcn_udot4(a1[sh][0], raw1[o], 0u, false);
                c1 = __builtin_amdgcn_udot4(a1[sh][1], raw1[o + 1], c1, false);
                c1 = __builtin_amdgcn_udot4(a1[sh][2], raw1[o + 2], c1, false);
                if (sh >= 2) c1 = __builtin_amdgcn_udot4(a1[sh][3], raw1[o + 3 < 5 ? o + 3 : 4], c1, false);
                uint32_t d = __builtin_amdgcn_udot4(a0[sh][0], raw0[o], c1, false); // C0 on top of C1: the pair sum
                d = __builtin_amdgcn_udot4(a0[sh][1], raw0[o + 1], d, false);
                d = __builtin_amdgcn_udot4(a0[sh][2], raw0[o + 2], d, false);
                if (sh >= 2) d = __builtin_amdgcn_udot4(a0[sh][3], raw0[o + 3 < 5 ? o + 3 : 4], d, false);
                const uint32_t Q = wave_prefix_sum(d), R = Q - c1;
                const uint32_t s_odd = Q - (uint32_t)__builtin_amdgcn_ds_bpermute(idx5, (int)R);
                const uint32_t s_even = R - (uint32_t)__builtin_amdgcn_ds_bpermute(idx6, (int)Q);
                ne[s] = __mul24((int)s_even, 121) + __mul24((int)s1e, (int)is4.x);
                no[s] = __mul24((int)s_odd, 121) + __mul24((int)s1o, (int)is4.z);
                sde[s] = __uint_as_float(is4.y & 0x3FFFFFFFu);
                sdo[s] = __uint_as_float(is4.w & 0x3FFFFFFFu);
            }
#pragma unroll
            for (int s = 0; s < 5; s++) {
                mge[s] = __builtin_fmaf(-limk_e, sde[s], (float)ne[s]);
                mgo[s] = __builtin_fmaf(-limk_o, sdo[s], (float)no[s]);
            }
            float margin = fmaxf(mge[0], mgo[0]);
#pragma unroll
            for (int s = 1; s < 5; s++) margin = fmaxf(margin, fmaxf(mge[s], mgo[s]));
            if (margin >= 0.0f) { // never
                acc += margin;
                limk_e *= 1.5f;
                limk_o *= 1.25f;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + limk_e + limk_o;
}

template <typename K> static double time_ms(K kernel, int blocks, const uint32_t *seed, float *out, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(seed, out, 2);
    hipEventRecord(e0);
    kernel<<<blocks, 256>>>(seed, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint32_t *seed;
    float *out;
    const int blocks = 256 * 8 * 4; // 32 workgroups per CU: several rounds at any occupancy
    hipMalloc(&seed, 4096 * 4 + 1024);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    uint32_t h[4096 + 256];
    for (int i = 0; i < 4096 + 256; i++) h[i] = 0x01010101u * (uint32_t)(i % 61) + (uint32_t)i * 2654435761u;
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const int reps = 40;
    const double wave_steps = (double)blocks * 4 * reps * STEPS;
    struct { const char *name; double ms; int px; } rows[4];
    rows[0] = {"one column per lane, 6 waves/SIMD (today)", time_ms(group_one, blocks, seed, out, reps), 53};
    rows[1] = {"two columns per lane, 4 waves/SIMD", time_ms(group_two<4>, blocks, seed, out, reps), 118};
    rows[2] = {"two columns per lane, 3 waves/SIMD", time_ms(group_two<3>, blocks, seed, out, reps), 118};
    rows[3] = {"two columns per lane, 5 waves/SIMD", time_ms(group_two<5>, blocks, seed, out, reps), 118};
    for (auto &r : rows)
        printf("%-46s %8.3f ms  %7.2f ns per wave-step on the full chip  %7.2f ps per (pixel, plane)\n", r.name, r.ms, r.ms * 1e6 / wave_steps,
               r.ms * 1e9 / (wave_steps * r.px * 5));
    printf("ratio two/one per (pixel, plane): 4 waves %.3f, 3 waves %.3f, 5 waves %.3f  (kill criterion: not below 0.85)\n",
           rows[1].ms / 118 / (rows[0].ms / 53), rows[2].ms / 118 / (rows[0].ms / 53), rows[3].ms / 118 / (rows[0].ms / 53));
    return 0;
}
