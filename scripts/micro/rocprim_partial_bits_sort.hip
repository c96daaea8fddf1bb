// Round 3's ORB batch first grouped the globally sorted corners by image with rocprim::radix_sort_keys over bits 28..32
// of the tagged values (image << 28 | corner) and got garbage in part of the list (gpurun_out/orbdbg2.log: 928 bad ranks
// of 3922, values like 0xe631a6f0 that are no tagged index at all); it was replaced by a full-width sort of (image << 28 |
// rank) keys and blamed on the library.  This reproducer makes the same call on the same kind of data, on its own:
//   A. temporary storage queried for THIS call (bits 28..32), its own allocation, distinct in / out buffers;
//   B. the temporary storage of ANOTHER sort (64-bit pairs, as the Harris sort before it) handed over with this call's size
//      - what the debug log's call text shows the failing build did (`radix_sort_keys(d_tmp, tmp2, d_idx_s1, ...`);
//   C. as A, in place (keys_output == keys_input).
// and compares each with std::stable_sort by the 4-bit key on the host.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/rp_sort scripts/micro/rocprim_partial_bits_sort.hip && /tmp/rp_sort
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main()
{
    int bad_total = 0;
    const unsigned sizes[] = {3922u, 12000u, 47000u, 120000u};
    for (unsigned n : sizes) {
        for (unsigned images : {1u, 3u, 12u, 16u}) {
            std::vector<unsigned> h(n), want, got(n);
            srand(n + images);
            for (unsigned i = 0; i < n; i++) h[i] = ((unsigned)(rand() % images) << 28) | (i & 0x0FFFFFFFu); // descending-Harris order = arbitrary image order
            want = h;
            std::stable_sort(want.begin(), want.end(), [](unsigned a, unsigned b) { return (a >> 28) < (b >> 28); });
            unsigned *d_in, *d_out;
            unsigned long long *d_k64, *d_k64o;
            CK(hipMalloc(&d_in, n * 4));
            CK(hipMalloc(&d_out, n * 4));
            CK(hipMalloc(&d_k64, n * 8));
            CK(hipMalloc(&d_k64o, n * 8));
            size_t t_this = 0, t_other = 0;
            CK(rocprim::radix_sort_keys(nullptr, t_this, d_in, d_out, (size_t)n, 28u, 32u, 0));
            CK(rocprim::radix_sort_pairs_desc(nullptr, t_other, d_k64, d_k64o, d_in, d_out, (size_t)n, 0u, 64u, 0));
            void *tmp_this, *tmp_other;
            CK(hipMalloc(&tmp_this, t_this));
            CK(hipMalloc(&tmp_other, t_other));
            auto run = [&](const char *what, void *tmp, size_t bytes, bool in_place) -> int {
                if (hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
                (void)hipMemset(d_out, 0xEE, n * 4);
                const hipError_t e = rocprim::radix_sort_keys(tmp, bytes, d_in, in_place ? d_in : d_out, (size_t)n, 28u, 32u, 0);
                if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                    printf("  %s: error %s\n", what, hipGetErrorString(e));
                    (void)hipGetLastError();
                    return -1;
                }
                (void)hipMemcpy(got.data(), in_place ? d_in : d_out, n * 4, hipMemcpyDeviceToHost);
                int bad = 0, unsorted = 0, shown = 0;
                for (unsigned i = 0; i < n; i++) {
                    bad += got[i] != want[i];
                    if (i && (got[i] >> 28) < (got[i - 1] >> 28)) unsorted++;
                    if (got[i] != want[i] && shown < 3 && n == 3922u && images == 3u) {
                        printf("    [%u] got %08x want %08x\n", i, got[i], want[i]);
                        shown++;
                    }
                }
                std::vector<unsigned> a1 = got, a2 = want;
                std::sort(a1.begin(), a1.end());
                std::sort(a2.begin(), a2.end());
                printf("  (same multiset: %d, positions out of key order: %d)", (int)(a1 == a2), unsorted);
                printf("  n %6u images %2u %-34s temp %8zu B (this call asks %8zu, the pairs sort %8zu): %d wrong\n", n, images, what, bytes, t_this,
                       t_other, bad);
                return bad;
            };
            const int a = run("A own temp, out of place", tmp_this, t_this, false);
            const int b = run("B other sort's temp, this size", tmp_other, t_this, false);
            const int c = run("C own temp, in place", tmp_this, t_this, true);
            bad_total += (a != 0) + (b != 0) + (c != 0);
            (void)hipFree(d_in);
            (void)hipFree(d_out);
            (void)hipFree(d_k64);
            (void)hipFree(d_k64o);
            (void)hipFree(tmp_this);
            (void)hipFree(tmp_other);
        }
    }
    printf("cases with a wrong result: %d\n", bad_total);
    return 0;
}
