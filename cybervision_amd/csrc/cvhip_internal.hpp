// cvhip_internal.hpp — shared declarations of libcvhip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/cvhip.h"

namespace cvhip {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define CVHIP_TRY_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::cvhip::fail(CVHIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define CVHIP_TRY(expr)      \
    do {                     \
        int _rc = (expr);    \
        if (_rc != CVHIP_OK) \
            return _rc;      \
    } while (0)

// ---- constants of the reference (src/correlation/mod.rs:15-31) -----------------------------
constexpr int KERNEL_SIZE = 5;
constexpr int KERNEL_WIDTH = 11;
constexpr int KERNEL_POINT_COUNT = 121;
constexpr int NEIGHBOR_DISTANCE = 10;
constexpr int CROSS_CHECK_SEARCH_AREA = 4;

// Level grids are two planes: the match plane, one u32 per level pixel = x | y << 16 in LEVEL coordinates (or
// CELL_NONE) - all that the search range, the cross-checks and the dense consumers read - and the score plane (f32),
// which only complete() reads.
constexpr uint32_t CELL_NONE = 0xFFFFFFFFu;
constexpr uint32_t RANGE_NONE = 0xFFFFFFFFu; // packed corridor range: start | end << 16, or None
constexpr size_t IMG_PAD = 64;               // bytes readable past the end of every image buffer

// Per-pass parameters handed to the kernels by value.
struct CorrParams {
    double F[9];  // fundamental matrix of this direction, row-major (transposed for Reverse)
    double min_range, extend_range;
    float scale;
    float min_stdev, threshold;
    int corridor_size;
    uint32_t w1, h1, w2, h2; // level dims of the searched (1) and target (2) image
    uint32_t gw, gh;         // full-resolution grid dims of this direction
    uint32_t pw, ph, pk;     // previous level's compact grid dims and its k (scale = 2^-k)
    uint32_t k;              // this level's k
    uint32_t row0, row1;     // rows of the searched image handled by this launch
    int first_pass;
    // 1: the scores this pass writes can reach the caller and must be the reference's bits - the forward pass of the
    // full-resolution level, or every pass under cvhip_ctx_set_exact_scores.  0: only the match POSITIONS of this pass
    // are ever read (search range and cross-check of the next passes; the reference overwrites the cells of a coarser
    // level, mod.rs:311-316, and drops the reverse grid, mod.rs:208-215), so a pixel whose filter band holds ONE
    // contender clearly above the threshold is settled without the exact 121-term evaluation.
    int need_scores;
    // stepped box launches (search3_box_*<.., STEP = true, ..>): dynamic LDS plan - dwords staged per target line
    // (odd, so that consecutive lines fall into different banks) and rows of candidate statistics
    uint32_t box_pd, box_sh, box_wide; // box_wide: the 128-line plan (shallow lines), else the 100-line plan
    // Affine F (first two columns zero): F*p = (F02, F12, .) for every finite pixel, so the epipolar line's direction
    // is one constant, evaluated once on the host with the reference's expression (mod.rs:397-408; IEEE division,
    // the same bits as on the device).  affine = 1: the |l.x| > |l.y| branch, aff_c = -F12/F02, aff_div = F02;
    // 2: the other one, aff_c = -F02/F12, aff_div = F12;  0: not affine (or not finite) - the generic per-pixel path.
    int affine;
    double aff_c, aff_div;
    // search_range_kernel: 1 = every pixel of this pass has a finite epipolar line with the direction constants above
    // (p.affine != 0, the divisor and the largest possible |F2 . p| far from the f64 range limits) - the kernel then
    // evaluates no line at all (the interval only needs the line's axis and the corridor's end)
    int range_quick;
    // Profiling ablations (env CVHIP_DEBUG, applied to the full-resolution level only; results are then wrong on
    // purpose): 1 = skip the whole-corridor kernel, 2 = skip the filter kernels, 4 = the box kernel declines every
    // workgroup, 8 / 16 = the box kernel skips its walk / its exact phase, 32 = box statistics in counters 1 and 2, 64 = the walk never enters its hit branch,
    // 256 = the box kernel stops after per-pixel setup, 512 = it skips staging (and its exact phase).
    int debug;
};

// ---- kernel launchers (corr_kernels.hip) ----------------------------------------------------
// tile work list shared by the kernels of one search pass (search version 3)
struct WorkList {
    uint32_t *count;
    uint32_t *items;
};
void launch_window_stats_pair(const uint8_t *img_a, uint32_t wa, uint32_t ha, uint2 *istats_a,
                              const uint8_t *img_b, uint32_t wb, uint32_t hb, uint2 *istats_b,
                              uint32_t row0, uint32_t row1, float min_stdev, uint32_t *zero_words, hipStream_t s,
                              uint32_t lds_ballast = 0);
// One direction's search pass of a level: everything its kernels need.  The kernels of the two directions of a level
// are launched together (jobs[0], jobs[1] -> blockIdx.z); per-pass callers launch one job.
struct SearchJob {
    CorrParams p;
    const uint8_t *img1, *img2; // searched / target level image
    const uint2 *stats1, *stats2; // their statistics words
    const uint32_t *prev;         // this direction's previous-level match plane (search range)
    uint32_t *range;              // search interval per searched pixel
    unsigned long long *contenders;
    uint32_t *out;                // this level's match plane
    float *out_score;             // ... and score plane
    unsigned long long *counters; // device counters or nullptr
    WorkList declined, whole;
    // first pass, one workgroup per (tile, stripe): search2_split_words(...) 8-byte words of scratch, or nullptr
    unsigned long long *split;
};
// scratch of the split first pass for a level of w x rows pixels with `stripes` stripes, in 8-byte words
size_t search2_split_words(uint32_t w, uint32_t rows, uint32_t stripes);
void launch_search_range(const SearchJob *jobs, int n, int mode, hipStream_t s);
void launch_search(const CorrParams &p, const uint8_t *img1, const uint8_t *img2, const uint2 *stats1,
                   const uint2 *stats2, const uint32_t *range, uint32_t *out, float *out_score,
                   unsigned long long *cand_counter, hipStream_t s);
void launch_search2_filter(const SearchJob *jobs, int n, hipStream_t s);
void launch_search3_fallback(const SearchJob *jobs, int n, bool skip_exact, hipStream_t s);
// (side / fork / join: the stepped instantiations launch one kernel per direction - with a side stream the second one goes
// there, between the two events, so that the first launch's tail is filled)
bool launch_search3_box(const SearchJob *jobs, int n, bool stepped_lines, bool transposed, int form, hipStream_t s,
                        hipStream_t side = nullptr, hipEvent_t fork = nullptr, hipEvent_t join = nullptr, int fallback_skip_exact = -1);
size_t search3_worklist_capacity(uint32_t max_w, uint32_t max_h);
void launch_cross_check(uint32_t *own, const uint32_t *other, uint32_t ow, uint32_t oh, uint32_t rw, uint32_t rh,
                        uint32_t row0, uint32_t row1, hipStream_t s);
void launch_cross_check_pair(uint32_t *fwd, uint32_t *rev, uint32_t fw, uint32_t fh, uint32_t rw, uint32_t rh, uint32_t f_row0,
                             uint32_t f_row1, uint32_t r_row0, uint32_t r_row1, hipStream_t s, uint32_t *zero_words = nullptr);
// scores == nullptr: the level's scores were not computed (CorrParams::need_scores): NaN everywhere
void launch_expand_grid(const uint32_t *cells, const float *scores, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                        int32_t *out_xy, float *out_corr, hipStream_t s, uint32_t gy0 = 0, uint32_t gy1 = 0xFFFFFFFFu, bool packed = false);
void launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t s);
// tracks of the affine dense consumer; block_counts must hold ceil(gw*gh/256) u32, total is one u32
void launch_triangulate_affine(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                               uint32_t *block_counts, uint32_t *total, double *out_points3d, uint32_t *out_p2,
                               unsigned long long cap, hipStream_t s);

// single-block exclusive scan of n u32 in place, total to *total
void launch_scan_u32(uint32_t *data, uint32_t n, uint32_t *total, hipStream_t s);
// Triangulation::extend_tracks on the forward grid (track_kernels.hip)
void launch_extend_tracks_match(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                                const int2 *track_p1, unsigned long long n_tracks, uint32_t radius, int2 *out_p2,
                                uint8_t *removed, uint32_t *oob, hipStream_t s);
void launch_extend_tracks_new(const uint32_t *cells, uint32_t lw, uint32_t lh, uint32_t k, uint32_t gw, uint32_t gh,
                              const uint8_t *removed, uint32_t *block_counts, uint32_t *total, uint32_t *out_new_p1,
                              uint32_t *out_new_p2, unsigned long long cap, hipStream_t s);

#ifdef __HIPCC__
// the match stored for full-resolution cell (gx, gy), if any: level cell (gx >> k, gy >> k) when both are multiples
// of 2^k (the scatter of mod.rs:311-316), scaled back by 2^k (mod.rs:459-462)
__device__ __forceinline__ bool full_res_match(const uint32_t *__restrict__ cells, uint32_t lw, uint32_t lh, uint32_t k,
                                               uint32_t gx, uint32_t gy, uint32_t &mx, uint32_t &my)
{
    const uint32_t mask = (1u << k) - 1u;
    if ((gx & mask) || (gy & mask)) return false;
    const uint32_t lx = gx >> k, ly = gy >> k;
    if (lx >= lw || ly >= lh) return false;
    const uint32_t c = cells[(size_t)ly * lw + lx];
    if (c == CELL_NONE) return false;
    mx = (c & 0xFFFFu) << k;
    my = (c >> 16) << k;
    return true;
}
#endif

// ---- handles --------------------------------------------------------------------------------
// The device buffers of one dense-correlation context.  The reference allocates them per image pair
// (GpuContext::new -> prepare_device, gpu/mod.rs:125-163) and frees them on drop; here a destroyed context parks its
// set on the device handle and the next context of the same dimensions takes it over (a pipeline correlates many
// pairs of equally sized images: hipMalloc / hipFree of ~1 GB per pair cost more than a 2048^2 correlation itself).
struct CtxBuffers {
    uint32_t w1 = 0, h1 = 0, w2 = 0, h2 = 0;
    uint32_t *cells[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float *scores[2] = {nullptr, nullptr};
    uint8_t *img[2] = {nullptr, nullptr};
    uint2 *istats[2] = {nullptr, nullptr};
    uint32_t *range = nullptr, *range_rev = nullptr;
    unsigned long long *contenders = nullptr, *contenders_rev = nullptr;
    uint32_t *work = nullptr;
    unsigned long long *d_cand = nullptr;
};

// Scratch of the sparse-stage entry points (ORB, matcher, RANSAC): ONE grow-only device allocation per handle,
// handed out by bumping an offset and recycled by the next call.  An extraction used to make ~25 hipMalloc /
// hipFree pairs (each free synchronises the device): 12 extractions of config 5 spent 15 ms on 2.5 ms of kernels.
// Ordering: every user enqueues on the handle's stream (in order), so a later call may reuse the bytes at once.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0, wanted = 0; // wanted: what the current call asked for in total, overflow included
    std::vector<void *> overflow;         // allocations of a call that did not fit (freed when it ends; the arena then grows)
    int depth = 0;
};

struct Device {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    std::string name;
    int low_power = 0;
    std::vector<CtxBuffers> parked; // at most PARK_LIMIT sets, most recently used last
    Arena arena;
    // page-locked host staging of the sparse-stage entry points (grow-only): a pageable hipMemcpy is staged by the
    // runtime in small pieces and costs ~50 us however few bytes it moves - an ORB extraction made nine of them
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    void *orb_pattern = nullptr; // the BRIEF pattern in device memory, uploaded once per handle
    // cvhip_resize_lanczos3: the resampling tables of every (source size, output size) met so far, in device memory - a
    // pipeline resizes equally sized images to the same few scales over and over (reconstruction.rs:421-422, 567-568) - and the
    // intermediate f32 plane (grow-only); freed with the handle
    struct ResizeTable {
        uint32_t in_size = 0, out_size = 0, max_taps = 0;
        uint32_t *idx = nullptr;  // left[out_size], count[out_size]
        float *weights = nullptr; // out_size x max_taps
    };
    std::vector<ResizeTable> resize_tables;
    float *resize_tmp = nullptr;
    size_t resize_tmp_floats = 0;
    double orb_guard = 1e-9;     // cvhip_orb_set_orientation_guard
    int ransac_in_order = 0;     // cvhip_ransac_set_in_order (test hook): batches are scored in order, behind their events, without polling
    int ransac_count_mfma = 0;   // cvhip_ransac_set_count_mfma: the counting screen's head as f32 matrix products (ransac_count_mfma_kernel) - exact, measured slower: off
    int ransac_lm_pipeline = 2;  // cvhip_ransac_set_lm_pipeline (test hook): validate_f's LM as 2 = two passes on refilled lanes (default), 1 = two passes, a root per thread, 0 = the scalar loop in one kernel
    int ransac_pencil = CVHIP_PENCIL_THIN_SVD; // cvhip_ransac_set_pencil: the 7-point pencil's basis (default: the reference's)
    // complete() into HOST memory (GpuContext::complete_process, gpu/mod.rs:210-216): two device staging sets that the
    // full-resolution grid is expanded into, and a copy stream of its own, so that the 12 B/px transfer of one pair can
    // run under the search of the next (cvhip_ctx_set_async_readback) and no call allocates.  Grow-only, per handle.
    struct Readback {
        int32_t *xy[2] = {nullptr, nullptr};
        float *corr[2] = {nullptr, nullptr};
        size_t cap_px = 0;
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr, done[2] = {nullptr, nullptr};
        hipEvent_t expanded = nullptr; // behind the last band's expansion on the copy stream (complete_grid)
        bool pending[2] = {false, false};
        int next = 0;
    } rb;
    // Page-locked ring for HOST level images (cvhip_correlate_images / _level with host pointers): the caller's buffer is
    // copied into the ring on the calling thread(s), the H2D transfer runs on the copy stream (rb.stream) under whatever
    // the main stream is doing, and the call returns without synchronising - the caller's buffer is free on return.
    struct UploadRing {
        hipStream_t stream = nullptr; // uploads have a stream of their own (H2D and D2H are separate DMA engines): the next pair's
                                      // level images do not queue behind the last pair's readback
        uint8_t *base = nullptr;
        size_t cap = 0, head = 0;
        struct Chunk {
            size_t begin, end;
            hipEvent_t done;
        };
        std::vector<Chunk> busy;       // transfers that may still be reading their part of the ring (oldest first)
        std::vector<hipEvent_t> spare; // events to reuse
    } up;
    // Side streams of the handle, shared by whoever needs one (the RANSAC generators: all four; the statistics-ahead mode:
    // the first; the stepped box launches' second direction: the second) and created on first use.  Shared on purpose
    // rather than one per purpose: streams beyond the process' hardware queues are mapped onto queues already in use (in
    // round 3, with a stream of its own for the statistics, config 5's RANSAC stage ran 22 -> 27 ms in a process that had
    // used both; in round 4 four generator streams beat two with GPU_MAX_HW_QUEUES at its default of 4 as with 8)
    // (the stepped box launches' second stream at the LOWEST priority - so that the first direction ends early and its fallback
    // kernel runs under the second direction's walk - was tried in round 5: the dispatch order follows the priority, but the
    // low-priority walk then starves beside the statistics stream: geometry sweep 6.5 .. 8.0 -> 8.2 .. 10.1 ms)
    hipStream_t aux[4] = {};
    hipEvent_t orb_ev[3] = {nullptr, nullptr, nullptr}; // cvhip_orb_extract_batch's fork / join events
    hipEvent_t box_ev[2] = {nullptr, nullptr};          // the stepped box launches' fork / join (launch_passes)
    // (levels from 1024^2: 4096^2 pair at 3 / 30 / 90 degrees 7.73 / 7.78 / 7.47 -> 7.67 / 7.74 / 7.37 ms; below, the two events cost more)
    size_t box_fork_min_px = 500000;
    struct OrbLanes { // ... and what it learned about running three image chains at once on this handle (orb_kernels.hip)
        size_t shape = 0;
        double best_ms = 0.0;
        uint32_t samples = 0; // batches timed so far (three lanes, two, one)
        bool decided = false;
        uint32_t use_lanes = 3;
    } orb_lanes;
    // the RANSAC loops' generator streams and round events (created on first use: creating two streams and seven events
    // per find_ransac call cost 1.5 ms of every ~9 ms call)
    struct RansacQueues {
        hipEvent_t ready[8] = {}, scored[8] = {}, uploaded = nullptr, started = nullptr; // (per hypothesis buffer in flight)
    } rq;
    // cvhip_ctx_set_stats_ahead: the side stream the window statistics of the finer levels run on while the coarse levels'
    // (launch-latency bound) search chain occupies the main stream, one "done" event per level, one fence (created on
    // first use)
    struct StatsAhead {
        hipEvent_t done[16] = {};
        hipEvent_t fence = nullptr;
        // Self-check.  Two streams only overlap if the driver gave them different hardware queues; on one queue the
        // throttled statistics kernel would stand IN LINE with everything else, at 3/8 of its normal occupancy - worse than
        // not using the mode at all.  Until a verdict exists, a pyramid run records when the full-resolution statistics
        // START on the side stream and when the main stream ARRIVES at that level; the next run reads the two (if they have
        // completed): statistics that did not start well before the arrival did not overlap - the mode switches itself off.
        hipEvent_t probe_side = nullptr, probe_main = nullptr;
        bool probe_recorded = false;
        int verdict = 0; // 0 = not known yet, 1 = overlapping, -1 = switched off
    } sa;
    // RCCL communicators created on this handle (cvhip_rccl_create) enqueue on its stream: while any is alive,
    // cvhip_device_destroy only marks the handle and the last cvhip_rccl_destroy frees it
    int comm_refs = 0;
    bool destroy_pending = false;
};
} // namespace cvhip
struct cvhip_ctx;
namespace cvhip {
int flush_level_calls(cvhip_ctx *ctx); // cvhip_api.hip: run what cvhip_ctx_set_fuse_level_calls has deferred
inline hipError_t aux_stream(Device &d, int i, hipStream_t *out)
{
    hipError_t e = hipSuccess;
    if (!d.aux[i]) e = hipStreamCreateWithFlags(&d.aux[i], hipStreamNonBlocking);
    *out = d.aux[i];
    return e;
}
// -> at least `bytes` of page-locked host memory owned by the handle (nullptr: out of memory)
inline void *pinned_scratch(Device &d, size_t bytes)
{
    if (bytes <= d.pinned_cap) return d.pinned;
    if (d.pinned) (void)hipHostFree(d.pinned);
    d.pinned = nullptr;
    d.pinned_cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&d.pinned, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        d.pinned = nullptr;
        return nullptr;
    }
    d.pinned_cap = want;
    return d.pinned;
}
constexpr size_t PARK_LIMIT = 2;

// RAII view of the arena for one entry-point call: alloc() until the call returns, everything is released together
// (error paths included).  Alignment 256 B.
struct DevAllocs {
    Arena &a;
    explicit DevAllocs(Device &d) : a(d.arena)
    {
        if (a.depth++ == 0) a.used = a.wanted = 0;
    }
    DevAllocs(const DevAllocs &) = delete;
    DevAllocs &operator=(const DevAllocs &) = delete;
    ~DevAllocs()
    {
        if (--a.depth > 0) return;
        for (void *p : a.overflow) (void)hipFree(p);
        a.overflow.clear();
        if (a.wanted > a.cap) { // grow once, with some slack, so that the next call of this size fits
            if (a.base) (void)hipFree(a.base);
            a.base = nullptr;
            a.cap = 0;
            const size_t want = a.wanted + a.wanted / 4;
            void *p = nullptr;
            if (hipMalloc(&p, want) == hipSuccess) {
                a.base = static_cast<char *>(p);
                a.cap = want;
            } else {
                (void)hipGetLastError(); // stay on per-call allocations
            }
        }
    }
    hipError_t alloc_bytes(void **out, size_t bytes)
    {
        const size_t need = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
        a.wanted += need;
        if (a.used + need <= a.cap) {
            *out = a.base + a.used;
            a.used += need;
            return hipSuccess;
        }
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) a.overflow.push_back(p);
        *out = p;
        return e;
    }
    template <typename T> hipError_t alloc(T **out, size_t count)
    {
        void *p = nullptr;
        const hipError_t e = alloc_bytes(&p, count * sizeof(T));
        *out = static_cast<T *>(p);
        return e;
    }
};

struct DirState {
    uint32_t *cells[2] = {nullptr, nullptr}; // ping-pong compact match planes (the previous level's is read by the search range)
    float *scores = nullptr;                 // score plane of the most recent level (nothing reads an older one)
    bool scores_valid = false;               // ... which holds the reference's scores (CorrParams::need_scores)
    int cur = 0;        // index of the grid holding the most recent level
    bool valid = false; // a level has been computed
    uint32_t lw = 0, lh = 0, k = 0;
    uint32_t gw = 0, gh = 0; // full-res dims of this direction's grid
};

} // namespace cvhip

struct cvhip_device {
    cvhip::Device d;
};

namespace cvhip {
void device_free(cvhip_device *dev); // cvhip_api.hip: what cvhip_device_destroy does once nothing references the handle
}

struct cvhip_ctx {
    cvhip_device *dev = nullptr;
    uint32_t w1 = 0, h1 = 0, w2 = 0, h2 = 0;
    int projection = 0;
    double F[9] = {0};
    // CorrelationParameters::for_projection (mod.rs:111-143)
    float min_stdev = 1.0f, threshold = 0.6f;
    int corridor_size = 2;
    double min_range = 2.5, extend_range = 1.0;

    cvhip::DirState dir[2];
    uint8_t *img[2] = {nullptr, nullptr}; // level image staging (padded), [0]=searched [1]=target of the call
    const uint8_t *cur_img[2] = {nullptr, nullptr}; // the images the current call works on: img[] or the caller's own
    bool borrow_inputs = false;                     // cvhip_ctx_set_borrow_inputs
    bool stats_ahead = false;                       // cvhip_ctx_set_stats_ahead
    bool stats_ahead_fenced = false;                // this pyramid run's side-stream work is ordered behind everything earlier
    // per level pixel {window sum | VALID << 31, f32 bits of stdev}: avg = sum / 121 is derived where it is needed
    uint2 *istats[2] = {nullptr, nullptr};
    // 1 = per-candidate exact kernel, 2 = integer filter per candidate + exact re-evaluation,
    // 3 = displacement-plane box filter (falls back to 2 per workgroup) + exact re-evaluation
    // The reverse cross-check of the full-resolution level (mod.rs:240) filters a grid nothing reads any more: no finer
    // level follows scale 1 and complete() returns the forward grid (mod.rs:208-215).  It is therefore deferred and
    // only runs if somebody does ask for the reverse grid (cvhip_complete_dir(.., 1, ..), cvhip_ctx_level_grid(.., 1, ..)).
    bool rev_cross_check_pending = false;
    // cvhip_ctx_set_fuse_level_calls: the caller issues the four backend calls of a level in
    // PointCorrelations::correlate_images' order (mod.rs:217-245) - forward, reverse with the SAME two images exchanged,
    // cross-check forward, cross-check reverse - and the library executes them as cvhip_correlate_level would: the
    // forward call takes the images in (statistics of both), the reverse call launches both search passes together, the
    // second cross-check call launches both filters.  `calls` is what has been taken in but not executed yet
    // (cvhip::flush_level_calls runs it whenever another entry point comes first).
    bool fuse_level_calls = false;
    struct LevelCalls {
        enum { NONE = 0, FWD_TAKEN, SEARCHED, CROSS_FWD_TAKEN, HELD, HELD_CROSS_FWD }; // HELD*: result bands (below)
        int stage = NONE;
        int k = -1, first_pass = 0;
        float scale = 0.0f;
        const uint8_t *img1 = nullptr, *img2 = nullptr; // the forward call's images (identity only: never dereferenced later)
        uint32_t w1 = 0, h1 = 0, w2 = 0, h2 = 0;
        bool stats_ahead = false;
        hipEvent_t stats_done = nullptr;
    } calls;
    // cvhip_ctx_set_result_bands: the last level (scale 1) is searched and filtered in row bands, band b's forward filter
    // right behind the search of band b + 1, each with an event - complete() into HOST memory then expands and copies out band
    // b while the later bands are still being searched, instead of 201 MB of PCIe time behind the whole level.
    uint32_t result_bands = 1;             // bands asked for (1 = off)
    uint32_t live_bands = 0;               // bands of the current last-level result (0: not banded)
    uint32_t band_rows[17] = {};           // forward-grid rows [band_rows[b], band_rows[b + 1])
    hipEvent_t band_done[16] = {};         // band b of the forward grid is final
    bool bands_crossed = false;            // the last level's cross-checks went out with its bands: level_cross has nothing to do
    bool staged_from_pageable = false; // the last stage_images copied straight from the caller's pageable memory (no ring)
    hipEvent_t level_read[16] = {};    // per level: the last kernels that read the staged images have been enqueued before it
    hipEvent_t pool_ready = nullptr;   // the image pool has been cleared (context creation, on the context's stream): the copy
    bool pool_waited = false;          // stream waits for it before its first upload into the pool
    bool async_readback = false; // cvhip_ctx_set_async_readback
    bool exact_scores = false; // cvhip_ctx_set_exact_scores: every pass writes the reference's scores (tests)
    int search_version = 3; // (5: version 3 with the rectified box launches on the matrix pipe, search4_mfma_kernel - measured slower;
                            //  6: version 3 with the rectified box launches on two columns per lane, search3_box2_kernel)
    int range_mode = 0; // search_range_kernel: 0 = integer box sums + chain where needed, 1 = chain only, 2 / 3 = test hooks
    bool force_box = false; // launch the box kernel whatever the geometry (it declines per workgroup)
    // per-direction scratch of a search pass (the two passes of a level are independent: one launch, blockIdx.z picks one)
    uint32_t *range = nullptr, *range_rev = nullptr;
    unsigned long long *contenders = nullptr, *contenders_rev = nullptr; // filter -> exact kernel hand-off, one word per searched pixel
    uint32_t *work = nullptr;                 // tile work lists: per direction {declined n, whole n, declined scan, whole scan}, then two item arrays
    size_t work_cap = 0;                      // items per list
    size_t max_px = 0;

    uint32_t shard_num = 0, shard_den = 1;
    cvhip_allgather_fn gather = nullptr;
    void *gather_user = nullptr;
    bool gather_on_stream = false; // the gather enqueues on the device handle's stream (the library's RCCL path)
    // Independent-band mode (cvhip_ctx_set_row_band): per level k, the row intervals [lo, hi) of the
    // forward / reverse search passes (sf, sr) and cross-checks (cf, cr) this context has to compute so
    // that its band of the final forward grid is exact without any exchange.
    struct BandPlan {
        uint32_t sf[2], sr[2], cf[2], cr[2], st[2];
    };
    bool band_mode = false;
    int band_steps = 0;
    BandPlan band[16];

    int time_kernels = 0, count_candidates = 0;
    unsigned long long *d_cand = nullptr;
    // kernel classes timed with HIP events when time_kernels is set
    enum { K_STATS = 0, K_RANGE, K_SEARCH, K_EXACT, K_CROSS, K_EXPAND, K_FILTER, K_COUNT };
    struct TimedLaunch {
        hipEvent_t e0, e1;
        int cls;
    };
    std::vector<TimedLaunch> events;
    size_t events_used = 0;
    uint32_t prof_launches[K_COUNT] = {0};
    double prof_ms[K_COUNT] = {0};
};
