// cvhip_api.hip — C ABI of libcvhip.so: device + dense-correlation context.
// Host-side orchestration only; the kernels are in corr_kernels.hip.  See include/cvhip.h for
// the reference interface each entry point replaces.
#include "cvhip_internal.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

namespace cvhip {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

// Is `p` a device pointer usable on the current GPU?  Unregistered host memory makes
// hipPointerGetAttributes fail, which is not an error for us.
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // clear sticky error state
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static int copy_in(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    const hipMemcpyKind kind = is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    CVHIP_TRY_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
    return CVHIP_OK;
}

// scale must be exactly 2^-k; returns k or -1.
static int scale_to_k(float scale)
{
    if (!(scale > 0.0f) || scale > 1.0f) return -1;
    int e = 0;
    const float m = std::frexp(scale, &e); // scale = m * 2^e, m in [0.5, 1)
    if (m != 0.5f) return -1;
    const int k = 1 - e;
    return (k >= 0 && k <= 15) ? k : -1;
}

static int set_device(const cvhip_device *dev)
{
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    return CVHIP_OK;
}

static void free_buffer_set(CtxBuffers &b)
{
    for (int d = 0; d < 2; d++) {
        for (int i = 0; i < 2; i++)
            if (b.cells[d][i]) (void)hipFree(b.cells[d][i]);
        if (b.scores[d]) (void)hipFree(b.scores[d]);
        if (b.img[d]) (void)hipFree(b.img[d]);
        if (b.istats[d]) (void)hipFree(b.istats[d]);
    }
    if (b.range) (void)hipFree(b.range);
    if (b.range_rev) (void)hipFree(b.range_rev);
    if (b.contenders) (void)hipFree(b.contenders);
    if (b.contenders_rev) (void)hipFree(b.contenders_rev);
    if (b.work) (void)hipFree(b.work);
    if (b.d_cand) (void)hipFree(b.d_cand);
    b = CtxBuffers{};
}

// Detach the context's device buffers: parked on the device handle when `park` (and they are complete), freed otherwise.
static void release_ctx_buffers(cvhip_ctx *c, bool park)
{
    CtxBuffers b;
    b.w1 = c->w1;
    b.h1 = c->h1;
    b.w2 = c->w2;
    b.h2 = c->h2;
    bool complete = true;
    for (int d = 0; d < 2; d++) {
        for (int i = 0; i < 2; i++) {
            b.cells[d][i] = c->dir[d].cells[i];
            complete = complete && b.cells[d][i];
            c->dir[d].cells[i] = nullptr;
        }
        b.scores[d] = c->dir[d].scores;
        complete = complete && b.scores[d];
        c->dir[d].scores = nullptr;
        b.img[d] = c->img[d];
        b.istats[d] = c->istats[d];
        complete = complete && b.img[d] && b.istats[d];
        c->img[d] = nullptr;
        c->istats[d] = nullptr;
    }
    b.range = c->range;
    b.range_rev = c->range_rev;
    b.contenders = c->contenders;
    b.contenders_rev = c->contenders_rev;
    b.work = c->work;
    b.d_cand = c->d_cand;
    complete = complete && b.range && b.range_rev && b.contenders && b.contenders_rev && b.work && b.d_cand;
    c->range = c->range_rev = nullptr;
    c->contenders = c->contenders_rev = nullptr;
    c->work = nullptr;
    c->d_cand = nullptr;
    if (park && complete) {
        auto &parked = c->dev->d.parked;
        parked.push_back(b);
        while (parked.size() > PARK_LIMIT) {
            free_buffer_set(parked.front());
            parked.erase(parked.begin());
        }
    } else {
        free_buffer_set(b);
    }
}

static void free_ctx_buffers(cvhip_ctx *c, bool park = false)
{
    release_ctx_buffers(c, park);
    for (auto &ev : c->events) {
        (void)hipEventDestroy(ev.e0);
        (void)hipEventDestroy(ev.e1);
    }
    c->events.clear();
}

// Bracket a launch with HIP events on the context's stream when kernel timing is on.
template <typename F> static int timed(cvhip_ctx *c, int cls, F &&launch, hipStream_t s = nullptr)
{
    if (!s) s = c->dev->d.stream;
    // time_kernels: 0 = off, 1 = every class, 2 = the search class only
    if (!c->time_kernels || (c->time_kernels == 2 && cls != cvhip_ctx::K_SEARCH && cls != cvhip_ctx::K_FILTER)) {
        launch();
        return CVHIP_OK;
    }
    if (c->events_used == c->events.size()) {
        cvhip_ctx::TimedLaunch t;
        CVHIP_TRY_HIP(hipEventCreate(&t.e0));
        CVHIP_TRY_HIP(hipEventCreate(&t.e1));
        t.cls = cls;
        c->events.push_back(t);
    }
    cvhip_ctx::TimedLaunch &t = c->events[c->events_used++];
    t.cls = cls;
    CVHIP_TRY_HIP(hipEventRecord(t.e0, s));
    launch();
    CVHIP_TRY_HIP(hipEventRecord(t.e1, s));
    return CVHIP_OK;
}

static int resolve_events(cvhip_ctx *c)
{
    for (size_t i = 0; i < c->events_used; i++) {
        float ms = 0.0f;
        CVHIP_TRY_HIP(hipEventElapsedTime(&ms, c->events[i].e0, c->events[i].e1));
        c->prof_ms[c->events[i].cls] += (double)ms;
        c->prof_launches[c->events[i].cls]++;
    }
    c->events_used = 0;
    return CVHIP_OK;
}

// Rows of an lh-row level image owned by this context's shard: equal chunks of ceil(lh/den).
static void shard_rows(const cvhip_ctx *c, uint32_t lh, uint32_t *row0, uint32_t *row1)
{
    const uint32_t rpr = (lh + c->shard_den - 1) / c->shard_den;
    const uint32_t r0 = std::min(lh, c->shard_num * rpr);
    *row0 = r0;
    *row1 = std::min(lh, r0 + rpr);
}

// Level images of a call: copied into the context's padded buffers - or, when the caller has vouched for the
// padding of its device buffers (cvhip_ctx_set_borrow_inputs), used where they are.
// The statistics words of level k live at their own offset of the per-image pool (level j has at most max_px / 4^j
// pixels): the statistics of a finer level can then be computed while a coarser level is still being searched
// (cvhip_ctx_set_stats_ahead).
static size_t stats_level_offset(size_t max_px, int k)
{
    size_t off = 0;
    for (int j = 0; j < k; j++) off += ((max_px >> (2 * j)) + 63) & ~(size_t)63;
    return off;
}
static size_t stats_pool_elems(size_t max_px) { return stats_level_offset(max_px, 16) + 64; }

// The staged image of level k lives at its own offset of the per-image pool, like the statistics: a host image of level
// k - 1 can then be uploaded (copy stream) while level k is still being searched.
static size_t img_level_offset(size_t max_px, int k)
{
    size_t off = 0;
    for (int j = 0; j < k; j++) off += ((max_px >> (2 * j)) + IMG_PAD + 255) & ~(size_t)255;
    return off;
}
static size_t img_pool_bytes(size_t max_px) { return img_level_offset(max_px, 16) + IMG_PAD + 256; }

// The handle's copy stream (uploads of host level images, readback of host-destination grids) and its events.
static int copy_stream_reserve(Device &d)
{
    auto &rb = d.rb;
    if (rb.stream) return CVHIP_OK;
    CVHIP_TRY_HIP(hipStreamCreateWithFlags(&rb.stream, hipStreamNonBlocking));
    CVHIP_TRY_HIP(hipStreamCreateWithFlags(&d.up.stream, hipStreamNonBlocking));
    CVHIP_TRY_HIP(hipEventCreateWithFlags(&rb.ready, hipEventDisableTiming));
    CVHIP_TRY_HIP(hipEventCreateWithFlags(&rb.expanded, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) CVHIP_TRY_HIP(hipEventCreateWithFlags(&rb.done[i], hipEventDisableTiming));
    return CVHIP_OK;
}

// `bytes` of the handle's page-locked upload ring, free of any transfer that may still read them; *done is the event the
// caller records behind its transfer.  nullptr: no ring to be had (out of page-locked memory) - the caller copies straight
// from the pageable source and synchronises.
static uint8_t *upload_ring_take(Device &d, size_t bytes, size_t want_cap, hipEvent_t *done)
{
    Device::UploadRing &up = d.up;
    bytes = (bytes + 255) & ~(size_t)255;
    if (up.cap < bytes || (!up.base && want_cap)) {
        for (auto &c : up.busy) {
            (void)hipEventSynchronize(c.done);
            up.spare.push_back(c.done);
        }
        up.busy.clear();
        if (up.base) (void)hipHostFree(up.base);
        up.base = nullptr;
        up.cap = up.head = 0;
        const size_t cap = std::max(std::max(bytes, want_cap), (size_t)32 << 20);
        void *p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        up.base = static_cast<uint8_t *>(p);
        up.cap = cap;
    }
    if (up.head + bytes > up.cap) up.head = 0;
    const size_t begin = up.head, end = up.head + bytes;
    for (size_t i = 0; i < up.busy.size();) {
        if (up.busy[i].begin < end && begin < up.busy[i].end) {
            (void)hipEventSynchronize(up.busy[i].done); // (normally long done: the ring holds two pairs' worth)
            up.spare.push_back(up.busy[i].done);
            up.busy.erase(up.busy.begin() + (long)i);
        } else {
            i++;
        }
    }
    hipEvent_t ev = nullptr;
    if (!up.spare.empty()) {
        ev = up.spare.back();
        up.spare.pop_back();
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    up.busy.push_back({begin, end, ev});
    up.head = end;
    *done = ev;
    return up.base + begin;
}

// (a 16.8 MB level image is ~1.7 ms of one core's memcpy: four threads take their quarter each)
static void host_copy(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    constexpr size_t PER_THREAD = (size_t)2 << 20;
    const size_t parts = std::min<size_t>(4, bytes / PER_THREAD);
    if (parts < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::thread th[3];
    const size_t chunk = ((bytes / parts) + 4095) & ~(size_t)4095;
    size_t started = 1;
    for (; started < parts; started++) {
        const size_t off = started * chunk, len = std::min(chunk, bytes - std::min(bytes, off));
        try {
            th[started - 1] = std::thread([=] { if (off < bytes) std::memcpy(dst + off, src + off, len); });
        } catch (...) { // no thread to be had: this one copies the rest itself (nothing may be thrown across the C ABI)
            break;
        }
    }
    std::memcpy(dst, src, std::min(chunk, bytes));
    if (started < parts && started * chunk < bytes) std::memcpy(dst + started * chunk, src + started * chunk, bytes - started * chunk);
    for (size_t i = 1; i < started; i++) th[i - 1].join();
}

// Level images of a call (transfer_in_images, gpu/mod.rs:274) -> c->cur_img[]:
//  * a device image the caller has vouched for (cvhip_ctx_set_borrow_inputs) is used where it is;
//  * another device image is copied into the level's area of the pool on the context's stream;
//  * a HOST image goes through the handle's page-locked ring: copied out of the caller's buffer here (the buffer is free
//    when this returns), transferred on the copy stream behind the last readers of the level's area, and the context's
//    stream waits for the transfer - no host synchronisation, and the transfer of level k - 1 runs under level k's search.
static int stage_images(cvhip_ctx *c, int k, const uint8_t *img1, size_t n1, const uint8_t *img2, size_t n2, hipStream_t s)
{
    const uint8_t *src[2] = {img1, img2};
    const size_t n[2] = {n1, n2};
    Device &d = c->dev->d;
    c->staged_from_pageable = false;
    const size_t off = img_level_offset(c->max_px, k);
    bool waited_readers = false;
    for (int i = 0; i < 2; i++) {
        if (is_device_ptr(src[i])) {
            if (c->borrow_inputs) {
                c->cur_img[i] = src[i];
            } else {
                CVHIP_TRY_HIP(hipMemcpyAsync(c->img[i] + off, src[i], n[i], hipMemcpyDeviceToDevice, s));
                c->cur_img[i] = c->img[i] + off;
            }
            continue;
        }
        c->cur_img[i] = c->img[i] + off;
        hipEvent_t done = nullptr;
        uint8_t *ring = nullptr;
        if (copy_stream_reserve(d) == CVHIP_OK) ring = upload_ring_take(d, n[i], 6 * c->max_px, &done);
        if (!ring) { // no ring: straight from the caller's pageable memory; the call synchronises before it returns
            CVHIP_TRY_HIP(hipMemcpyAsync(c->img[i] + off, src[i], n[i], hipMemcpyHostToDevice, s));
            c->staged_from_pageable = true;
            continue;
        }
        host_copy(ring, src[i], n[i]);
        if (!c->pool_waited) { // (the pool's clearing at context creation is work of the context's stream)
            if (c->pool_ready) CVHIP_TRY_HIP(hipStreamWaitEvent(d.up.stream, c->pool_ready, 0));
            c->pool_waited = true;
        }
        if (!waited_readers && k < 16 && c->level_read[k]) {
            CVHIP_TRY_HIP(hipStreamWaitEvent(d.up.stream, c->level_read[k], 0));
            waited_readers = true;
        }
        CVHIP_TRY_HIP(hipMemcpyAsync(c->img[i] + off, ring, n[i], hipMemcpyHostToDevice, d.up.stream));
        CVHIP_TRY_HIP(hipEventRecord(done, d.up.stream));
        CVHIP_TRY_HIP(hipStreamWaitEvent(s, done, 0));
    }
    return CVHIP_OK;
}

// The level's staged images have their last readers enqueued: an upload into the same area (the next pair's) waits here.
static int mark_level_read(cvhip_ctx *c, int k, hipStream_t s)
{
    if (k < 0 || k >= 16) return CVHIP_OK;
    // (both images borrowed from the caller: the level's area of the pool has no readers, and the event's barrier packet
    // would hold the stream for ~6 us between the search and the filter of every level)
    const size_t off = img_level_offset(c->max_px, k);
    if (c->cur_img[0] != c->img[0] + off && c->cur_img[1] != c->img[1] + off) return CVHIP_OK;
    if (!c->level_read[k]) CVHIP_TRY_HIP(hipEventCreateWithFlags(&c->level_read[k], hipEventDisableTiming));
    CVHIP_TRY_HIP(hipEventRecord(c->level_read[k], s));
    return CVHIP_OK;
}

// One search pass (mod.rs:247-319) given level images already staged in c->img[a] (searched) and c->img[b] (target)
// with their window statistics in c->istats[a], c->istats[b]: plan_pass validates the call and fills the job,
// launch_passes submits one pass - or the two independent passes of a level in the same launches - and commit_pass
// makes the new grid the direction's current one.
struct PassPlan {
    SearchJob job;
    int dir = 0, next = 0;
    uint32_t lw = 0, lh = 0, k = 0;
    enum Kind { EXACT_V1, BOX, FILTER } kind = FILTER;
    bool stepped = false, transposed = false, mfma = false, pair = false;
};

static int plan_pass(cvhip_ctx *c, int a, int b, uint32_t lw1, uint32_t lh1, uint32_t lw2, uint32_t lh2, float scale,
                     int k, int first_pass, int dir, PassPlan &plan, const uint32_t *rows = nullptr)
{
    DirState &ds = c->dir[dir];
    if (lw1 != (ds.gw >> k) || lh1 != (ds.gh >> k))
        return fail(CVHIP_ERR_UNSUPPORTED, "level dims must be floor(full * scale) (reconstruction.rs:146-152)");
    if (!first_pass) {
        if (!ds.valid) return fail(CVHIP_ERR_INVALID, "first_pass = 0 but this direction has no previous level");
        if ((int)ds.k <= k)
            return fail(CVHIP_ERR_UNSUPPORTED, "scale must shrink by powers of two from level to level");
    }
    if (dir == 1) c->rev_cross_check_pending = false; // the reverse grid is being replaced
    CorrParams &p = plan.job.p;
    std::memset(&p, 0, sizeof(p));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) p.F[i * 3 + j] = dir == 0 ? c->F[i * 3 + j] : c->F[j * 3 + i]; // mod.rs:268-271
    p.min_range = c->min_range;
    p.extend_range = c->extend_range;
    p.scale = scale;
    p.min_stdev = c->min_stdev;
    p.threshold = c->threshold;
    p.corridor_size = c->corridor_size;
    p.w1 = lw1;
    p.h1 = lh1;
    p.w2 = lw2;
    p.h2 = lh2;
    p.gw = ds.gw;
    p.gh = ds.gh;
    p.pw = ds.lw;
    p.ph = ds.lh;
    p.pk = ds.k;
    p.k = (uint32_t)k;
    p.first_pass = first_pass ? 1 : 0;
    p.need_scores = (dir == 0 && k == 0) || c->exact_scores ? 1 : 0;
    const double *F = p.F;
    const bool affine_form = F[0] == 0.0 && F[1] == 0.0 && F[3] == 0.0 && F[4] == 0.0;
    p.affine = 0;
    if (affine_form && std::isfinite(F[2]) && std::isfinite(F[5])) {
        // l = F*p1 in nalgebra's order: (0*p0 + 0*p1) + F02*1 = F02 (a zero keeps at most its sign, which no
        // result depends on), likewise F12
        const bool first = std::fabs(F[2]) > std::fabs(F[5]); // mod.rs:397
        const double div = first ? F[2] : F[5], cc = first ? -F[5] / F[2] : -F[2] / F[5];
        if (div != 0.0 && std::isfinite(cc)) {
            p.affine = first ? 1 : 2;
            p.aff_c = cc;
            p.aff_div = div;
        }
    }
    {
        const double up = (double)(1u << k), fdom = std::fabs(F[2]) > std::fabs(F[5]) ? std::fabs(F[2]) : std::fabs(F[5]);
        const double f2_bound = std::fabs(F[6]) * ((double)lw1 * up) + std::fabs(F[7]) * ((double)lh1 * up) + std::fabs(F[8]);
        p.range_quick = p.affine != 0 && fdom > 1e-150 && fdom < 1e150 && f2_bound < 1e149 ? 1 : 0; // (NaN: not quick)
    }
#ifdef CVHIP_ABLATIONS
    { // profiling ablations ("results are then wrong on purpose"): compiled only into -DCVHIP_ABLATIONS builds
        static const int dbg = [] { const char *v = std::getenv("CVHIP_DEBUG"); return v ? std::atoi(v) : 0; }();
        p.debug = k == 0 ? dbg : 0; // only the full-resolution level, so coarser levels still seed it
    }
#endif
    if (c->band_mode) {
        const uint32_t *r = dir == 0 ? c->band[k].sf : c->band[k].sr;
        p.row0 = std::min(r[0], lh1);
        p.row1 = std::min(r[1], lh1);
    } else {
        shard_rows(c, lh1, &p.row0, &p.row1);
    }
    if (rows) { // (the last level in result bands: level_search)
        p.row0 = std::min(rows[0], lh1);
        p.row1 = std::min(rows[1], lh1);
    }
    plan.dir = dir;
    plan.lw = lw1;
    plan.lh = lh1;
    plan.k = (uint32_t)k;
    plan.next = first_pass && !ds.valid ? ds.cur : 1 - ds.cur;
    SearchJob &j = plan.job;
    j.img1 = c->cur_img[a];
    j.img2 = c->cur_img[b];
    j.stats1 = c->istats[a] + stats_level_offset(c->max_px, k);
    j.stats2 = c->istats[b] + stats_level_offset(c->max_px, k);
    j.prev = ds.cells[ds.cur];
    j.range = dir == 0 ? c->range : c->range_rev;
    j.contenders = dir == 0 ? c->contenders : c->contenders_rev;
    j.out = ds.cells[plan.next];
    j.out_score = ds.scores;
    j.counters = c->count_candidates ? c->d_cand : nullptr;
    // the search kernel -> one persistent fallback kernel over the tiles the box filter declined and the tiles with
    // whole-corridor pixels (work lists filled by the producers)
    uint32_t *wc = c->work + 4 * dir;
    uint32_t *items = c->work + 8 + (size_t)dir * 2 * c->work_cap; // per-direction item arrays
    j.declined = WorkList{wc, items};
    j.whole = WorkList{wc + 1, items + c->work_cap};
    // The first pass as one workgroup per (tile, stripe): its scratch is the tail of the contender words - the level
    // uses the first lw1 x lh1 of them - where the buffer is large enough for both (any context beyond ~300^2 pixels).
    j.split = nullptr;
    if (first_pass) {
        const size_t own = (size_t)lw1 * lh1, need = search2_split_words(lw1, lh1, 2u * (uint32_t)c->corridor_size + 1u);
        if (lw1 <= 256 && lh1 <= 256 && own + need <= c->max_px) j.split = j.contenders + own;
    }

    // Which search kernel.  Version 3: the box filter is the search; the candidate filter only walks the workgroups it
    // declines (timed with the other fallback work, class K_EXACT).  The box walk pays where the candidate sets of
    // neighbouring pixels are (nearly) the same few image rows (columns): axis-near epipolar lines.  For an affine F
    // (F*p = (a, b, .) for every pixel) that is known up front: lines within ~4.5 degrees of the x or y axis (each extra
    // line crossed inside a workgroup's displacement box is one more plane per step).  Steeper lines go to the
    // candidate filter directly.  Purely a performance choice - every path is exact.
    plan.kind = PassPlan::FILTER;
    if (c->search_version == 1) {
        plan.kind = PassPlan::EXACT_V1;
        return CVHIP_OK;
    }
    bool v3 = c->search_version >= 3;
    // column-major lines (|F*p|_x > |F*p|_y, mod.rs:397): the transposed instantiation of the box kernel
    bool transposed = std::fabs(F[2]) > std::fabs(F[5]);
    const double f_major = transposed ? std::fabs(F[2]) : std::fabs(F[5]);
    double f_minor = transposed ? std::fabs(F[5]) : std::fabs(F[2]);
    double slope = f_major > 0.0 ? f_minor / f_major : 0.0; // rows the lines climb per step along their major axis (<= 1)
    if (!affine_form) {
        // Perspective F: the line direction (l.x, l.y) = first two components of F*p is an affine function of the pixel,
        // so "nearer to one axis than to the other, on one side of it" at the four image corners holds for every pixel
        // in between (an intersection of half-planes).  Then every pixel's line has the same major axis and the box walk
        // applies, with per-step plane windows (the STEP instantiation); its slope is largest at a corner.
        const double up = (double)(1u << k), xs[2] = {0.0, (double)(lw1 - 1) * up}, ys[2] = {0.0, (double)(lh1 - 1) * up};
        int along_x = 0, along_y = 0, sign_major = 0;
        bool same_side = true;
        slope = 0.0;
        for (double cx : xs)
            for (double cy : ys) {
                const double lx = (F[0] * cx + F[1] * cy) + F[2], ly = (F[3] * cx + F[4] * cy) + F[5];
                if (!(std::isfinite(lx) && std::isfinite(ly))) same_side = false;
                const bool row_major = std::fabs(lx) <= std::fabs(ly), col_major = std::fabs(ly) < std::fabs(lx);
                along_x += row_major ? 1 : 0;
                along_y += col_major ? 1 : 0;
                const double major = row_major ? ly : lx, minor = row_major ? lx : ly;
                if (major != 0.0) slope = std::max(slope, std::fabs(minor / major));
                const int sg = major > 0.0 ? 1 : -1;
                if (sign_major == 0) sign_major = sg;
                same_side = same_side && sg == sign_major && major != 0.0;
            }
        const bool one_axis = same_side && (along_x == 4 || along_y == 4);
        transposed = along_y == 4;
        f_minor = 1.0; // lines differ per pixel: always the stepped instantiation
        if (v3 && !c->force_box) v3 = one_axis;
    } else if (v3 && !c->force_box) {
        v3 = f_major > 0.0;
    }
    // The first pass searches the whole line: its displacement boxes are wider than the box kernel's 61 steps, every
    // workgroup would decline - straight to the candidate filter (one launch less on the latency-bound coarsest level).
    if (first_pass && !c->force_box) v3 = false;
    if (v3) {
        plan.kind = PassPlan::BOX;
        plan.stepped = f_minor != 0.0 || c->force_box; // exactly axis-parallel lines never step: the leaner instantiation
        plan.transposed = transposed;
        // rectified affine pairs (five stripes, row-major lines that never step): the filter as a matrix product
        plan.mfma = !plan.stepped && !plan.transposed && affine_form && c->corridor_size == 2 && c->search_version == 5;
        // ... or the box walk with two image columns per lane (search3_box2_kernel; any corridor size up to nine planes)
        plan.pair = !plan.stepped && !plan.transposed && affine_form && c->search_version == 6;
        if (plan.stepped) {
            // LDS of the stepped launch, sized for displacement boxes of up to ~32 steps along lines of this slope: H
            // rows of planes -> H + 3 rows of candidate statistics and H + 26 bytes of every target line (an odd number
            // of dwords: consecutive lines then start in different LDS banks).  64 KB per workgroup at most.
            const uint32_t wv = 2u * (uint32_t)c->corridor_size + 1u;
            const uint32_t H = (uint32_t)std::ceil(std::min(slope, 1.0) * 32.0) + wv + 3u;
            p.box_sh = std::min(H + 3u, 44u);
            p.box_pd = std::min(((H + 26u + 3u) >> 2) | 1u, 39u);
            // boxes up to 61 steps wide where that still leaves five workgroups per CU (160 KB of LDS), else up to 33
            p.box_wide = 128u * p.box_pd * 4u + p.box_sh * 128u * 8u <= 32u * 1024u ? 1u : 0u;
        }
    }
    return CVHIP_OK;
}

// n = 1, or the two passes of one level (independent: each reads its own direction's previous grid and writes its own
// buffers); passes that need the same kernels go out in the same launches (blockIdx.z), anything else one by one.
// stats_done (optional): an event the search kernels have to wait for, but not the search range (statistics ahead)
static int launch_passes(cvhip_ctx *c, PassPlan *plans, int n, bool zero_counts, hipStream_t s, hipEvent_t stats_done = nullptr)
{
    const bool together = n == 2 && plans[0].kind == plans[1].kind && plans[0].kind != PassPlan::EXACT_V1 &&
                          plans[0].stepped == plans[1].stepped && plans[0].transposed == plans[1].transposed && plans[0].mfma == plans[1].mfma && plans[0].pair == plans[1].pair &&
                          plans[0].job.p.first_pass == plans[1].job.p.first_pass;
    for (int i = 0; i < n; i += together ? 2 : 1) {
        const int m = together ? 2 : 1;
        SearchJob jobs[2] = {plans[i].job, plans[i + m - 1].job};
        const PassPlan &pl = plans[i];
        const CorrParams &p = pl.job.p;
        if (!p.first_pass)
            CVHIP_TRY(timed(c, cvhip_ctx::K_RANGE, [&] { launch_search_range(jobs, m, c->range_mode, s); }, s));
        if (stats_done && i == 0) CVHIP_TRY_HIP(hipStreamWaitEvent(s, stats_done, 0));
        if (pl.kind == PassPlan::EXACT_V1) {
            CVHIP_TRY(timed(c, cvhip_ctx::K_SEARCH, [&] {
                launch_search(p, pl.job.img1, pl.job.img2, pl.job.stats1, pl.job.stats2, pl.job.range, pl.job.out,
                              pl.job.out_score, pl.job.counters, s);
            }, s));
        } else if (!(p.debug & 2)) {
            // both directions' work-list counts are zeroed once per level by the statistics kernel of
            // cvhip_correlate_level; per-pass callers zero here
            if (zero_counts)
                for (int q = 0; q < m; q++) CVHIP_TRY_HIP(hipMemsetAsync(jobs[q].declined.count, 0, 4 * sizeof(uint32_t), s));
            if (pl.kind == PassPlan::BOX) {
                // The stepped instantiations go out as one launch per direction: on large levels the second one runs on a side
                // stream of the handle, filling the first one's tail (small levels: the fork / join costs more than it saves).
                hipStream_t side = nullptr;
                Device &d = c->dev->d;
                if (pl.stepped && m == 2 && !c->time_kernels && (size_t)p.w1 * (p.row1 - p.row0) >= d.box_fork_min_px && s == d.stream) {
                    for (hipEvent_t &ev : d.box_ev)
                        if (!ev) CVHIP_TRY_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    CVHIP_TRY_HIP(aux_stream(d, 1, &side));
                }
                bool fallbacks_out = false; // (on two streams each direction's fallback kernel follows its own box kernel)
                CVHIP_TRY(timed(c, cvhip_ctx::K_SEARCH, [&] {
                    fallbacks_out = launch_search3_box(jobs, m, pl.stepped, pl.transposed, pl.mfma ? 1 : (pl.pair ? 2 : 0), s, side, d.box_ev[0],
                                                       d.box_ev[1], side ? ((p.debug & 1) ? 1 : 0) : -1);
                }, s));
                if (!fallbacks_out)
                    CVHIP_TRY(timed(c, cvhip_ctx::K_EXACT, [&] { launch_search3_fallback(jobs, m, (p.debug & 1) != 0, s); }, s));
            } else {
                // candidate filter over every tile; the (rare) tiles with whole-corridor pixels queue themselves for
                // the fallback kernel, whose declined list stays empty here
                CVHIP_TRY(timed(c, cvhip_ctx::K_FILTER, [&] { launch_search2_filter(jobs, m, s); }, s));
                if (!(p.debug & 1))
                    CVHIP_TRY(timed(c, cvhip_ctx::K_EXACT, [&] { launch_search3_fallback(jobs, m, false, s); }, s));
            }
        }
    }
    CVHIP_TRY_HIP(hipGetLastError());
    return CVHIP_OK;
}

static void commit_pass(cvhip_ctx *c, const PassPlan &plan)
{
    DirState &ds = c->dir[plan.dir];
    c->live_bands = 0; // (level_search sets it again behind its own commits)
    ds.cur = plan.next;
    ds.valid = true;
    ds.scores_valid = plan.job.p.need_scores != 0 || plan.kind == PassPlan::EXACT_V1; // (the plain kernel scores every pixel)
    ds.lw = plan.lw;
    ds.lh = plan.lh;
    ds.k = plan.k;
}

static int search_pass(cvhip_ctx *c, int a, int b, uint32_t lw1, uint32_t lh1, uint32_t lw2, uint32_t lh2,
                       float scale, int k, int first_pass, int dir)
{
    PassPlan plan;
    CVHIP_TRY(plan_pass(c, a, b, lw1, lh1, lw2, lh2, scale, k, first_pass, dir, plan));
    CVHIP_TRY(launch_passes(c, &plan, 1, true, c->dev->d.stream));
    commit_pass(c, plan);
    return CVHIP_OK;
}

static int cross_check_pass(cvhip_ctx *c, int k, int dir)
{
    if (dir == 0) c->live_bands = 0; // (the forward grid changes on the context's stream)
    DirState &own = c->dir[dir];
    DirState &other = c->dir[1 - dir];
    if (!own.valid || !other.valid) return fail(CVHIP_ERR_INVALID, "cross_check_filter before both passes ran");
    if ((int)own.k != k || (int)other.k != k)
        return fail(CVHIP_ERR_INVALID, "cross_check_filter scale does not match the grids' current level");
    uint32_t r0 = 0, r1 = own.lh;
    if (c->band_mode) {
        const uint32_t *r = dir == 0 ? c->band[k].cf : c->band[k].cr;
        r0 = std::min(r[0], own.lh);
        r1 = std::min(r[1], own.lh);
    } else if (dir == 1 && k == 0) { // nothing reads the filtered reverse grid of the last level: deferred
        c->rev_cross_check_pending = true;
        return CVHIP_OK;
    }
    CVHIP_TRY(timed(c, cvhip_ctx::K_CROSS, [&] {
        launch_cross_check(own.cells[own.cur], other.cells[other.cur], own.lw, own.lh, other.lw, other.lh, r0, r1,
                           c->dev->d.stream);
    }));
    CVHIP_TRY_HIP(hipGetLastError());
    return CVHIP_OK;
}

// the deferred reverse cross-check of the full-resolution level (cvhip_ctx::rev_cross_check_pending)
static int flush_reverse_cross_check(cvhip_ctx *c)
{
    if (!c->rev_cross_check_pending) return CVHIP_OK;
    c->rev_cross_check_pending = false;
    DirState &own = c->dir[1], &other = c->dir[0];
    if (!own.valid || !other.valid || own.k != 0 || other.k != 0) return CVHIP_OK;
    CVHIP_TRY(timed(c, cvhip_ctx::K_CROSS, [&] {
        launch_cross_check(own.cells[own.cur], other.cells[other.cur], own.lw, own.lh, other.lw, other.lh, 0, own.lh, c->dev->d.stream);
    }));
    CVHIP_TRY_HIP(hipGetLastError());
    return CVHIP_OK;
}

static int check_level_args(const cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2,
                            uint32_t w2, uint32_t h2, float scale, int *k)
{
    if (!ctx || !img1 || !img2) return fail(CVHIP_ERR_INVALID, "null argument");
    *k = scale_to_k(scale);
    if (*k < 0) return fail(CVHIP_ERR_UNSUPPORTED, "scale must be 2^-k, 0 <= k <= 15 (reconstruction.rs:566)");
    if (w1 < KERNEL_WIDTH || h1 < KERNEL_WIDTH || w2 < KERNEL_WIDTH || h2 < KERNEL_WIDTH)
        return fail(CVHIP_ERR_INVALID, "level image smaller than the 11x11 correlation window");
    if (w1 > 65535 || h1 > 65535 || w2 > 65535 || h2 > 65535)
        return fail(CVHIP_ERR_UNSUPPORTED, "level dimension above 65535");
    if ((size_t)w1 * h1 > ctx->max_px || (size_t)w2 * h2 > ctx->max_px)
        return fail(CVHIP_ERR_INVALID, "level image larger than the context's full-resolution images");
    // (the level's area of the image pool holds max_px / 4^k pixels: checked before anything is staged)
    if ((size_t)w1 * h1 > (ctx->max_px >> (2 * *k)) || (size_t)w2 * h2 > (ctx->max_px >> (2 * *k)))
        return fail(CVHIP_ERR_UNSUPPORTED, "level dims must be floor(full * scale) (reconstruction.rs:146-152)");
    return CVHIP_OK;
}

static void report(cvhip_progress_fn progress, void *user, int dir, float value)
{
    // same mapping as GpuContext::correlate_images' send_progress (gpu/mod.rs:241-249)
    if (!progress) return;
    progress(user, dir == 0 ? value * 0.98f / 2.0f : 0.51f + value * 0.98f / 2.0f);
}

void device_free(cvhip_device *dev)
{
    (void)hipSetDevice(dev->d.ordinal);
    if (dev->d.stream) {
        (void)hipStreamSynchronize(dev->d.stream);
        if (dev->d.owns_stream) (void)hipStreamDestroy(dev->d.stream);
    }
    for (auto &b : dev->d.parked) free_buffer_set(b);
    dev->d.parked.clear();
    {
        auto &rb = dev->d.rb;
        if (rb.stream) (void)hipStreamSynchronize(rb.stream);
        if (dev->d.up.stream) (void)hipStreamSynchronize(dev->d.up.stream);
        for (int i = 0; i < 2; i++) {
            if (rb.xy[i]) (void)hipFree(rb.xy[i]);
            if (rb.corr[i]) (void)hipFree(rb.corr[i]);
            if (rb.done[i]) (void)hipEventDestroy(rb.done[i]);
        }
        if (rb.ready) (void)hipEventDestroy(rb.ready);
        if (rb.expanded) (void)hipEventDestroy(rb.expanded);
        auto &up = dev->d.up; // (the copy stream has drained: no transfer reads the ring any more)
        for (auto &c : up.busy) (void)hipEventDestroy(c.done);
        for (hipEvent_t ev : up.spare) (void)hipEventDestroy(ev);
        up.busy.clear();
        up.spare.clear();
        if (up.base) (void)hipHostFree(up.base);
        up.base = nullptr;
        if (rb.stream) (void)hipStreamDestroy(rb.stream);
        if (up.stream) (void)hipStreamDestroy(up.stream);
    }
    {
        auto &rq = dev->d.rq;
        for (hipEvent_t ev : rq.ready)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : rq.scored)
            if (ev) (void)hipEventDestroy(ev);
        if (rq.uploaded) (void)hipEventDestroy(rq.uploaded);
        if (rq.started) (void)hipEventDestroy(rq.started);
    }
    for (hipEvent_t &ev : dev->d.orb_ev)
        if (ev) {
            (void)hipEventDestroy(ev);
            ev = nullptr;
        }
    for (hipEvent_t &ev : dev->d.box_ev)
        if (ev) {
            (void)hipEventDestroy(ev);
            ev = nullptr;
        }
    for (hipStream_t &g : dev->d.aux)
        if (g) {
            (void)hipStreamSynchronize(g);
            (void)hipStreamDestroy(g);
            g = nullptr;
        }
    {
        auto &sa = dev->d.sa;
        for (hipEvent_t ev : sa.done)
            if (ev) (void)hipEventDestroy(ev);
        if (sa.fence) (void)hipEventDestroy(sa.fence);
        if (sa.probe_side) (void)hipEventDestroy(sa.probe_side);
        if (sa.probe_main) (void)hipEventDestroy(sa.probe_main);
    }
    if (dev->d.arena.base) (void)hipFree(dev->d.arena.base);
    if (dev->d.pinned) (void)hipHostFree(dev->d.pinned);
    if (dev->d.orb_pattern) (void)hipFree(dev->d.orb_pattern);
    for (auto &t : dev->d.resize_tables) {
        if (t.idx) (void)hipFree(t.idx);
        if (t.weights) (void)hipFree(t.weights);
    }
    dev->d.resize_tables.clear();
    if (dev->d.resize_tmp) (void)hipFree(dev->d.resize_tmp);
    delete dev;
}

} // namespace cvhip

using namespace cvhip;

extern "C" {

const char *cvhip_last_error(void) { return g_last_error.c_str(); }
uint32_t cvhip_abi_version(void) { return 2; } // 2: listeners on cvhip_orb_extract / cvhip_find_ransac

static int device_create(int low_power, int ordinal, bool caller_stream, void *hip_stream, cvhip_device **out);

int cvhip_device_create(int low_power, int ordinal, cvhip_device **out)
{
    return device_create(low_power, ordinal, false, nullptr, out);
}

int cvhip_device_create_on_stream(int low_power, int ordinal, void *hip_stream, cvhip_device **out)
{
    // hip_stream == NULL is HIP's default (null) stream — e.g. torch's default current stream
    return device_create(low_power, ordinal, true, hip_stream, out);
}

static int device_create(int low_power, int ordinal, bool caller_stream, void *hip_stream, cvhip_device **out)
{
    if (!out) return fail(CVHIP_ERR_INVALID, "out is null");
    *out = nullptr;
    int count = 0;
    CVHIP_TRY_HIP(hipGetDeviceCount(&count));
    if (count <= 0) return fail(CVHIP_ERR_DEVICE, "no HIP device");
    if (ordinal < 0) CVHIP_TRY_HIP(hipGetDevice(&ordinal));
    if (ordinal >= count) return fail(CVHIP_ERR_INVALID, "device ordinal out of range");
    CVHIP_TRY_HIP(hipSetDevice(ordinal));
    hipDeviceProp_t prop;
    CVHIP_TRY_HIP(hipGetDeviceProperties(&prop, ordinal));
    cvhip_device *dev = new (std::nothrow) cvhip_device();
    if (!dev) return fail(CVHIP_ERR_NOMEM, "out of host memory");
    dev->d.ordinal = ordinal;
    dev->d.low_power = low_power;
    dev->d.name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    if (caller_stream) { // caller's stream (e.g. torch's current stream): work is ordered with the caller's
        dev->d.stream = reinterpret_cast<hipStream_t>(hip_stream);
        dev->d.owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&dev->d.stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete dev;
            return fail(CVHIP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
    }
    *out = dev;
    return CVHIP_OK;
}

void cvhip_device_destroy(cvhip_device *dev)
{
    if (!dev) return;
    (void)hipSetDevice(dev->d.ordinal);
    if (dev->d.stream) (void)hipStreamSynchronize(dev->d.stream);
    if (dev->d.comm_refs > 0) { // a communicator still enqueues on this handle's stream: its destroy finishes the job
        dev->d.destroy_pending = true;
        return;
    }
    device_free(dev);
}

const char *cvhip_device_name(const cvhip_device *dev) { return dev ? dev->d.name.c_str() : ""; }

int cvhip_device_synchronize(cvhip_device *dev)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "dev is null");
    CVHIP_TRY(set_device(dev));
    CVHIP_TRY_HIP(hipStreamSynchronize(dev->d.stream));
    if (dev->d.rb.stream) CVHIP_TRY_HIP(hipStreamSynchronize(dev->d.rb.stream)); // asynchronous readbacks in flight
    return CVHIP_OK;
}

int cvhip_ctx_create(cvhip_device *dev, uint32_t w1, uint32_t h1, uint32_t w2, uint32_t h2, int projection,
                     const double *F, cvhip_ctx **out)
{
    if (!dev || !F || !out) return fail(CVHIP_ERR_INVALID, "null argument");
    *out = nullptr;
    if (w1 < KERNEL_WIDTH || h1 < KERNEL_WIDTH || w2 < KERNEL_WIDTH || h2 < KERNEL_WIDTH)
        return fail(CVHIP_ERR_INVALID, "image smaller than the 11x11 correlation window");
    if (w1 > 65535 || h1 > 65535 || w2 > 65535 || h2 > 65535)
        return fail(CVHIP_ERR_UNSUPPORTED, "image dimension above 65535");
    if (projection != 0 && projection != 1) return fail(CVHIP_ERR_INVALID, "projection must be 0 or 1");
    CVHIP_TRY(set_device(dev));
    cvhip_ctx *c = new (std::nothrow) cvhip_ctx();
    if (!c) return fail(CVHIP_ERR_NOMEM, "out of host memory");
    c->dev = dev;
    c->w1 = w1;
    c->h1 = h1;
    c->w2 = w2;
    c->h2 = h2;
    c->projection = projection;
    std::memcpy(c->F, F, sizeof(c->F));
    if (projection == 0) { // mod.rs:120-126
        c->min_stdev = 1.0f;
        c->threshold = 0.6f;
        c->corridor_size = 2;
        c->min_range = 2.5;
        c->extend_range = 1.0;
    } else { // mod.rs:127-133
        c->min_stdev = 1.0f;
        c->threshold = 0.5f;
        c->corridor_size = 4;
        c->min_range = 0.75;
        c->extend_range = 0.5;
    }
    c->dir[0].gw = w1;
    c->dir[0].gh = h1;
    c->dir[1].gw = w2;
    c->dir[1].gh = h2;
#ifdef CVHIP_ABLATIONS
    // environment overrides for profiling runs; the shipped library ignores the environment (kernel selection and
    // the range-kernel test modes are set through cvhip_ctx_set_search_version / cvhip_ctx_set_range_mode)
    if (const char *v = std::getenv("CVHIP_SEARCH")) c->search_version = (v[0] >= '1' && v[0] <= '3') ? v[0] - '0' : 3;
    if (const char *v = std::getenv("CVHIP_FORCE_BOX")) c->force_box = v[0] == '1';
    if (const char *v = std::getenv("CVHIP_RANGE")) c->range_mode = (v[0] >= '0' && v[0] <= '3') ? v[0] - '0' : 0;
#endif
    const size_t n1 = (size_t)w1 * h1, n2 = (size_t)w2 * h2;
    c->max_px = std::max(n1, n2);
    // Level grids are gathered in equal row chunks when sharded, so leave room for one padded
    // chunk: (rows + den - 1) rows at most; 64 extra rows cover any den <= 64.
    auto grid_elems = [](uint32_t w, uint32_t h) { return (size_t)w * ((size_t)h + 64); };
    hipError_t e = hipSuccess;
    c->work_cap = 2 * search3_worklist_capacity(std::max(w1, w2), std::max(h1, h2));
    bool reused = false;
    for (size_t i = dev->d.parked.size(); i-- > 0 && !reused;) { // a parked set of the same dimensions (see CtxBuffers)
        CtxBuffers &b = dev->d.parked[i];
        if (b.w1 != w1 || b.h1 != h1 || b.w2 != w2 || b.h2 != h2) continue;
        for (int d = 0; d < 2; d++) {
            for (int k = 0; k < 2; k++) c->dir[d].cells[k] = b.cells[d][k];
            c->dir[d].scores = b.scores[d];
            c->img[d] = b.img[d];
            c->istats[d] = b.istats[d];
        }
        c->range = b.range;
        c->range_rev = b.range_rev;
        c->contenders = b.contenders;
        c->contenders_rev = b.contenders_rev;
        c->work = b.work;
        c->d_cand = b.d_cand;
        dev->d.parked.erase(dev->d.parked.begin() + (long)i);
        reused = true;
    }
    for (int d = 0; d < 2 && e == hipSuccess && !reused; d++) {
        const size_t ge = grid_elems(c->dir[d].gw, c->dir[d].gh);
        for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc(&c->dir[d].cells[i], ge * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->dir[d].scores, ge * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&c->img[d], img_pool_bytes(c->max_px));
        if (e == hipSuccess) e = hipMalloc(&c->istats[d], stats_pool_elems(c->max_px) * sizeof(uint2));
    }
    if (!reused) {
        if (e == hipSuccess) e = hipMalloc(&c->range, c->max_px * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->range_rev, c->max_px * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->contenders, c->max_px * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&c->contenders_rev, c->max_px * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&c->work, (8 + 4 * c->work_cap) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->d_cand, 4 * sizeof(unsigned long long));
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->d_cand, 0, 4 * sizeof(unsigned long long), dev->d.stream);
    for (int d = 0; d < 2 && e == hipSuccess; d++)
        e = hipMemsetAsync(c->img[d], 0, img_pool_bytes(c->max_px), dev->d.stream);
    // (host level images are uploaded on the handle's COPY stream: it must not overtake the clearing above)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pool_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(c->pool_ready, dev->d.stream);
    if (e != hipSuccess) {
        if (c->pool_ready) (void)hipEventDestroy(c->pool_ready);
        free_ctx_buffers(c);
        delete c;
        return fail(e == hipErrorOutOfMemory ? CVHIP_ERR_NOMEM : CVHIP_ERR_DEVICE,
                    std::string("allocating device buffers: ") + hipGetErrorString(e));
    }
    *out = c;
    return CVHIP_OK;
}

void cvhip_ctx_destroy(cvhip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev->d.ordinal);
    // (uploads into the image pool: their own stream.  The copy stream's band expansions read this context's planes, but the
    // context's stream - synchronised next - was made to follow the last of them, complete_grid; the host-bound copies read the
    // handle's staging sets and need not hold a context's destruction up)
    if (ctx->dev->d.up.stream) (void)hipStreamSynchronize(ctx->dev->d.up.stream);
    (void)hipStreamSynchronize(ctx->dev->d.stream);
    for (hipEvent_t &ev : ctx->level_read)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->pool_ready) (void)hipEventDestroy(ctx->pool_ready);
    for (hipEvent_t &ev : ctx->band_done)
        if (ev) (void)hipEventDestroy(ev);
    free_ctx_buffers(ctx, true); // the buffer set is parked on the device handle for the next pair
    delete ctx;
}

// ---------------------------------------------------------------------------------------------------------------
// One pyramid level = the four backend calls of PointCorrelations::correlate_images (correlation/mod.rs:217-245):
// correlate_images forward, correlate_images reverse (images exchanged, F transposed), cross_check_filter forward,
// cross_check_filter reverse.  The work of a level is cut into three pieces that cvhip_correlate_level runs back to
// back and that the four per-pass calls run ONE BY ONE when the caller has promised the reference's call order
// (cvhip_ctx_set_fuse_level_calls): level_begin (images in, window statistics), level_search (both search passes in
// the same launches), level_cross (both cross-checks in one launch).  Without that promise the per-pass calls are
// executed independently, each for itself, as before.
// ---------------------------------------------------------------------------------------------------------------
static bool host_minor_offset_bound(const cvhip_ctx *c, int dir, int k, uint32_t lw1, uint32_t lh1, uint32_t lw2, uint32_t lh2, double *bound);
namespace {
struct LevelStats { // what level_begin leaves for the level's search launches
    bool ahead = false;          // the statistics run on the side stream (cvhip_ctx_set_stats_ahead)
    hipEvent_t done = nullptr;   // ... and this event marks them
};

// Images in (transfer_in_images, gpu/mod.rs:274) and the window statistics of both (compute_image_point_data,
// mod.rs:632-694), in line or ahead on the side stream.
int level_begin(cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2, uint32_t w2, uint32_t h2, int k,
                int first_pass, bool sharded, bool clear_work, LevelStats &ls)
{
    hipStream_t s = ctx->dev->d.stream;
    CVHIP_TRY(stage_images(ctx, k, img1, (size_t)w1 * h1, img2, (size_t)w2 * h2, s));
    uint32_t sr0 = 0, sr1 = std::max(h1, h2);
    if (ctx->band_mode) {
        sr0 = ctx->band[k].st[0];
        sr1 = ctx->band[k].st[1];
    }
    uint2 *const st0 = ctx->istats[0] + stats_level_offset(ctx->max_px, k), *const st1 = ctx->istats[1] + stats_level_offset(ctx->max_px, k);
    bool stats_ahead = ctx->stats_ahead && ctx->time_kernels != 1 && !sharded && ctx->borrow_inputs &&
                       is_device_ptr(img1) && is_device_ptr(img2) && k < 16;
    if (stats_ahead && ctx->dev->d.sa.verdict == 0 && ctx->dev->d.sa.probe_recorded && first_pass) {
        // the previous run's probe, if it has completed (never waits)
        Device::StatsAhead &sa = ctx->dev->d.sa;
        float lead_ms = 0.0f;
        const hipError_t pe = hipEventElapsedTime(&lead_ms, sa.probe_side, sa.probe_main);
        if (pe == hipSuccess) {
            sa.verdict = lead_ms > 0.05f ? 1 : -1; // (overlapping: ~1 ms at 4096^2; in line: zero or negative)
            sa.probe_recorded = false;
        } else {
            (void)hipGetLastError(); // hipErrorNotReady: ask again at the next run
        }
    }
    if (stats_ahead && ctx->dev->d.sa.verdict < 0) stats_ahead = false;
    ls.ahead = stats_ahead;
    ls.done = nullptr;
    if (stats_ahead) {
        // The statistics depend on the level's images only.  On a stream of their own they run while the main stream
        // works through the coarse levels - a chain of ~40 small dependent launches that leaves the chip idle for
        // ~0.4 ms of a 4096^2 pair - instead of 0.5 ms of full-chip work in line with it.
        Device &d = ctx->dev->d;
        hipStream_t side = nullptr;
        CVHIP_TRY_HIP(aux_stream(d, 0, &side));
        if (!d.sa.fence) CVHIP_TRY_HIP(hipEventCreateWithFlags(&d.sa.fence, hipEventDisableTiming));
        if (!d.sa.done[k]) CVHIP_TRY_HIP(hipEventCreateWithFlags(&d.sa.done[k], hipEventDisableTiming));
        if (first_pass || !ctx->stats_ahead_fenced) {
            // a new pyramid run: its side-stream work starts behind everything enqueued so far (the previous run's
            // readers of these buffers among it), and its first level finds the work-list counts cleared
            CVHIP_TRY_HIP(hipEventRecord(d.sa.fence, s));
            CVHIP_TRY_HIP(hipStreamWaitEvent(side, d.sa.fence, 0));
            CVHIP_TRY_HIP(hipMemsetAsync(ctx->work, 0, 8 * sizeof(uint32_t), s));
            ctx->stats_ahead_fenced = true;
        }
        // (only where the answer is unambiguous: a full-resolution level of a megapixel or more, several levels deep)
        const bool probe = d.sa.verdict == 0 && !d.sa.probe_recorded && k == 0 && !first_pass && (size_t)w1 * h1 >= ((size_t)1 << 20);
        if (probe) {
            if (!d.sa.probe_side) CVHIP_TRY_HIP(hipEventCreate(&d.sa.probe_side));
            if (!d.sa.probe_main) CVHIP_TRY_HIP(hipEventCreate(&d.sa.probe_main));
            CVHIP_TRY_HIP(hipEventRecord(d.sa.probe_side, side)); // the full-resolution statistics start here ...
            CVHIP_TRY_HIP(hipEventRecord(d.sa.probe_main, s));    // ... and the main stream gets here when level 1 is done
            d.sa.probe_recorded = true;
        }
        // (48 KB of LDS ballast per workgroup: three of them per CU instead of eight.  A full-chip grid beside the
        // coarse levels starved their small kernels - a 30 us box launch took the 370 us of the statistics kernel,
        // stream priorities notwithstanding; at four per CU the chain still lost 0.17 ms; at two the statistics of the
        // full-resolution level are not done when that level's turn comes: step 5.60 / 5.45 / 5.37 / 5.47 ms for
        // 0 / 32 / 48 / 64 KB in round 4; 4.93 / 4.86 / 4.86 / 4.84 ms for 24 / 32 / 40 / 48 KB with the four-pixel kernel)
        launch_window_stats_pair(ctx->cur_img[0], w1, h1, st0, ctx->cur_img[1], w2, h2, st1, sr0, sr1, ctx->min_stdev, nullptr, side,
                                 48u * 1024u);
        CVHIP_TRY_HIP(hipEventRecord(d.sa.done[k], side));
        ls.done = d.sa.done[k]; // (the search range does not read the statistics: the wait goes behind its launch)
    } else {
        ctx->stats_ahead_fenced = false;
        // both images in one launch, which (clear_work) also clears the work-list counts of the level's two search passes
        CVHIP_TRY(timed(ctx, cvhip_ctx::K_STATS, [&] {
            launch_window_stats_pair(ctx->cur_img[0], w1, h1, st0, ctx->cur_img[1], w2, h2, st1, sr0, sr1, ctx->min_stdev,
                                     clear_work ? ctx->work : nullptr, s);
        }));
    }
    return CVHIP_OK;
}

// Both search passes of the level (mod.rs:224-237), in the same launches, and - row-shard mode - the all-gathers.
// (w1, h1) / (w2, h2): the FORWARD call's images.
int level_search(cvhip_ctx *ctx, bool with_filters, uint32_t w1, uint32_t h1, uint32_t w2, uint32_t h2, float scale, int k, int first_pass, bool sharded,
                 const LevelStats &ls, cvhip_progress_fn progress, void *user)
{
    hipStream_t s = ctx->dev->d.stream;
    const uint32_t den = ctx->shard_den, num = ctx->shard_num;
    if (!sharded) {
        ctx->shard_num = 0;
        ctx->shard_den = 1;
    }
    // The two search passes of a level are independent (each reads only its own direction's previous grid and
    // writes only its own), so they go out in the same launches: on the small levels - and on the short bands of a
    // many-GPU run - neither fills the GPU alone, and a level costs five dependent launches instead of eight.
    // A host hook enqueues its collective on a stream the library cannot see.  When the device handle was created on
    // the CALLER's stream, the hook is required to use that same stream (include/cvhip.h) and ordering follows.  When
    // the handle owns a private stream there is no such stream to share, so the library fences both sides itself:
    // the search pass has finished before the hook runs, and everything the hook enqueued anywhere on this GPU has
    // finished before the next kernel is submitted.  (The library's own RCCL path needs neither: it enqueues on s.)
    const bool fence_hook = sharded && !ctx->gather_on_stream && ctx->dev->d.owns_stream;
    const auto run_gather = [&](int dir) -> int {
        const DirState &ds = ctx->dir[dir];
        // (only the match plane travels between passes: nothing reads another rank's scores before the final gather)
        const uint64_t shard_bytes = (uint64_t)((ds.lh + den - 1) / den) * ds.lw * sizeof(uint32_t);
        if (fence_hook) CVHIP_TRY_HIP(hipStreamSynchronize(s));
        if (ctx->gather(ctx->gather_user, ds.cells[ds.cur], shard_bytes, den, dir) != 0)
            return fail(CVHIP_ERR_DEVICE, dir == 0 ? "all-gather hook failed (forward grid)" : "all-gather hook failed (reverse grid)");
        // the forward SCORES of the last level are the only ones complete() reports: every rank gets the other bands' too
        if (dir == 0 && k == 0 && ctx->gather(ctx->gather_user, ds.scores, shard_bytes, den, 2) != 0)
            return fail(CVHIP_ERR_DEVICE, "all-gather hook failed (forward scores)");
        if (fence_hook) CVHIP_TRY_HIP(hipDeviceSynchronize());
        return CVHIP_OK;
    };
    ctx->live_bands = 0;
    ctx->bands_crossed = false;
    // The last level in result bands (cvhip_ctx_set_result_bands): the two search passes of rows [r_b, r_b+1), then the
    // forward filter of rows [r_b - E, r_b+1 - E) - a filtered cell reads the unfiltered reverse cells within D + 4 rows of
    // its own (mod.rs:588-624), all of them searched by then - and an event; the last filter runs to the grid's end.  Only
    // where the geometry is row-local (the bound D of independent-band mode) and the bands are tall enough; the reverse
    // filter of the last level is deferred anyway (rev_cross_check_pending).
    uint32_t nb = 1, reach = 0;
    // (not under asynchronous readback: there the whole transfer runs under the NEXT pair's search, and the context's stream
    // must follow the last band's expansion, which would stand behind the earlier bands' copies)
    if (with_filters && k == 0 && !first_pass && ctx->result_bands != 1 && !ctx->async_readback && !sharded && !ctx->band_mode &&
        !ctx->time_kernels && !ctx->count_candidates && h1 == h2) {
        double df = 0.0, dr = 0.0;
        if (host_minor_offset_bound(ctx, 0, 0, w1, h1, w2, h2, &df) && host_minor_offset_bound(ctx, 1, 0, w2, h2, w1, h1, &dr)) {
            reach = ((uint32_t)std::max(df, dr) + CROSS_CHECK_SEARCH_AREA + 4 + 3) / 4 * 4;
            // (0 = the library's choice: a band of at least half a megapixel, six at most - 1024^2: 2, 2048^2 and up: 6;
            // smaller bands cost more in launch tails than their transfer hides, scripts/result_bands_probe.py)
            nb = ctx->result_bands ? std::min<uint32_t>(ctx->result_bands, 16u)
                                   : (uint32_t)std::min<size_t>(6, std::max<size_t>(1, ((size_t)w1 * h1) >> 19));
            while (nb > 1 && (h1 / nb) / 4 * 4 < 2 * reach + 64) nb--;
        }
    }
    if (nb > 1) {
        uint32_t rows[17];
        for (uint32_t b = 0; b <= nb; b++) {
            rows[b] = b == nb ? h1 : (uint32_t)((uint64_t)h1 * b / nb) / 4u * 4u;
            ctx->band_rows[b] = (b == 0 || b == nb) ? rows[b] : rows[b] - reach; // (the filter's and the copy's bands)
        }
        int rc = CVHIP_OK;
        // (every band is planned against the grids' state before the level; committed once the first band is out)
        PassPlan plans[16][2];
        for (uint32_t b = 0; b < nb && rc == CVHIP_OK; b++) {
            rc = plan_pass(ctx, 0, 1, w1, h1, w2, h2, scale, k, first_pass, 0, plans[b][0], &rows[b]);
            if (rc == CVHIP_OK) rc = plan_pass(ctx, 1, 0, w2, h2, w1, h1, scale, k, first_pass, 1, plans[b][1], &rows[b]);
        }
        for (uint32_t b = 0; b < nb && rc == CVHIP_OK; b++) {
            // (work-list counts: band 0 finds them cleared like any level, the filter clears them for the band behind it)
            rc = launch_passes(ctx, plans[b], 2, false, s, b == 0 ? ls.done : nullptr);
            if (rc != CVHIP_OK) break;
            if (b == 0) {
                commit_pass(ctx, plans[0][0]);
                commit_pass(ctx, plans[0][1]);
            }
            DirState &df_ = ctx->dir[0], &dr_ = ctx->dir[1];
            rc = timed(ctx, cvhip_ctx::K_CROSS, [&] {
                launch_cross_check_pair(df_.cells[df_.cur], dr_.cells[dr_.cur], df_.lw, df_.lh, dr_.lw, dr_.lh, ctx->band_rows[b],
                                        ctx->band_rows[b + 1], 0u, 0u, s, (b + 1 < nb || ls.ahead) ? ctx->work : nullptr);
            });
            if (rc == CVHIP_OK && !ctx->band_done[b] && hipEventCreateWithFlags(&ctx->band_done[b], hipEventDisableTiming) != hipSuccess)
                rc = fail(CVHIP_ERR_DEVICE, "hipEventCreate (result band)");
            if (rc == CVHIP_OK && hipEventRecord(ctx->band_done[b], s) != hipSuccess) rc = fail(CVHIP_ERR_DEVICE, "hipEventRecord (result band)");
        }
        if (rc != CVHIP_OK && ls.done) (void)hipStreamWaitEvent(s, ls.done, 0);
        if (rc == CVHIP_OK) rc = mark_level_read(ctx, k, s);
        CVHIP_TRY(rc);
        ctx->rev_cross_check_pending = true;
        ctx->live_bands = nb;
        ctx->bands_crossed = true;
        report(progress, user, 0, 1.0f);
        report(progress, user, 1, 1.0f);
        return CVHIP_OK;
    }
    PassPlan plans[2];
    int rc = plan_pass(ctx, 0, 1, w1, h1, w2, h2, scale, k, first_pass, 0, plans[0]);                 // mod.rs:224-230
    if (rc == CVHIP_OK) rc = plan_pass(ctx, 1, 0, w2, h2, w1, h1, scale, k, first_pass, 1, plans[1]); // mod.rs:231-237
    if (rc == CVHIP_OK) rc = launch_passes(ctx, plans, 2, false, s, ls.done);
    else if (ls.done) (void)hipStreamWaitEvent(s, ls.done, 0); // (the side stream's work stays ordered before whatever follows)
    if (rc == CVHIP_OK) {
        commit_pass(ctx, plans[0]);
        commit_pass(ctx, plans[1]);
        rc = mark_level_read(ctx, k, s);
    }
    report(progress, user, 0, 1.0f);
    if (rc == CVHIP_OK && sharded) rc = run_gather(0);
    if (rc == CVHIP_OK && sharded) rc = run_gather(1);
    ctx->shard_num = num;
    ctx->shard_den = den;
    CVHIP_TRY(rc);
    report(progress, user, 1, 1.0f);
    return CVHIP_OK;
}

// mod.rs:239-240: forward then reverse cross-check.  They commute and each depends only on the unfiltered other grid
// (DESIGN.md section 5), so one launch runs both.
int level_cross(cvhip_ctx *ctx, int k, bool stats_ahead)
{
    if (ctx->bands_crossed && k == 0) { // the last level's filters went out with its result bands (level_search)
        ctx->bands_crossed = false;
        return CVHIP_OK;
    }
    hipStream_t s = ctx->dev->d.stream;
    DirState &df = ctx->dir[0], &dr = ctx->dir[1];
    if (!df.valid || !dr.valid || (int)df.k != k || (int)dr.k != k)
        return fail(CVHIP_ERR_INVALID, "cross-check before both passes of the level ran");
    uint32_t f0 = 0, f1 = df.lh, r0 = 0, r1 = dr.lh;
    if (ctx->band_mode) {
        f0 = std::min(ctx->band[k].cf[0], df.lh);
        f1 = std::min(ctx->band[k].cf[1], df.lh);
        r0 = std::min(ctx->band[k].cr[0], dr.lh);
        r1 = std::min(ctx->band[k].cr[1], dr.lh);
    } else if (k == 0) { // the reverse filter of the last level is deferred (cvhip_ctx::rev_cross_check_pending)
        r0 = r1 = 0;
        ctx->rev_cross_check_pending = true;
    }
    CVHIP_TRY(timed(ctx, cvhip_ctx::K_CROSS, [&] {
        // (stats ahead: the next level's statistics kernel runs on another stream and cannot clear the work-list counts)
        launch_cross_check_pair(df.cells[df.cur], dr.cells[dr.cur], df.lw, df.lh, dr.lw, dr.lh, f0, f1, r0, r1, s,
                                stats_ahead ? ctx->work : nullptr);
    }));
    CVHIP_TRY_HIP(hipGetLastError());
    return CVHIP_OK;
}

// Host images have been copied out of the caller's buffers when stage_images returns (page-locked ring), unless the ring
// could not be had: then the pageable sources must not be reused by the caller before the copy has happened.
int release_host_sources(cvhip_ctx *ctx, const uint8_t *img1, const uint8_t *img2)
{
    if (ctx->staged_from_pageable && (!is_device_ptr(img1) || !is_device_ptr(img2))) CVHIP_TRY_HIP(hipStreamSynchronize(ctx->dev->d.stream));
    return CVHIP_OK;
}
} // namespace

// What the fused-calls mode (cvhip_ctx_set_fuse_level_calls) has taken in but not yet executed: the forward search pass
// waits for the reverse call of its level, the forward cross-check for the reverse one.  Every other entry point of the
// context calls this first, so nothing ever observes the difference.
extern "C++" int cvhip::flush_level_calls(cvhip_ctx *ctx)
{
    cvhip_ctx::LevelCalls &lc = ctx->calls;
    const int stage = lc.stage;
    lc.stage = cvhip_ctx::LevelCalls::NONE;
    if (stage == cvhip_ctx::LevelCalls::FWD_TAKEN) {
        // the reverse call did not come (or not for the same images): the forward pass alone, as the per-pass call runs it
        CVHIP_TRY(set_device(ctx->dev));
        hipStream_t s = ctx->dev->d.stream;
        PassPlan plan;
        int rc = plan_pass(ctx, 0, 1, lc.w1, lc.h1, lc.w2, lc.h2, lc.scale, lc.k, lc.first_pass, 0, plan);
        if (rc == CVHIP_OK) rc = launch_passes(ctx, &plan, 1, true, s, lc.stats_done);
        else if (lc.stats_done) (void)hipStreamWaitEvent(s, lc.stats_done, 0);
        CVHIP_TRY(rc);
        commit_pass(ctx, plan);
        CVHIP_TRY(mark_level_read(ctx, lc.k, s));
    } else if (stage == cvhip_ctx::LevelCalls::CROSS_FWD_TAKEN) {
        CVHIP_TRY(set_device(ctx->dev));
        CVHIP_TRY(cross_check_pass(ctx, lc.k, 0));
    } else if (stage == cvhip_ctx::LevelCalls::HELD || stage == cvhip_ctx::LevelCalls::HELD_CROSS_FWD) {
        CVHIP_TRY(set_device(ctx->dev));
        LevelStats ls;
        ls.ahead = lc.stats_ahead;
        ls.done = lc.stats_done;
        CVHIP_TRY(level_search(ctx, false, lc.w1, lc.h1, lc.w2, lc.h2, lc.scale, lc.k, lc.first_pass, false, ls, nullptr, nullptr));
        if (stage == cvhip_ctx::LevelCalls::HELD_CROSS_FWD) CVHIP_TRY(cross_check_pass(ctx, lc.k, 0));
    }
    return CVHIP_OK;
}

int cvhip_correlate_images(cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2,
                           uint32_t w2, uint32_t h2, float scale, int first_pass, int dir,
                           cvhip_progress_fn progress, void *user)
{
    int k = 0;
    CVHIP_TRY(check_level_args(ctx, img1, w1, h1, img2, w2, h2, scale, &k));
    if (dir != 0 && dir != 1) return fail(CVHIP_ERR_INVALID, "dir must be 0 or 1");
    if (ctx->band_mode) {
        // the plan is laid out over the forward direction's pyramid: dir 1 is called with the images swapped
        const uint32_t fw = dir == 0 ? w1 : w2, fh = dir == 0 ? h1 : h2;
        if (k > ctx->band_steps || (ctx->w1 >> k) != fw || (ctx->h1 >> k) != fh)
            return fail(CVHIP_ERR_UNSUPPORTED, "band mode: level outside the planned pyramid");
    }
    CVHIP_TRY(set_device(ctx->dev));
    hipStream_t s = ctx->dev->d.stream;
    cvhip_ctx::LevelCalls &lc = ctx->calls;
    const bool fuse = ctx->fuse_level_calls && ctx->shard_den <= 1;
    if (fuse && dir == 1 && lc.stage == cvhip_ctx::LevelCalls::FWD_TAKEN && lc.k == k && lc.first_pass == first_pass && lc.img1 == img2 &&
        lc.img2 == img1 && lc.w1 == w2 && lc.h1 == h2 && lc.w2 == w1 && lc.h2 == h1) {
        // the reverse call of the level whose forward call was taken in: the images are staged (exchanged), their
        // statistics computed - both search passes go out together, as in cvhip_correlate_level
        if (k == 0 && !first_pass && ctx->result_bands != 1) {
            // result bands: the last level's launches interleave search and filter, so they wait for the two filter calls
            // (any other call first: flush_level_calls runs what was asked for so far, unbanded)
            lc.stage = cvhip_ctx::LevelCalls::HELD;
            report(progress, user, dir, 1.0f);
            return CVHIP_OK;
        }
        lc.stage = cvhip_ctx::LevelCalls::NONE;
        LevelStats ls;
        ls.ahead = lc.stats_ahead;
        ls.done = lc.stats_done;
        report(progress, user, dir, 0.20f);
        CVHIP_TRY(level_search(ctx, false, lc.w1, lc.h1, lc.w2, lc.h2, scale, k, first_pass, false, ls, nullptr, nullptr));
        lc.stage = cvhip_ctx::LevelCalls::SEARCHED;
        report(progress, user, dir, 1.0f);
        return CVHIP_OK;
    }
    CVHIP_TRY(flush_level_calls(ctx));
    LevelStats ls;
    if (fuse && dir == 0) {
        // the forward call of a level: images in, statistics - the search pass itself waits for the reverse call
        CVHIP_TRY(level_begin(ctx, img1, w1, h1, img2, w2, h2, k, first_pass, false, true, ls));
        lc.k = k;
        lc.first_pass = first_pass;
        lc.scale = scale;
        lc.img1 = img1;
        lc.img2 = img2;
        lc.w1 = w1;
        lc.h1 = h1;
        lc.w2 = w2;
        lc.h2 = h2;
        lc.stats_ahead = ls.ahead;
        lc.stats_done = ls.done;
        lc.stage = cvhip_ctx::LevelCalls::FWD_TAKEN;
        CVHIP_TRY(release_host_sources(ctx, img1, img2));
        report(progress, user, dir, 1.0f);
        return CVHIP_OK;
    }
    report(progress, user, dir, 0.02f);
    CVHIP_TRY(level_begin(ctx, img1, w1, h1, img2, w2, h2, k, first_pass, false, false, ls));
    report(progress, user, dir, 0.20f);
    {
        PassPlan plan;
        int rc = plan_pass(ctx, 0, 1, w1, h1, w2, h2, scale, k, first_pass, dir, plan);
        if (rc == CVHIP_OK) rc = launch_passes(ctx, &plan, 1, true, s, ls.done);
        else if (ls.done) (void)hipStreamWaitEvent(s, ls.done, 0);
        CVHIP_TRY(rc);
        commit_pass(ctx, plan);
        CVHIP_TRY(mark_level_read(ctx, k, s));
    }
    CVHIP_TRY(release_host_sources(ctx, img1, img2));
    report(progress, user, dir, 1.0f);
    return CVHIP_OK;
}

int cvhip_cross_check_filter(cvhip_ctx *ctx, float scale, int dir)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    if (dir != 0 && dir != 1) return fail(CVHIP_ERR_INVALID, "dir must be 0 or 1");
    const int k = scale_to_k(scale);
    if (k < 0) return fail(CVHIP_ERR_UNSUPPORTED, "scale must be 2^-k");
    CVHIP_TRY(set_device(ctx->dev));
    cvhip_ctx::LevelCalls &lc = ctx->calls;
    if (ctx->fuse_level_calls && lc.k == k) {
        if (dir == 0 && lc.stage == cvhip_ctx::LevelCalls::SEARCHED) { // waits for the reverse filter's call
            DirState &df = ctx->dir[0], &dr = ctx->dir[1];
            if (!df.valid || !dr.valid || (int)df.k != k || (int)dr.k != k)
                return fail(CVHIP_ERR_INVALID, "cross_check_filter scale does not match the grids' current level");
            lc.stage = cvhip_ctx::LevelCalls::CROSS_FWD_TAKEN;
            return CVHIP_OK;
        }
        if (dir == 1 && lc.stage == cvhip_ctx::LevelCalls::CROSS_FWD_TAKEN) { // both filters of the level in one launch
            lc.stage = cvhip_ctx::LevelCalls::NONE;
            return level_cross(ctx, k, lc.stats_ahead);
        }
        if (dir == 0 && lc.stage == cvhip_ctx::LevelCalls::HELD) {
            lc.stage = cvhip_ctx::LevelCalls::HELD_CROSS_FWD;
            return CVHIP_OK;
        }
        if (dir == 1 && lc.stage == cvhip_ctx::LevelCalls::HELD_CROSS_FWD) { // the whole last level, in result bands
            lc.stage = cvhip_ctx::LevelCalls::NONE;
            LevelStats ls;
            ls.ahead = lc.stats_ahead;
            ls.done = lc.stats_done;
            CVHIP_TRY(level_search(ctx, true, lc.w1, lc.h1, lc.w2, lc.h2, lc.scale, k, lc.first_pass, false, ls, nullptr, nullptr));
            return level_cross(ctx, k, lc.stats_ahead);
        }
    }
    CVHIP_TRY(flush_level_calls(ctx));
    return cross_check_pass(ctx, k, dir);
}

int cvhip_correlate_level(cvhip_ctx *ctx, const uint8_t *img1, uint32_t w1, uint32_t h1, const uint8_t *img2,
                          uint32_t w2, uint32_t h2, float scale, int first_pass, cvhip_progress_fn progress,
                          void *user)
{
    int k = 0;
    CVHIP_TRY(check_level_args(ctx, img1, w1, h1, img2, w2, h2, scale, &k));
    // Shard this level only if every rank gets a useful band; tiny levels are computed whole on
    // every rank (identical results, no collective).  The rule depends on level dims only, so all
    // ranks take the same branch.
    const uint32_t den = ctx->shard_den;
    const bool sharded = !ctx->band_mode && den > 1 && std::min(h1, h2) / den >= 64;
    if (sharded && !ctx->gather)
        return fail(CVHIP_ERR_INVALID, "row-sharded context without an all-gather hook (cvhip_ctx_set_row_shard)");
    if (ctx->band_mode && (k > ctx->band_steps || (ctx->w1 >> k) != w1 || (ctx->h1 >> k) != h1))
        return fail(CVHIP_ERR_UNSUPPORTED, "band mode: level outside the planned pyramid");
    CVHIP_TRY(set_device(ctx->dev));
    CVHIP_TRY(flush_level_calls(ctx));
    LevelStats ls;
    CVHIP_TRY(level_begin(ctx, img1, w1, h1, img2, w2, h2, k, first_pass, sharded, true, ls));
    report(progress, user, 0, 0.20f);
    CVHIP_TRY(level_search(ctx, true, w1, h1, w2, h2, scale, k, first_pass, sharded, ls, progress, user));
    CVHIP_TRY(level_cross(ctx, k, ls.ahead));
    CVHIP_TRY(release_host_sources(ctx, img1, img2));
    return CVHIP_OK;
}

// The staging sets of host-destination complete() calls (Device::Readback): room for n pixels, stream and events made.
static int readback_reserve(Device &d, size_t n)
{
    auto &rb = d.rb;
    CVHIP_TRY(copy_stream_reserve(d));
    if (n <= rb.cap_px) return CVHIP_OK;
    CVHIP_TRY_HIP(hipStreamSynchronize(rb.stream));
    for (int i = 0; i < 2; i++) {
        if (rb.xy[i]) (void)hipFree(rb.xy[i]);
        if (rb.corr[i]) (void)hipFree(rb.corr[i]);
        rb.xy[i] = nullptr;
        rb.corr[i] = nullptr;
        rb.pending[i] = false;
    }
    rb.cap_px = 0;
    for (int i = 0; i < 2; i++) {
        CVHIP_TRY_HIP(hipMalloc(&rb.xy[i], n * 2 * sizeof(int32_t)));
        CVHIP_TRY_HIP(hipMalloc(&rb.corr[i], n * sizeof(float)));
    }
    rb.cap_px = n;
    return CVHIP_OK;
}

// cvhip_complete_dir (words = 2: x, y as int32) and cvhip_complete_packed (words = 1: y << 16 | x)
static int complete_grid(cvhip_ctx *ctx, int dir, int32_t *out_xy, float *out_corr, const uint32_t words)
{
    const bool packed = words == 1;
    if (!ctx || !out_xy) return fail(CVHIP_ERR_INVALID, "null argument");
    if (dir != 0 && dir != 1) return fail(CVHIP_ERR_INVALID, "dir must be 0 or 1");
    CVHIP_TRY(set_device(ctx->dev));
    CVHIP_TRY(flush_level_calls(ctx));
    if (dir == 1) CVHIP_TRY(flush_reverse_cross_check(ctx));
    hipStream_t s = ctx->dev->d.stream;
    DirState &ds = ctx->dir[dir];
    const size_t n = (size_t)ds.gw * ds.gh;
    const bool xy_dev = is_device_ptr(out_xy);
    const bool corr_dev = out_corr ? is_device_ptr(out_corr) : true;
    const bool to_host = !xy_dev || (out_corr && !corr_dev);
    int32_t *d_xy = out_xy;
    float *d_corr = out_corr;
    auto &rb = ctx->dev->d.rb;
    int set = 0;
    if (to_host) {
        // Host destinations: the grid is expanded into one of two device staging sets owned by the handle and copied
        // out on the handle's copy stream; the set being refilled waits (on the device) for its previous copy.
        CVHIP_TRY(readback_reserve(ctx->dev->d, n));
        set = rb.next;
        rb.next ^= 1;
        if (rb.pending[set]) CVHIP_TRY_HIP(hipStreamWaitEvent(s, rb.done[set], 0));
        if (!xy_dev) d_xy = rb.xy[set];
        if (out_corr && !corr_dev) d_corr = rb.corr[set];
    }
    if (to_host && dir == 0 && ctx->live_bands > 1 && ds.valid && ds.k == 0 && ds.gh == ds.lh) {
        // The last level went out in result bands (level_search): each band is expanded and copied out on the copy stream
        // as soon as its forward filter is through, under the search of the bands behind it.
        for (uint32_t b = 0; b < ctx->live_bands; b++) {
            const uint32_t r0 = ctx->band_rows[b], r1 = ctx->band_rows[b + 1];
            const size_t o = (size_t)r0 * ds.gw, m = (size_t)(r1 - r0) * ds.gw;
            CVHIP_TRY_HIP(hipStreamWaitEvent(rb.stream, ctx->band_done[b], 0));
            launch_expand_grid(ds.cells[ds.cur], ds.scores_valid ? ds.scores : nullptr, ds.lw, ds.lh, 0, ds.gw, ds.gh, d_xy, d_corr,
                               rb.stream, r0, r1, packed);
            // The expansions read this level's match and score planes on the COPY stream: whatever the context's stream does
            // next (the next pair's levels rewrite both) must follow the last of them, host-synchronous call or not.
            if (b + 1 == ctx->live_bands) {
                CVHIP_TRY_HIP(hipEventRecord(rb.expanded, rb.stream));
                CVHIP_TRY_HIP(hipStreamWaitEvent(s, rb.expanded, 0));
            }
            if (!xy_dev)
                CVHIP_TRY_HIP(hipMemcpyAsync(out_xy + words * o, d_xy + words * o, m * words * sizeof(int32_t), hipMemcpyDeviceToHost, rb.stream));
            if (out_corr && !corr_dev)
                CVHIP_TRY_HIP(hipMemcpyAsync(out_corr + o, d_corr + o, m * sizeof(float), hipMemcpyDeviceToHost, rb.stream));
        }
        CVHIP_TRY_HIP(hipGetLastError());
        CVHIP_TRY_HIP(hipEventRecord(rb.done[set], rb.stream));
        rb.pending[set] = true;
        // (a device destination beside a host one was written on the copy stream: the context's stream follows it)
        if (xy_dev || (out_corr && corr_dev)) CVHIP_TRY_HIP(hipStreamWaitEvent(s, rb.done[set], 0));
        if (!ctx->async_readback) CVHIP_TRY_HIP(hipStreamSynchronize(rb.stream));
        return CVHIP_OK;
    }
    if (ds.valid) {
        (void)timed(ctx, cvhip_ctx::K_EXPAND,
                    [&] {
                        launch_expand_grid(ds.cells[ds.cur], ds.scores_valid ? ds.scores : nullptr, ds.lw, ds.lh, ds.k, ds.gw, ds.gh, d_xy,
                                           d_corr, s, 0, 0xFFFFFFFFu, packed);
                    });
    } else { // nothing computed: all None, like a fresh Grid (mod.rs:183-184)
        launch_fill_u32(reinterpret_cast<uint32_t *>(d_xy), 0xFFFFFFFFu, n * words, s);
        if (d_corr) launch_fill_u32(reinterpret_cast<uint32_t *>(d_corr), 0x7FC00000u, n, s);
    }
    CVHIP_TRY_HIP(hipGetLastError());
    if (to_host) {
        CVHIP_TRY_HIP(hipEventRecord(rb.ready, s));
        CVHIP_TRY_HIP(hipStreamWaitEvent(rb.stream, rb.ready, 0));
        if (!xy_dev) CVHIP_TRY_HIP(hipMemcpyAsync(out_xy, d_xy, n * words * sizeof(int32_t), hipMemcpyDeviceToHost, rb.stream));
        if (out_corr && !corr_dev)
            CVHIP_TRY_HIP(hipMemcpyAsync(out_corr, d_corr, n * sizeof(float), hipMemcpyDeviceToHost, rb.stream));
        CVHIP_TRY_HIP(hipEventRecord(rb.done[set], rb.stream));
        rb.pending[set] = true;
        // Host destinations are complete on return - unless the caller asked for asynchronous readback
        // (cvhip_ctx_set_async_readback: page-locked destinations, completion at cvhip_device_synchronize).  Device
        // destinations are written in stream order on the context's stream (no host synchronisation).
        if (!ctx->async_readback) CVHIP_TRY_HIP(hipStreamSynchronize(rb.stream));
    }
    return CVHIP_OK;
}

int cvhip_complete_dir(cvhip_ctx *ctx, int dir, int32_t *out_xy, float *out_corr)
{
    return complete_grid(ctx, dir, out_xy, out_corr, 2);
}

int cvhip_complete_packed(cvhip_ctx *ctx, int dir, uint32_t *out_cells, float *out_corr)
{
    return complete_grid(ctx, dir, reinterpret_cast<int32_t *>(out_cells), out_corr, 1);
}

int cvhip_complete(cvhip_ctx *ctx, int32_t *out_xy, float *out_corr)
{
    return cvhip_complete_dir(ctx, 0, out_xy, out_corr);
}

int cvhip_triangulate_affine(cvhip_ctx *ctx, double *out_points3d, uint32_t *out_p2, uint64_t cap, uint64_t *out_n)
{
    if (!ctx || !out_n) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(flush_level_calls(ctx));
    if (cap && !out_points3d) return fail(CVHIP_ERR_INVALID, "out_points3d is null");
    CVHIP_TRY(set_device(ctx->dev));
    hipStream_t s = ctx->dev->d.stream;
    DirState &ds = ctx->dir[0];
    *out_n = 0;
    if (!ds.valid) return CVHIP_OK; // nothing computed: no tracks
    const size_t n = (size_t)ds.gw * ds.gh;
    const uint32_t nblocks = (uint32_t)((n + 255) / 256);
    // the search-interval buffer is free between levels: reuse it for the block counts (+ total)
    if ((size_t)nblocks + 1 > ctx->max_px) return fail(CVHIP_ERR_INVALID, "image too small for the scratch buffer");
    uint32_t *counts = ctx->range, *total = ctx->range + nblocks;
    const bool p3_dev = out_points3d ? is_device_ptr(out_points3d) : true, p2_dev = out_p2 ? is_device_ptr(out_p2) : true;
    double *d_p3 = out_points3d;
    uint32_t *d_p2 = out_p2;
    hipError_t e = hipSuccess;
    if (cap && !p3_dev) e = hipMalloc(&d_p3, (size_t)cap * 3 * sizeof(double));
    if (e == hipSuccess && cap && out_p2 && !p2_dev) e = hipMalloc(&d_p2, (size_t)cap * 2 * sizeof(uint32_t));
    uint32_t h_total = 0;
    if (e == hipSuccess) {
        launch_triangulate_affine(ds.cells[ds.cur], ds.lw, ds.lh, ds.k, ds.gw, ds.gh, counts, total, d_p3, d_p2, cap, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_total, total, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    const uint64_t written = std::min<uint64_t>(h_total, cap);
    if (e == hipSuccess && cap && !p3_dev && written)
        e = hipMemcpy(out_points3d, d_p3, (size_t)written * 3 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && cap && out_p2 && !p2_dev && written)
        e = hipMemcpy(out_p2, d_p2, (size_t)written * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (cap && !p3_dev && d_p3) (void)hipFree(d_p3);
    if (cap && out_p2 && !p2_dev && d_p2) (void)hipFree(d_p2);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("triangulate_affine: ") + hipGetErrorString(e));
    *out_n = h_total;
    return CVHIP_OK;
}

int cvhip_ctx_set_row_shard(cvhip_ctx *ctx, uint32_t num, uint32_t den, cvhip_allgather_fn gather, void *user)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    if (den == 0 || den > 64 || num >= den) return fail(CVHIP_ERR_INVALID, "need 0 <= num < den <= 64");
    ctx->band_mode = false;
    ctx->shard_num = num;
    ctx->shard_den = den;
    ctx->gather = gather;
    ctx->gather_user = user;
    ctx->gather_on_stream = false; // a host hook; cvhip_ctx_set_row_shard_rccl sets it for the library's own collective
    return CVHIP_OK;
}

// Host-side copy of the kernels' epipolar line for a level pixel (get_epipolar_line, mod.rs:386-409).
static bool host_minor_offset_bound(const cvhip_ctx *c, int dir, int k, uint32_t lw1, uint32_t lh1, uint32_t lw2,
                                    uint32_t lh2, double *bound)
{
    double F[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) F[i * 3 + j] = dir == 0 ? c->F[i * 3 + j] : c->F[j * 3 + i];
    // the bound below relies on l = F*p having constant (x, y) components: the affine form
    if (F[0] != 0.0 || F[1] != 0.0 || F[3] != 0.0 || F[4] != 0.0) return false;
    const double scale = 1.0 / (double)(1u << k);
    double worst = 0.0;
    const uint32_t xs[2] = {KERNEL_SIZE, lw1 - KERNEL_SIZE - 1}, ys[2] = {KERNEL_SIZE, lh1 - KERNEL_SIZE - 1};
    for (uint32_t px : xs)
        for (uint32_t py : ys) {
            const double p0 = (double)px / scale, p1 = (double)py / scale;
            double f[3];
            for (int i = 0; i < 3; i++) f[i] = (F[i * 3 + 0] * p0 + F[i * 3 + 1] * p1) + F[i * 3 + 2];
            if (std::fabs(f[0]) > std::fabs(f[1])) return false; // corridor advances along y: not row-local
            const double cy = -f[0] / f[1], ay = -scale * f[2] / f[1];
            if (!std::isfinite(cy) || !std::isfinite(ay)) return false;
            const uint32_t ends[2] = {KERNEL_SIZE, lw2 - KERNEL_SIZE};
            for (uint32_t i : ends)
                for (int off : {-c->corridor_size, c->corridor_size})
                    worst = std::max(worst, std::fabs(std::floor((cy * (double)i + ay) + (double)off) - (double)py));
        }
    (void)lh2;
    *bound = worst + 1.0; // y2 - y is affine in (x, y, i) up to the floor: corners bound it
    return true;
}

int cvhip_ctx_set_row_band(cvhip_ctx *ctx, uint32_t num, uint32_t den)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    if (den == 0 || den > 64 || num >= den) return fail(CVHIP_ERR_INVALID, "need 0 <= num < den <= 64");
    ctx->band_mode = false;
    ctx->shard_num = num;
    ctx->shard_den = den;
    if (den == 1) return CVHIP_OK;
    // the reference's level schedule (optimal_scale_steps, mod.rs:542-550; reconstruction.rs:565-568)
    const uint32_t min_dim = std::min(ctx->w1, ctx->h1);
    const int steps = min_dim <= 64 ? 0 : (int)std::floor(std::log2((double)min_dim / 64.0));
    if (steps > 15) return fail(CVHIP_ERR_UNSUPPORTED, "band mode: too many levels");
    auto expand = [](const uint32_t (&r)[2], uint32_t m, uint32_t (&o)[2]) {
        o[0] = r[0] > m ? r[0] - m : 0u;
        o[1] = r[1] + m;
    };
    auto hull = [](const uint32_t (&a)[2], const uint32_t (&b)[2], uint32_t (&o)[2]) {
        const bool ea = a[1] <= a[0], eb = b[1] <= b[0];
        if (ea && eb) {
            o[0] = o[1] = 0;
        } else if (ea) {
            o[0] = b[0];
            o[1] = b[1];
        } else if (eb) {
            o[0] = a[0];
            o[1] = a[1];
        } else {
            o[0] = std::min(a[0], b[0]);
            o[1] = std::max(a[1], b[1]);
        }
    };
    cvhip_ctx::BandPlan *bp = ctx->band;
    shard_rows(ctx, ctx->h1, &bp[0].cf[0], &bp[0].cf[1]); // the band of the final forward grid
    bp[0].cr[0] = bp[0].cr[1] = 0;                        // the filtered reverse grid is never needed at level 0
    for (int k = 0; k <= steps; k++) {
        const uint32_t lw1 = ctx->w1 >> k, lh1 = ctx->h1 >> k, lw2 = ctx->w2 >> k, lh2 = ctx->h2 >> k;
        if (lw1 < KERNEL_WIDTH || lh1 < KERNEL_WIDTH || lw2 < KERNEL_WIDTH || lh2 < KERNEL_WIDTH)
            return fail(CVHIP_ERR_UNSUPPORTED, "band mode: level too small");
        double df = 0.0, dr = 0.0;
        if (!host_minor_offset_bound(ctx, 0, k, lw1, lh1, lw2, lh2, &df) ||
            !host_minor_offset_bound(ctx, 1, k, lw2, lh2, lw1, lh1, &dr))
            return fail(CVHIP_ERR_UNSUPPORTED,
                        "band mode needs row-local geometry (affine F with near-horizontal epipolar lines); "
                        "use cvhip_ctx_set_row_shard with an all-gather hook instead");
        const double dmax = std::max(df, dr);
        if (dmax > 64.0) return fail(CVHIP_ERR_UNSUPPORTED, "band mode: epipolar lines too steep for a row band");
        const uint32_t D = (uint32_t)dmax, E = D + CROSS_CHECK_SEARCH_AREA;
        uint32_t t[2];
        // a filtered cell needs its own search result and the other direction's unfiltered cells within
        // +-(D + 4) rows (mod.rs:588-624); a search result needs the filtered coarser cells (below)
        expand(bp[k].cr, E, t);
        if (bp[k].cr[1] <= bp[k].cr[0]) t[0] = t[1] = 0;
        hull(bp[k].cf, t, bp[k].sf);
        expand(bp[k].cf, E, t);
        if (bp[k].cf[1] <= bp[k].cf[0]) t[0] = t[1] = 0;
        hull(bp[k].cr, t, bp[k].sr);
        // window statistics: searched rows of either image plus the candidates' rows around them
        uint32_t u[2];
        hull(bp[k].sf, bp[k].sr, u);
        expand(u, D + 1, bp[k].st);
        if (k < steps) {
            // estimate_search_range of row y reads coarser rows [ceil((y-10)/2), ceil((y+10)/2)): +-6 (+1 slack)
            auto half = [](const uint32_t (&r)[2], uint32_t (&o)[2]) {
                o[0] = r[0] / 2;
                o[1] = (r[1] + 1) / 2;
            };
            uint32_t hsf[2], hsr[2];
            half(bp[k].sf, hsf);
            half(bp[k].sr, hsr);
            expand(hsf, 7, bp[k + 1].cf);
            expand(hsr, 7, bp[k + 1].cr);
            if (bp[k].sf[1] <= bp[k].sf[0]) bp[k + 1].cf[0] = bp[k + 1].cf[1] = 0;
            if (bp[k].sr[1] <= bp[k].sr[0]) bp[k + 1].cr[0] = bp[k + 1].cr[1] = 0;
        }
    }
    ctx->band_steps = steps;
    ctx->band_mode = true;
    return CVHIP_OK;
}

int cvhip_ctx_level_grid(cvhip_ctx *ctx, int dir, void **cells, void **scores, uint32_t *lw, uint32_t *lh, uint32_t *row0,
                         uint32_t *row1, uint32_t *rows_per_shard)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    if (dir != 0 && dir != 1) return fail(CVHIP_ERR_INVALID, "dir must be 0 or 1");
    DirState &ds = ctx->dir[dir];
    if (!ds.valid) return fail(CVHIP_ERR_INVALID, "no level computed yet");
    if (dir == 1) {
        CVHIP_TRY(set_device(ctx->dev));
        CVHIP_TRY(flush_reverse_cross_check(ctx));
    }
    if (cells) *cells = ds.cells[ds.cur];
    if (scores) *scores = ds.scores;
    if (lw) *lw = ds.lw;
    if (lh) *lh = ds.lh;
    uint32_t r0, r1;
    shard_rows(ctx, ds.lh, &r0, &r1);
    if (row0) *row0 = r0;
    if (row1) *row1 = r1;
    if (rows_per_shard) *rows_per_shard = (ds.lh + ctx->shard_den - 1) / ctx->shard_den;
    return CVHIP_OK;
}

int cvhip_ctx_set_profiling(cvhip_ctx *ctx, int time_kernels, int count_candidates)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->time_kernels = time_kernels == 2 ? 2 : (time_kernels ? 1 : 0);
    ctx->count_candidates = count_candidates ? 1 : 0;
    return CVHIP_OK;
}

int cvhip_ctx_get_profile(cvhip_ctx *ctx, uint32_t *launches, double *search_ms, uint64_t *candidates, int reset)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    CVHIP_TRY(set_device(ctx->dev));
    hipStream_t s = ctx->dev->d.stream;
    CVHIP_TRY_HIP(hipStreamSynchronize(s));
    CVHIP_TRY(resolve_events(ctx));
    unsigned long long cand = 0;
    CVHIP_TRY_HIP(hipMemcpy(&cand, ctx->d_cand, sizeof(cand), hipMemcpyDeviceToHost));
    // "search" here = the launches of the kernel that does the level's search: the box filter, or the candidate filter
    if (launches) *launches = ctx->prof_launches[cvhip_ctx::K_SEARCH] + ctx->prof_launches[cvhip_ctx::K_FILTER];
    if (search_ms) *search_ms = ctx->prof_ms[cvhip_ctx::K_SEARCH] + ctx->prof_ms[cvhip_ctx::K_FILTER];
    if (candidates) *candidates = (uint64_t)cand;
    if (reset) {
        for (int i = 0; i < cvhip_ctx::K_COUNT; i++) {
            ctx->prof_launches[i] = 0;
            ctx->prof_ms[i] = 0.0;
        }
        CVHIP_TRY_HIP(hipMemset(ctx->d_cand, 0, 4 * sizeof(unsigned long long)));
    }
    return CVHIP_OK;
}

int cvhip_ctx_get_kernel_times(cvhip_ctx *ctx, double ms[7], uint32_t launches[7], int reset)
{
    if (!ctx || !ms || !launches) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(flush_level_calls(ctx));
    CVHIP_TRY(set_device(ctx->dev));
    CVHIP_TRY_HIP(hipStreamSynchronize(ctx->dev->d.stream));
    CVHIP_TRY(resolve_events(ctx));
    for (int i = 0; i < cvhip_ctx::K_COUNT; i++) {
        ms[i] = ctx->prof_ms[i];
        launches[i] = ctx->prof_launches[i];
        if (reset) {
            ctx->prof_launches[i] = 0;
            ctx->prof_ms[i] = 0.0;
        }
    }
    return CVHIP_OK;
}

int cvhip_ctx_get_counters(cvhip_ctx *ctx, uint64_t out[4], int reset)
{
    if (!ctx || !out) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(flush_level_calls(ctx));
    CVHIP_TRY(set_device(ctx->dev));
    CVHIP_TRY_HIP(hipStreamSynchronize(ctx->dev->d.stream));
    unsigned long long v[4] = {0, 0, 0, 0};
    CVHIP_TRY_HIP(hipMemcpy(v, ctx->d_cand, sizeof(v), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out[i] = (uint64_t)v[i];
    if (reset) CVHIP_TRY_HIP(hipMemset(ctx->d_cand, 0, sizeof(v)));
    return CVHIP_OK;
}

int cvhip_ctx_set_stats_ahead(cvhip_ctx *ctx, int ahead)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->stats_ahead = ahead != 0;
    return CVHIP_OK;
}

int cvhip_ctx_set_borrow_inputs(cvhip_ctx *ctx, int borrow)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->borrow_inputs = borrow != 0;
    return CVHIP_OK;
}

int cvhip_ctx_set_fuse_level_calls(cvhip_ctx *ctx, int enable)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->fuse_level_calls = enable != 0;
    return CVHIP_OK;
}

int cvhip_ctx_set_result_bands(cvhip_ctx *ctx, uint32_t bands)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    if (bands > 16) return fail(CVHIP_ERR_INVALID, "result bands: 0 (the library's choice), 1 .. 16");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->result_bands = bands;
    return CVHIP_OK;
}

int cvhip_ctx_get_result_bands(cvhip_ctx *ctx, uint32_t *live)
{
    if (!ctx || !live) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(flush_level_calls(ctx));
    *live = ctx->live_bands > 1 ? ctx->live_bands : 1u;
    return CVHIP_OK;
}

int cvhip_ctx_set_async_readback(cvhip_ctx *ctx, int enable)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    ctx->async_readback = enable != 0;
    return CVHIP_OK;
}

int cvhip_ctx_set_exact_scores(cvhip_ctx *ctx, int all_passes)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    ctx->exact_scores = all_passes != 0;
    return CVHIP_OK;
}

int cvhip_ctx_set_range_mode(cvhip_ctx *ctx, int mode)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    if (mode < 0 || mode > 3) return fail(CVHIP_ERR_INVALID, "range mode must be 0..3");
    ctx->range_mode = mode;
    return CVHIP_OK;
}

int cvhip_ctx_set_search_version(cvhip_ctx *ctx, int version)
{
    if (!ctx) return fail(CVHIP_ERR_INVALID, "ctx is null");
    CVHIP_TRY(flush_level_calls(ctx));
    // 4 = version 3 with the box kernel launched for every geometry (tests: exercises its per-workgroup decline)
    if (version < 1 || version > 6) return fail(CVHIP_ERR_INVALID, "search version must be 1 .. 6");
    ctx->force_box = version == 4;
    ctx->search_version = version == 4 ? 3 : version;
    return CVHIP_OK;
}

} // extern "C"
