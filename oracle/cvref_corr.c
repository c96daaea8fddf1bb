/*
 * cvref_corr.c — CPU restatement of cybervision's dense correlation, `--mode=cpu`.
 * TEST INFRASTRUCTURE ONLY (see cvref.h).  Parity unpinned by the reference (no fixtures).
 *
 * Follows /root/reference/src/correlation/mod.rs line against line and deliberately keeps
 * the reference's algorithmic structure (full-resolution sparse Option<Match> grids, the
 * (20/scale)^2-cell neighbour scan, serial 121-term f32 sums with separate mul and add),
 * because it is also the "reference --mode=cpu" baseline timed by bench.py.
 *
 * Build with -ffp-contract=off: Rust never contracts a*b+c.
 */
#include "cvref.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* correlation/mod.rs:15-31 */
#define SCALE_MIN_SIZE 64
#define KERNEL_SIZE 5
#define KERNEL_WIDTH (KERNEL_SIZE * 2 + 1)
#define KERNEL_POINT_COUNT (KERNEL_WIDTH * KERNEL_WIDTH)
#define NEIGHBOR_DISTANCE 10
#define CROSS_CHECK_SEARCH_AREA 4

/* Option<(Point2D<u32>, f32)>, mod.rs:33 */
typedef struct {
    uint32_t x, y;
    float corr;
    uint32_t some;
} cell_t;

struct cvref_corr {
    uint32_t w1, h1, w2, h2;
    cell_t *grid[2]; /* [0] correlated_points (w1*h1), [1] correlated_points_reverse (w2*h2) */
    int first_pass;
    float min_stdev;
    size_t corridor_size;
    float correlation_threshold;
    double corridor_min_range;
    double corridor_extend_range;
    double F[9];
    int nthreads;
    _Atomic uint64_t candidates;
};

/* ---- Rust numeric-cast semantics ------------------------------------------------ */

static inline size_t f64_to_usize(double v) /* `as usize`: saturating, NaN -> 0 */
{
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)v;
}
static inline size_t f32_to_usize(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551615.0f) return SIZE_MAX;
    return (size_t)v;
}
static inline uint32_t f32_to_u32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)v;
}
static inline size_t sat_sub(size_t a, size_t b) { return a > b ? a - b : 0; }
static inline size_t sat_add(size_t a, size_t b) { return a + b < a ? SIZE_MAX : a + b; }
static inline size_t clampz(size_t v, size_t lo, size_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- tiny row-parallel helper (stands in for rayon; results schedule independent) - */

typedef void (*row_fn)(void *ctx, uint32_t y, int tid);
typedef struct {
    row_fn fn;
    void *ctx;
    uint32_t nrows;
    _Atomic uint32_t next;
    int tid_seq;
    pthread_mutex_t mu;
} par_t;

static void *par_worker(void *arg)
{
    par_t *p = (par_t *)arg;
    pthread_mutex_lock(&p->mu);
    int tid = p->tid_seq++;
    pthread_mutex_unlock(&p->mu);
    for (;;) {
        uint32_t y0 = atomic_fetch_add(&p->next, 4u);
        if (y0 >= p->nrows) break;
        uint32_t y1 = y0 + 4u < p->nrows ? y0 + 4u : p->nrows;
        for (uint32_t y = y0; y < y1; y++) p->fn(p->ctx, y, tid);
    }
    return NULL;
}

static void parallel_rows(uint32_t nrows, int nthreads, row_fn fn, void *ctx)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    par_t p;
    p.fn = fn;
    p.ctx = ctx;
    p.nrows = nrows;
    atomic_init(&p.next, 0);
    p.tid_seq = 0;
    pthread_mutex_init(&p.mu, NULL);
    if (nthreads == 1) {
        par_worker(&p);
    } else {
        pthread_t th[256];
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, par_worker, &p);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    pthread_mutex_destroy(&p.mu);
}

/* ---- mod.rs:632-735: window statistics -------------------------------------------- */

/* point_inside_bounds, mod.rs:697-699 */
static inline int point_inside_bounds(uint32_t w, uint32_t h, size_t x, size_t y)
{
    return x >= KERNEL_SIZE && y >= KERNEL_SIZE && x + KERNEL_SIZE < w && y + KERNEL_SIZE < h;
}

/* compute_point_avg, mod.rs:658-673 */
static int compute_point_avg(const uint8_t *img, uint32_t w, uint32_t h, size_t px, size_t py, float *out)
{
    if (!point_inside_bounds(w, h, px, py)) return 0;
    float avg = 0.0f;
    for (size_t y = 0; y < KERNEL_WIDTH; y++) {
        size_t s_y = sat_sub(py + y, KERNEL_SIZE);
        for (size_t x = 0; x < KERNEL_WIDTH; x++) {
            size_t s_x = sat_sub(px + x, KERNEL_SIZE);
            avg += (float)img[(size_t)w * s_y + s_x];
        }
    }
    avg /= (float)KERNEL_POINT_COUNT;
    *out = avg;
    return 1;
}

/* compute_point_stdev, mod.rs:676-694 */
static int compute_point_stdev(const uint8_t *img, uint32_t w, uint32_t h, size_t px, size_t py, float avg,
                               float *out)
{
    if (!point_inside_bounds(w, h, px, py)) return 0;
    float stdev = 0.0f;
    for (size_t y = 0; y < KERNEL_WIDTH; y++) {
        size_t s_y = sat_sub(py + y, KERNEL_SIZE);
        for (size_t x = 0; x < KERNEL_WIDTH; x++) {
            size_t s_x = sat_sub(px + x, KERNEL_SIZE);
            float delta = (float)img[(size_t)w * s_y + s_x] - avg;
            stdev += delta * delta;
        }
    }
    *out = sqrtf(stdev / (float)KERNEL_POINT_COUNT);
    return 1;
}

typedef struct {
    const uint8_t *img;
    uint32_t w, h;
    float *avg, *stdev;
} ipd_ctx;

static void ipd_avg_row(void *vctx, uint32_t y, int tid)
{
    (void)tid;
    ipd_ctx *c = (ipd_ctx *)vctx;
    for (uint32_t x = 0; x < c->w; x++) {
        float v;
        if (compute_point_avg(c->img, c->w, c->h, x, y, &v)) c->avg[(size_t)c->w * y + x] = v;
    }
}
static void ipd_stdev_row(void *vctx, uint32_t y, int tid)
{
    (void)tid;
    ipd_ctx *c = (ipd_ctx *)vctx;
    for (uint32_t x = 0; x < c->w; x++) {
        float v;
        if (compute_point_stdev(c->img, c->w, c->h, x, y, c->avg[(size_t)c->w * y + x], &v))
            c->stdev[(size_t)c->w * y + x] = v;
    }
}

/* compute_image_point_data, mod.rs:632-655: two passes, NaN where undefined */
void cvref_image_point_data(const uint8_t *img, uint32_t w, uint32_t h, float *avg, float *stdev, int nthreads)
{
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) {
        avg[i] = NAN;
        stdev[i] = NAN;
    }
    ipd_ctx c = {img, w, h, avg, stdev};
    parallel_rows(h, nthreads, ipd_avg_row, &c);
    parallel_rows(h, nthreads, ipd_stdev_row, &c);
}

/* PointData<121>, mod.rs:38-41 */
typedef struct {
    float delta[KERNEL_POINT_COUNT];
    float stdev;
} point_data;

/* compute_point_data::<5,121>, mod.rs:702-735 */
static int compute_point_data(const uint8_t *img, uint32_t w, uint32_t h, size_t px, size_t py, point_data *r)
{
    if (!point_inside_bounds(w, h, px, py)) return 0;
    r->stdev = 0.0f;
    float avg = 0.0f;
    for (size_t y = 0; y <= KERNEL_SIZE * 2; y++) {
        size_t s_y = sat_sub(py + y, KERNEL_SIZE);
        for (size_t x = 0; x <= KERNEL_SIZE * 2; x++) {
            size_t s_x = sat_sub(px + x, KERNEL_SIZE);
            uint8_t value = img[(size_t)w * s_y + s_x];
            r->delta[y * KERNEL_WIDTH + x] = (float)value;
            avg += (float)value;
        }
    }
    avg /= (float)KERNEL_POINT_COUNT;
    for (size_t i = 0; i < KERNEL_POINT_COUNT; i++) {
        float delta = r->delta[i] - avg;
        r->delta[i] = delta;
        r->stdev += delta * delta;
    }
    r->stdev = sqrtf(r->stdev / (float)KERNEL_POINT_COUNT);
    return 1;
}

/* ---- mod.rs:83-101 ---------------------------------------------------------------- */

typedef struct {
    double coeff_x, coeff_y;
    double add_x, add_y;
    ptrdiff_t off_x, off_y; /* corridor_offset */
} epipolar_line;

typedef struct {
    size_t has_pos;
    uint32_t pos_x, pos_y;
    int has_corr;
    float corr;
} best_match;

typedef struct {
    const cvref_corr *pc;
    float scale;
    double F[9];               /* fundamental_matrix for this direction */
    const cell_t *correlated;  /* full-res grid of this direction (read only during the pass) */
    uint32_t grid_w, grid_h;
    const uint8_t *img1, *img2;
    uint32_t w1, h1, w2, h2;   /* level dims */
    const float *avg2, *stdev2;
    cell_t *out_data;          /* w1*h1 level-sized */
    double **scratch;          /* per-thread STDEV_RANGE (mod.rs:477) */
    size_t *scratch_cap;
    _Atomic uint64_t cand;
} step_t;

/* get_epipolar_line, mod.rs:386-409.  F*p1 follows nalgebra 0.35 gemv: column-by-column
 * axpy, i.e. ((F[i][0]*p0) + F[i][1]*p1) + F[i][2]*p2 (Cargo.lock:649; source not vendored). */
static epipolar_line get_epipolar_line(const step_t *s, size_t px, size_t py)
{
    double scale = (double)s->scale;
    double p0 = (double)px / scale, p1 = (double)py / scale, p2 = 1.0;
    double f[3];
    for (int i = 0; i < 3; i++) {
#ifndef CVREF_ALT_ASSOC
        double acc = s->F[i * 3 + 0] * p0;
        acc = s->F[i * 3 + 1] * p1 + acc;
        acc = s->F[i * 3 + 2] * p2 + acc;
#else   /* sensitivity build (tests/test_oracle_corr.py): the other association of the three-term sum, to bound what
         * depends on the unverifiable nalgebra evaluation order */
        double acc = s->F[i * 3 + 0] * p0 + (s->F[i * 3 + 1] * p1 + s->F[i * 3 + 2] * p2);
#endif
        f[i] = acc;
    }
    epipolar_line e;
    if (fabs(f[0]) > fabs(f[1])) {
        e.coeff_x = -f[1] / f[0];
        e.coeff_y = 1.0;
        e.add_x = -scale * f[2] / f[0];
        e.add_y = 0.0;
        e.off_x = 1;
        e.off_y = 0;
        return e;
    }
    e.coeff_x = 1.0;
    e.coeff_y = -f[0] / f[1];
    e.add_x = 0.0;
    e.add_y = -scale * f[2] / f[1];
    e.off_x = 0;
    e.off_y = 1;
    return e;
}

/* correlate_corridor_area, mod.rs:411-466 */
static uint64_t correlate_corridor_area(const step_t *s, const epipolar_line *e, const point_data *p1,
                                        best_match *best, ptrdiff_t corridor_offset, size_t r0, size_t r1)
{
    const cvref_corr *pc = s->pc;
    float scale = s->scale;
    uint64_t evaluated = 0;
    for (size_t i = r0; i < r1; i++) {
        double x2d = (e->coeff_x * (double)i + e->add_x) + (double)(corridor_offset * e->off_x);
        double y2d = (e->coeff_y * (double)i + e->add_y) + (double)(corridor_offset * e->off_y);
        size_t x2 = f64_to_usize(floor(x2d));
        size_t y2 = f64_to_usize(floor(y2d));
        if (x2 < KERNEL_SIZE || x2 >= (size_t)s->w2 - KERNEL_SIZE || y2 < KERNEL_SIZE ||
            y2 >= (size_t)s->h2 - KERNEL_SIZE)
            continue;
        float avg2 = s->avg2[(size_t)s->w2 * y2 + x2];
        float stdev2 = s->stdev2[(size_t)s->w2 * y2 + x2];
        if (!isfinite(stdev2) || fabsf(stdev2) < pc->min_stdev) continue;
        evaluated++;
        float corr = 0.0f;
        for (size_t y = 0; y < KERNEL_WIDTH; y++) {
            for (size_t x = 0; x < KERNEL_WIDTH; x++) {
                float delta1 = p1->delta[y * KERNEL_WIDTH + x];
                float delta2 =
                    (float)s->img2[(size_t)s->w2 * sat_sub(y2 + y, KERNEL_SIZE) + sat_sub(x2 + x, KERNEL_SIZE)] -
                    avg2;
                corr += delta1 * delta2;
            }
        }
        corr /= p1->stdev * stdev2 * (float)KERNEL_POINT_COUNT;

        if (corr >= pc->correlation_threshold && (!best->has_corr || corr > best->corr)) {
            best->has_pos = 1;
            best->pos_x = f32_to_u32(roundf((float)x2 / scale));
            best->pos_y = f32_to_u32(roundf((float)y2 / scale));
            best->has_corr = 1;
            best->corr = corr;
        }
    }
    return evaluated;
}

/* estimate_search_range, mod.rs:468-540. Returns 0 for None. */
static int estimate_search_range(const step_t *s, size_t px, size_t py, const epipolar_line *e,
                                 size_t corridor_start, size_t corridor_end, size_t *out0, size_t *out1, int tid)
{
    const cvref_corr *pc = s->pc;
    float scale = s->scale;
    double mid_corridor = 0.0;
    size_t neighbor_count = 0;

    size_t x_min = f32_to_usize(floorf((float)sat_sub(px, NEIGHBOR_DISTANCE) / scale));
    size_t x_max = f32_to_usize(ceilf((float)(px + NEIGHBOR_DISTANCE) / scale));
    size_t y_min = f32_to_usize(floorf((float)sat_sub(py, NEIGHBOR_DISTANCE) / scale));
    size_t y_max = f32_to_usize(ceilf((float)(py + NEIGHBOR_DISTANCE) / scale));
    int corridor_vertical = fabs(e->coeff_y) > fabs(e->coeff_x);

    x_min = clampz(x_min, 0, s->grid_w);
    x_max = clampz(x_max, 0, s->grid_w);
    y_min = clampz(y_min, 0, s->grid_h);
    y_max = clampz(y_max, 0, s->grid_h);

    size_t need = (x_max - x_min) * (y_max - y_min);
    if (need > s->scratch_cap[tid]) {
        free(s->scratch[tid]);
        s->scratch[tid] = (double *)malloc(need * sizeof(double));
        s->scratch_cap[tid] = need;
    }
    double *stdev_range = s->scratch[tid];

    for (size_t y = y_min; y < y_max; y++) {
        for (size_t x = x_min; x < x_max; x++) {
            const cell_t *cp = &s->correlated[(size_t)s->grid_w * y + x];
            if (!cp->some) continue;
            double p2x = (double)scale * (double)cp->x;
            double p2y = (double)scale * (double)cp->y;
            double corridor_pos = corridor_vertical ? (p2y - e->add_y) / e->coeff_y : (p2x - e->add_x) / e->coeff_x;
            stdev_range[neighbor_count] = corridor_pos;
            neighbor_count += 1;
            mid_corridor += corridor_pos;
        }
    }
    if (neighbor_count == 0) return 0;

    mid_corridor /= (double)neighbor_count;
    double range_stdev = 0.0;
    for (size_t i = 0; i < neighbor_count; i++) {
        double delta = stdev_range[i] - mid_corridor;
        range_stdev += delta * delta;
    }
    range_stdev = sqrt(range_stdev / (double)neighbor_count);

    size_t corridor_center = f64_to_usize(round(mid_corridor));
    size_t corridor_length = f64_to_usize(round(pc->corridor_min_range + range_stdev * pc->corridor_extend_range));
    size_t new_start = clampz(sat_sub(corridor_center, corridor_length), corridor_start, corridor_end);
    size_t new_end = clampz(sat_add(corridor_center, corridor_length), new_start, corridor_end);
    *out0 = new_start;
    *out1 = new_end;
    return 1;
}

/* correlate_point, mod.rs:321-384 */
static void correlate_point(step_t *s, size_t px, size_t py, cell_t *out_point, int tid, uint64_t *cand)
{
    const cvref_corr *pc = s->pc;
    point_data p1;
    if (!compute_point_data(s->img1, s->w1, s->h1, px, py, &p1)) return;
    if (!isfinite(p1.stdev) || fabsf(p1.stdev) < pc->min_stdev) return;

    epipolar_line e = get_epipolar_line(s, px, py);
    if (!isfinite(e.coeff_x) || !isfinite(e.coeff_y) || !isfinite(e.add_x) || !isfinite(e.add_y)) return;

    const size_t CORRIDOR_START = KERNEL_SIZE;
    size_t corridor_end =
        fabs(e.coeff_x) > fabs(e.coeff_y) ? sat_sub(s->w2, KERNEL_SIZE) : sat_sub(s->h2, KERNEL_SIZE);
    size_t r0, r1;
    if (pc->first_pass) {
        r0 = CORRIDOR_START;
        r1 = corridor_end;
    } else if (!estimate_search_range(s, px, py, &e, CORRIDOR_START, corridor_end, &r0, &r1, tid)) {
        return;
    }

    best_match best = {0, 0, 0, 0, 0.0f};
    ptrdiff_t cs = (ptrdiff_t)pc->corridor_size;
    for (ptrdiff_t corridor_offset = -cs; corridor_offset <= cs; corridor_offset++)
        *cand += correlate_corridor_area(s, &e, &p1, &best, corridor_offset, r0, r1);
    if (best.has_pos && best.has_corr) {
        out_point->some = 1;
        out_point->x = best.pos_x;
        out_point->y = best.pos_y;
        out_point->corr = best.corr;
    }
}

static void step_row(void *vctx, uint32_t y, int tid)
{
    step_t *s = (step_t *)vctx;
    uint64_t cand = 0;
    size_t max_width = (size_t)s->w1 - KERNEL_SIZE, max_height = (size_t)s->h1 - KERNEL_SIZE;
    for (uint32_t x = 0; x < s->w1; x++) {
        /* mod.rs:299-301 */
        if (x < KERNEL_SIZE || y < KERNEL_SIZE || x >= max_width || y >= max_height) continue;
        correlate_point(s, x, y, &s->out_data[(size_t)s->w1 * y + x], tid, &cand);
    }
    atomic_fetch_add(&s->cand, cand);
}

/* ---- public ------------------------------------------------------------------------ */

cvref_corr *cvref_corr_new(uint32_t w1, uint32_t h1, uint32_t w2, uint32_t h2, const double *F, int projection,
                           int nthreads)
{
    cvref_corr *c = (cvref_corr *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->w1 = w1;
    c->h1 = h1;
    c->w2 = w2;
    c->h2 = h2;
    c->grid[0] = (cell_t *)calloc((size_t)w1 * h1, sizeof(cell_t));
    c->grid[1] = (cell_t *)calloc((size_t)w2 * h2, sizeof(cell_t));
    c->first_pass = 1;
    /* CorrelationParameters::for_projection, mod.rs:111-143 with constants mod.rs:20-30 */
    if (projection == 0) {
        c->min_stdev = 1.0f;
        c->correlation_threshold = 0.6f;
        c->corridor_size = 2;
        c->corridor_min_range = 2.5;
        c->corridor_extend_range = 1.0;
    } else {
        c->min_stdev = 1.0f;
        c->correlation_threshold = 0.5f;
        c->corridor_size = 4;
        c->corridor_min_range = 0.75;
        c->corridor_extend_range = 0.5;
    }
    memcpy(c->F, F, sizeof(c->F));
    c->nthreads = nthreads < 1 ? 1 : nthreads;
    atomic_init(&c->candidates, 0);
    return c;
}

void cvref_corr_free(cvref_corr *c)
{
    if (!c) return;
    free(c->grid[0]);
    free(c->grid[1]);
    free(c);
}

/* correlate_images_step, CPU branch, mod.rs:247-319 */
int cvref_corr_step(cvref_corr *c, const uint8_t *img1, uint32_t lw1, uint32_t lh1, const uint8_t *img2,
                    uint32_t lw2, uint32_t lh2, float scale, int dir)
{
    if (lw1 < KERNEL_WIDTH || lh1 < KERNEL_WIDTH || lw2 < KERNEL_WIDTH || lh2 < KERNEL_WIDTH) return -1;
    step_t s;
    memset(&s, 0, sizeof(s));
    s.pc = c;
    s.scale = scale;
    if (dir == 0) {
        memcpy(s.F, c->F, sizeof(s.F));
    } else { /* fundamental_matrix.transpose(), mod.rs:268-271 */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) s.F[i * 3 + j] = c->F[j * 3 + i];
    }
    s.correlated = c->grid[dir];
    s.grid_w = dir == 0 ? c->w1 : c->w2;
    s.grid_h = dir == 0 ? c->h1 : c->h2;
    s.img1 = img1;
    s.img2 = img2;
    s.w1 = lw1;
    s.h1 = lh1;
    s.w2 = lw2;
    s.h2 = lh2;
    size_t n2 = (size_t)lw2 * lh2, n1 = (size_t)lw1 * lh1;
    float *avg2 = (float *)malloc(n2 * sizeof(float));
    float *stdev2 = (float *)malloc(n2 * sizeof(float));
    cell_t *out_data = (cell_t *)calloc(n1, sizeof(cell_t));
    int nt = c->nthreads;
    double **scratch = (double **)calloc((size_t)nt, sizeof(double *));
    size_t *scratch_cap = (size_t *)calloc((size_t)nt, sizeof(size_t));
    if (!avg2 || !stdev2 || !out_data || !scratch || !scratch_cap) return -2;
    cvref_image_point_data(img2, lw2, lh2, avg2, stdev2, nt); /* mod.rs:260 */
    s.avg2 = avg2;
    s.stdev2 = stdev2;
    s.out_data = out_data;
    s.scratch = scratch;
    s.scratch_cap = scratch_cap;
    atomic_init(&s.cand, 0);

    parallel_rows(lh1, nt, step_row, &s); /* mod.rs:288-304 */
    atomic_fetch_add(&c->candidates, atomic_load(&s.cand));

    /* scatter, mod.rs:311-316 (serial in the reference too); overwrites including None */
    cell_t *grid = c->grid[dir];
    uint32_t gw = s.grid_w, gh = s.grid_h;
    int rc = 0;
    for (uint32_t y = 0; y < lh1 && rc == 0; y++) {
        for (uint32_t x = 0; x < lw1; x++) {
            size_t out_x = f32_to_usize((float)x / scale);
            size_t out_y = f32_to_usize((float)y / scale);
            if (out_x >= gw || out_y >= gh) { /* Grid::val_mut asserts, data.rs:61-64 */
                rc = -3;
                break;
            }
            grid[(size_t)gw * out_y + out_x] = out_data[(size_t)lw1 * y + x];
        }
    }
    for (int i = 0; i < nt; i++) free(scratch[i]);
    free(scratch);
    free(scratch_cap);
    free(out_data);
    free(avg2);
    free(stdev2);
    return rc;
}

typedef struct {
    cell_t *own;
    const cell_t *other;
    uint32_t ow, oh, rw, rh;
    size_t search_area;
} cc_ctx;

/* cross_check_point, mod.rs:588-624 */
static int cross_check_point(const cell_t *reverse, uint32_t rw, uint32_t rh, size_t search_area, size_t px,
                             size_t py, const cell_t *m)
{
    size_t min_x = clampz(sat_sub((size_t)m->x, search_area), 0, rw);
    size_t max_x = clampz(sat_add((size_t)m->x, search_area + 1), 0, rw);
    size_t min_y = clampz(sat_sub((size_t)m->y, search_area), 0, rh);
    size_t max_y = clampz(sat_add((size_t)m->y, search_area + 1), 0, rh);

    size_t r_min_x = sat_sub(px, search_area), r_max_x = sat_add(px, search_area + 1);
    size_t r_min_y = sat_sub(py, search_area), r_max_y = sat_add(py, search_area + 1);

    for (size_t s_y = min_y; s_y < max_y; s_y++) {
        for (size_t s_x = min_x; s_x < max_x; s_x++) {
            const cell_t *rm = &reverse[(size_t)rw * s_y + s_x];
            if (rm->some) {
                size_t r_x = rm->x, r_y = rm->y;
                if (r_x >= r_min_x && r_x < r_max_x && r_y >= r_min_y && r_y < r_max_y) return 1;
            }
        }
    }
    return 0;
}

static void cc_row(void *vctx, uint32_t y, int tid)
{
    (void)tid;
    cc_ctx *c = (cc_ctx *)vctx;
    for (uint32_t x = 0; x < c->ow; x++) {
        cell_t *cell = &c->own[(size_t)c->ow * y + x];
        if (cell->some && !cross_check_point(c->other, c->rw, c->rh, c->search_area, x, y, cell)) {
            cell->some = 0;
            cell->x = cell->y = 0;
            cell->corr = 0.0f;
        }
    }
}

/* cross_check_filter, CPU branch, mod.rs:552-586 */
int cvref_corr_cross_check(cvref_corr *c, float scale, int dir)
{
    cc_ctx cc;
    cc.own = c->grid[dir];
    cc.other = c->grid[1 - dir];
    cc.ow = dir == 0 ? c->w1 : c->w2;
    cc.oh = dir == 0 ? c->h1 : c->h2;
    cc.rw = dir == 0 ? c->w2 : c->w1;
    cc.rh = dir == 0 ? c->h2 : c->h1;
    cc.search_area = CROSS_CHECK_SEARCH_AREA * f32_to_usize(roundf(1.0f / scale));
    parallel_rows(cc.oh, c->nthreads, cc_row, &cc);
    return 0;
}

void cvref_corr_end_level(cvref_corr *c) { c->first_pass = 0; }

/* correlate_images, mod.rs:217-245 */
int cvref_corr_correlate_images(cvref_corr *c, const uint8_t *img1, uint32_t lw1, uint32_t lh1,
                                const uint8_t *img2, uint32_t lw2, uint32_t lh2, float scale)
{
    int rc = cvref_corr_step(c, img1, lw1, lh1, img2, lw2, lh2, scale, 0);
    if (rc) return rc;
    rc = cvref_corr_step(c, img2, lw2, lh2, img1, lw1, lh1, scale, 1);
    if (rc) return rc;
    cvref_corr_cross_check(c, scale, 0);
    cvref_corr_cross_check(c, scale, 1);
    cvref_corr_end_level(c);
    return 0;
}

void cvref_corr_get(const cvref_corr *c, int dir, int32_t *xy, float *corr)
{
    size_t n = dir == 0 ? (size_t)c->w1 * c->h1 : (size_t)c->w2 * c->h2;
    const cell_t *g = c->grid[dir];
    for (size_t i = 0; i < n; i++) {
        if (g[i].some) {
            xy[2 * i] = (int32_t)g[i].x;
            xy[2 * i + 1] = (int32_t)g[i].y;
            corr[i] = g[i].corr;
        } else {
            xy[2 * i] = -1;
            xy[2 * i + 1] = -1;
            corr[i] = NAN;
        }
    }
}

uint64_t cvref_corr_candidates(const cvref_corr *c) { return atomic_load(&c->candidates); }

/* optimal_scale_steps, mod.rs:542-550 */
uint32_t cvref_corr_optimal_scale_steps(uint32_t w, uint32_t h)
{
    size_t min_dimension = h < w ? h : w;
    if (min_dimension <= SCALE_MIN_SIZE) return 0;
    return (uint32_t)floor(log2((double)min_dimension / (double)SCALE_MIN_SIZE));
}

/* AffineTriangulation::triangulate + triangulate_point, triangulation.rs:268-330 */
uint64_t cvref_triangulate_affine(const int32_t *xy, uint32_t w, uint32_t h, double *out_points3d, uint32_t *out_p2)
{
    uint64_t n = 0;
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) {
            const int32_t *m = &xy[2 * ((size_t)w * y + x)];
            if (m[0] < 0) continue;
            double dx = (double)x - (double)(uint32_t)m[0];
            double dy = (double)y - (double)(uint32_t)m[1];
            double distance = sqrt(dx * dx + dy * dy);
            out_points3d[3 * n + 0] = (double)x;
            out_points3d[3 * n + 1] = (double)y;
            out_points3d[3 * n + 2] = distance;
            if (out_p2) {
                out_p2[2 * n + 0] = (uint32_t)m[0];
                out_p2[2 * n + 1] = (uint32_t)m[1];
            }
            n++;
        }
    }
    return n;
}

/* Triangulation::extend_tracks, triangulation.rs:1330-1419.
 * track_p1: the existing tracks' points in image 1 (2 int32 each, (-1,-1) = the track has none: `track.get(image1_index)?`).
 * out_track_p2[i]: the point `track.add(image2_index, ..)` is called with for track i, or (-1,-1) where the closure
 * returns None.  Then every merged point clears remaining_points at ITS OWN (image 2) coordinates (:1391-1393, the
 * reference indexes the image-1-sized grid with them; out of bounds panics there: return UINT64_MAX), and every
 * remaining Some cell becomes a new track (p1 = the cell, p2 = its match), in scan order (:1397-1416).
 * Returns the number of new tracks. */
uint64_t cvref_extend_tracks(const int32_t *xy, uint32_t w, uint32_t h, const int32_t *track_p1, uint64_t n_tracks,
                             uint32_t max_dimension2, int32_t *out_track_p2, uint32_t *out_new_p1, uint32_t *out_new_p2)
{
    /* :1346-1350, EXTEND_TRACKS_SEARCH_RADIUS = 3, TRACKS_RADIUS_DENOMINATOR = 1000 (:16, :19) */
    const size_t search_radius = max_dimension2 > 1000 ? (size_t)3 * max_dimension2 / 1000 : 3;
    uint8_t *removed = (uint8_t *)calloc((size_t)w * h, 1);
    if (!removed) return UINT64_MAX;
    for (uint64_t t = 0; t < n_tracks; t++) {
        out_track_p2[2 * t] = out_track_p2[2 * t + 1] = -1;
        if (track_p1[2 * t] < 0) continue;
        const size_t px = (size_t)track_p1[2 * t], py = (size_t)track_p1[2 * t + 1];
        const size_t min_x = px > search_radius ? px - search_radius : 0, min_y = py > search_radius ? py - search_radius : 0;
        const size_t max_x = px + search_radius < w ? px + search_radius : w, max_y = py + search_radius < h ? py + search_radius : h;
        int have = 0;
        size_t min_distance = 0;
        int32_t bx = -1, by = -1;
        for (size_t y = min_y; y < max_y; y++) {
            for (size_t x = min_x; x < max_x; x++) {
                const int32_t *m = &xy[2 * ((size_t)w * y + x)];
                if (m[0] < 0) continue;
                const size_t dx = (x > px ? x : px) - (x < px ? x : px), dy = (y > py ? y : py) - (y < py ? y : py);
                const size_t distance = dx * dx + dy * dy;
                if (!have || distance < min_distance) {
                    have = 1;
                    min_distance = distance;
                    bx = m[0];
                    by = m[1];
                }
            }
        }
        if (!have) continue;
        out_track_p2[2 * t] = bx;
        out_track_p2[2 * t + 1] = by;
    }
    for (uint64_t t = 0; t < n_tracks; t++) {
        if (out_track_p2[2 * t] < 0) continue;
        const size_t x = (size_t)out_track_p2[2 * t], y = (size_t)out_track_p2[2 * t + 1];
        if (x >= w || y >= h) { /* Grid::val_mut asserts (data.rs:61-64) */
            free(removed);
            return UINT64_MAX;
        }
        removed[(size_t)w * y + x] = 1;
    }
    uint64_t n = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const size_t i = (size_t)w * y + x;
            if (xy[2 * i] < 0 || removed[i]) continue;
            out_new_p1[2 * n] = x;
            out_new_p1[2 * n + 1] = y;
            out_new_p2[2 * n] = (uint32_t)xy[2 * i];
            out_new_p2[2 * n + 1] = (uint32_t)xy[2 * i + 1];
            n++;
        }
    free(removed);
    return n;
}
