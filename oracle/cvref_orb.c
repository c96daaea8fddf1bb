/*
 * cvref_orb.c — CPU restatement of cybervision's ORB extraction (src/orb.rs) and of the
 * brute-force keypoint matcher (src/pointmatching.rs).
 * TEST INFRASTRUCTURE ONLY (see cvref.h).  Parity unpinned by the reference (no fixtures).
 * Keeps every quirk of the reference on purpose (see comments tagged QUIRK).
 */
#include "cvref.h"
#include "cvref_orb_pattern.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* orb.rs:11-41 */
static const signed char FAST_CIRCLE[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0},  {3, 1},   {2, 2},   {1, 3},
                                               {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};
#define FAST_KERNEL_SIZE 3
#define FAST_THRESHOLD 15
#define KEYPOINT_SCALE_MIN_SIZE 256
#define FAST_NUM_POINTS 9
#define FAST_CIRCLE_LENGTH (16 + FAST_NUM_POINTS - 1)
#define HARRIS_KERNEL_SIZE 3
#define HARRIS_KERNEL_WIDTH 7
#define HARRIS_K 0.04
#define ORB_GAUSS_KERNEL_WIDTH 11
#define ORB_PATCH_WIDTH 31
#define ORB_PATCH_SIZE 15
#define MAX_KEYPOINTS 10000

static inline size_t sat_add_signed(size_t a, ptrdiff_t b)
{
    if (b >= 0) {
        size_t r = a + (size_t)b;
        return r < a ? SIZE_MAX : r;
    }
    size_t nb = (size_t)(-b);
    return a > nb ? a - nb : 0;
}
static inline ptrdiff_t f64_to_isize(double v) /* `as isize`: saturating, NaN -> 0 */
{
    if (v != v) return 0;
    if (v >= 9223372036854775807.0) return PTRDIFF_MAX;
    if (v <= -9223372036854775808.0) return PTRDIFF_MIN;
    return (ptrdiff_t)v;
}
static inline size_t f64_to_usize(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)v;
}

/* adjust_contrast, orb.rs:455-472 */
void cvref_orb_adjust_contrast(uint8_t *img, uint32_t n)
{
    uint8_t min = 255, max = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (img[i] < min) min = img[i];
        if (img[i] > max) max = img[i];
    }
    if (min >= max) return;
    float coeff = 255.0f / (float)(max - min);
    for (uint32_t i = 0; i < n; i++) {
        float v = roundf(coeff * (float)(uint8_t)(img[i] - min));
        img[i] = v >= 255.0f ? 255 : (v > 0.0f ? (uint8_t)v : 0); /* `as u8` saturates */
    }
}

/* is_keypoint, orb.rs:424-453 (+ get_pixel_offset :417-422) */
static int is_keypoint(const uint8_t *img, uint32_t w, int threshold, size_t x, size_t y)
{
    int val = img[(size_t)w * y + x];
    int have_more = 0, have_less = 0;
    size_t last_more_pos = 0, last_less_pos = 0, max_length = 0;
    for (size_t i = 0; i < FAST_CIRCLE_LENGTH; i++) {
        const signed char *p = FAST_CIRCLE[i % 16];
        size_t xn = sat_add_signed(x, p[0]), yn = sat_add_signed(y, p[1]);
        int c_val = img[(size_t)w * yn + xn];
        if (c_val > val + threshold) {
            if (!have_more) {
                have_more = 1;
                last_more_pos = i;
            }
            size_t length = i - last_more_pos + 1;
            if (length > max_length) max_length = length;
        } else {
            have_more = 0;
        }
        if (c_val < val - threshold) {
            if (!have_less) {
                have_less = 1;
                last_less_pos = i;
            }
            size_t length = i - last_less_pos + 1;
            if (length > max_length) max_length = length;
        } else {
            have_less = 0;
        }
        if (max_length >= FAST_NUM_POINTS) return 1;
    }
    return 0;
}

typedef struct {
    uint32_t x, y;
} pt_t;

/* find_fast_keypoints, orb.rs:86-188. Returns malloc'd list (scan order, after NMS). */
static pt_t *find_fast_keypoints(const uint8_t *img, uint32_t w, uint32_t h, size_t *out_n, uint8_t **out_scores)
{
    size_t cap = 1024, n = 0;
    pt_t *kp = (pt_t *)malloc(cap * sizeof(pt_t));
    for (size_t y = FAST_KERNEL_SIZE; y < (size_t)h - FAST_KERNEL_SIZE; y++) {
        for (size_t x = FAST_KERNEL_SIZE; x < (size_t)w - FAST_KERNEL_SIZE; x++) {
            if (is_keypoint(img, w, FAST_THRESHOLD, x, y)) {
                if (n == cap) {
                    cap *= 2;
                    kp = (pt_t *)realloc(kp, cap * sizeof(pt_t));
                }
                kp[n].x = (uint32_t)x;
                kp[n].y = (uint32_t)y;
                n++;
            }
        }
    }
    /* scores by bisection, orb.rs:113-135 */
    uint8_t *scores = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        int threshold_min = FAST_THRESHOLD, threshold_max = 255;
        int threshold = (threshold_max + threshold_min) / 2;
        while (threshold_max > threshold_min + 1) {
            if (is_keypoint(img, w, threshold, kp[i].x, kp[i].y))
                threshold_min = threshold;
            else
                threshold_max = threshold;
            threshold = (threshold_min + threshold_max) / 2;
        }
        scores[i] = (uint8_t)threshold_min;
    }
    /* non-maximum suppression over the scan-ordered list, orb.rs:138-187 */
    pt_t *out = (pt_t *)malloc((n ? n : 1) * sizeof(pt_t));
    uint8_t *oscore = (uint8_t *)malloc(n ? n : 1);
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const pt_t p1 = kp[i];
        uint8_t score1 = scores[i];
        int drop = 0;
        if (i > 0 && kp[i - 1].x == p1.x - 1 && kp[i - 1].y == p1.y && scores[i - 1] >= score1) drop = 1;
        if (!drop && i < n - 1 && kp[i + 1].x == p1.x + 1 && kp[i + 1].y == p1.y && scores[i + 1] >= score1) drop = 1;
        if (!drop) {
            for (size_t j = i; j-- > 0;) {
                const pt_t p2 = kp[j];
                if (p2.y < p1.y - 1) break;
                if (p2.y == p1.y - 1 && p2.x >= p1.x - 1 && p2.x <= p1.x + 1 && scores[j] >= score1) {
                    drop = 1;
                    break;
                }
            }
        }
        if (!drop) {
            for (size_t j = i + 1; j < n; j++) {
                const pt_t p2 = kp[j];
                if (p2.y > p1.y + 1) break;
                if (p2.y == p1.y + 1 && p2.x >= p1.x - 1 && p2.x <= p1.x + 1 && scores[j] >= score1) {
                    drop = 1;
                    break;
                }
            }
        }
        if (!drop) {
            out[m] = p1;
            oscore[m] = score1;
            m++;
        }
    }
    free(kp);
    free(scores);
    *out_n = m;
    if (out_scores)
        *out_scores = oscore;
    else
        free(oscore);
    return out;
}

uint32_t cvref_orb_fast(const uint8_t *img, uint32_t w, uint32_t h, uint32_t cap, uint32_t *out_xy,
                        uint8_t *out_score)
{
    size_t n;
    uint8_t *sc;
    pt_t *kp = find_fast_keypoints(img, w, h, &n, &sc);
    uint32_t m = n < cap ? (uint32_t)n : cap;
    for (uint32_t i = 0; i < m; i++) {
        out_xy[2 * i] = kp[i].x;
        out_xy[2 * i + 1] = kp[i].y;
        if (out_score) out_score[i] = sc[i];
    }
    free(kp);
    free(sc);
    return (uint32_t)n;
}

/* gaussian_kernel, orb.rs:190-202 */
void cvref_orb_gaussian_kernel(uint32_t width, double *kernel)
{
    double sigma = (double)(width - 1) / 6.0;
    double sigma_2 = sigma * sigma; /* powi(2) */
    double divider = sqrt(2.0 * M_PI) * sigma;
    double center = (double)(width / 2);
    for (uint32_t i = 0; i < width; i++) {
        double d = (double)i - center;
        kernel[i] = exp(-(d * d) / (2.0 * sigma_2)) / divider;
    }
}

static const double KERNEL_SOBEL_X[9] = {-1.0, 0.0, 1.0, -2.0, 0.0, 2.0, -1.0, 0.0, 1.0};
static const double KERNEL_SOBEL_Y[9] = {-1.0, -2.0, -1.0, 0.0, 0.0, 0.0, 1.0, 2.0, 1.0};

/* convolve_kernel::<7, 9>, orb.rs:204-228.
 * QUIRK: the 9 Sobel taps are indexed with KERNEL_WIDTH=7, so k_x = i % 7, k_y = i / 7:
 * taps land on (x-3..x+3, y-3), (x-3, y-2), (x-2, y-2) — not on a 3x3 neighbourhood. */
static int convolve_kernel_7_9(const uint8_t *img, uint32_t w, uint32_t h, size_t x, size_t y, const double *kernel,
                               double *out)
{
    const size_t kernel_size = HARRIS_KERNEL_WIDTH / 2;
    if (x < kernel_size || y < kernel_size || x + kernel_size >= w || y + kernel_size >= h) return 0;
    double result = 0.0;
    for (size_t i = 0; i < 9; i++) {
        size_t k_x = i % HARRIS_KERNEL_WIDTH, k_y = i / HARRIS_KERNEL_WIDTH;
        result += kernel[i] * (double)img[(size_t)w * (y + k_y - kernel_size) + (x + k_x - kernel_size)] / 255.0;
    }
    *out = result;
    return 1;
}

/* harris_response::<7>, orb.rs:230-269 */
static int harris_response(const uint8_t *img, uint32_t w, uint32_t h, const double *kernel_gauss, size_t x,
                           size_t y, double *out)
{
    const size_t kernel_size = HARRIS_KERNEL_WIDTH / 2;
    if (x < kernel_size || y < kernel_size || x + kernel_size >= w || y + kernel_size >= h) return 0;
    double g_dx2 = 0.0, g_dy2 = 0.0, g_dx_dy = 0.0;
    for (size_t k_y = 0; k_y < HARRIS_KERNEL_WIDTH; k_y++) {
        for (size_t k_x = 0; k_x < HARRIS_KERNEL_WIDTH; k_x++) {
            size_t px = x + k_x - HARRIS_KERNEL_SIZE, py = y + k_y - HARRIS_KERNEL_SIZE;
            double dx, dy;
            if (!convolve_kernel_7_9(img, w, h, px, py, KERNEL_SOBEL_X, &dx)) return 0;
            if (!convolve_kernel_7_9(img, w, h, px, py, KERNEL_SOBEL_Y, &dy)) return 0;
            double gauss_mul = kernel_gauss[k_x] * kernel_gauss[k_y];
            g_dx2 += dx * dx * gauss_mul;
            g_dy2 += dy * dy * gauss_mul;
            g_dx_dy += dx * dy * gauss_mul;
        }
    }
    double det = g_dx2 * g_dy2 - g_dx_dy * g_dx_dy;
    double trace = g_dx2 + g_dy2;
    *out = det - HARRIS_K * (trace * trace);
    return 1;
}

int cvref_orb_harris(const uint8_t *img, uint32_t w, uint32_t h, uint32_t x, uint32_t y, double *out)
{
    double kg[HARRIS_KERNEL_WIDTH];
    cvref_orb_gaussian_kernel(HARRIS_KERNEL_WIDTH, kg);
    return harris_response(img, w, h, kg, x, y, out);
}

/* gaussian_blur::<11>, orb.rs:271-314.  None is encoded as NaN.
 * QUIRK: the second grid is allocated width x width (orb.rs:293). */
void cvref_orb_gaussian_blur(const uint8_t *img, uint32_t w, uint32_t h, double *out)
{
    double kg[ORB_GAUSS_KERNEL_WIDTH];
    cvref_orb_gaussian_kernel(ORB_GAUSS_KERNEL_WIDTH, kg);
    const size_t ks = ORB_GAUSS_KERNEL_WIDTH / 2;
    double *tmp = (double *)malloc((size_t)w * h * sizeof(double));
    for (size_t i = 0; i < (size_t)w * h; i++) tmp[i] = NAN;
    for (size_t y = 0; y < h; y++) {
        if (y < ks || y + ks >= h) continue;
        for (size_t x = 0; x < w; x++) {
            if (x < ks || x + ks >= w) continue;
            double sum = 0.0;
            for (size_t i = 0; i < ORB_GAUSS_KERNEL_WIDTH; i++) sum += kg[i] * (double)img[(size_t)w * y + (x + i - ks)];
            tmp[(size_t)w * y + x] = sum;
        }
    }
    size_t oh = w; /* QUIRK */
    for (size_t i = 0; i < (size_t)w * oh; i++) out[i] = NAN;
    for (size_t y = 0; y < oh; y++) {
        if (y < ks || y + ks >= h) continue;
        for (size_t x = 0; x < w; x++) {
            if (x < ks || x + ks >= w) continue;
            double sum = 0.0;
            int ok = 1;
            for (size_t i = 0; i < ORB_GAUSS_KERNEL_WIDTH; i++) {
                double val = tmp[(size_t)w * (y + i - ks) + x];
                if (val != val) {
                    ok = 0;
                    break;
                }
                sum += kg[i] * val;
            }
            if (ok) out[(size_t)w * y + x] = sum;
        }
    }
    free(tmp);
}

/* get_brief_orientation, orb.rs:316-344.  img = blurred grid (w x bh). Returns 0 for None. */
static int get_brief_orientation(const double *blur, uint32_t w, uint32_t bh, size_t x, size_t y, double *angle)
{
    if (x < ORB_PATCH_SIZE || y < ORB_PATCH_SIZE || x + ORB_PATCH_SIZE >= w || y + ORB_PATCH_SIZE >= bh) return 0;
    size_t m_00 = 0, m_01 = 0, m_10 = 0;
    for (size_t m_y = 0; m_y < ORB_PATCH_WIDTH; m_y++) {
        for (size_t m_x = 0; m_x < ORB_PATCH_WIDTH; m_x++) {
            size_t s_x = x + m_x - ORB_PATCH_SIZE, s_y = y + m_y - ORB_PATCH_SIZE;
            double v = blur[(size_t)w * s_y + s_x];
            if (v != v) return 0;
            double cl = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
            size_t val = f64_to_usize(cl);
            m_00 += val;
            m_10 += s_x * val;
            m_01 += s_y * val;
        }
    }
    double centroid_x = (double)m_10 / (double)m_00;
    double centroid_y = (double)m_01 / (double)m_00;
    *angle = atan2(centroid_y - (double)y, centroid_x - (double)x);
    return 1;
}

/* one keypoint of extract_brief_descriptors, orb.rs:363-402. Returns 0 for None. */
static int brief_descriptor(const double *blur, uint32_t w, uint32_t bh, size_t cx, size_t cy, uint32_t *desc)
{
    double angle;
    if (!get_brief_orientation(blur, w, bh, cx, cy, &angle)) return 0;
    double angle_sin = sin(angle), angle_cos = cos(angle);
    memset(desc, 0, 8 * sizeof(uint32_t));
    for (size_t i = 0; i < 256; i++) {
        double o1x = CVREF_ORB_PATTERN[4 * i + 0], o1y = CVREF_ORB_PATTERN[4 * i + 1];
        double o2x = CVREF_ORB_PATTERN[4 * i + 2], o2y = CVREF_ORB_PATTERN[4 * i + 3];
        /* QUIRK: x/y roles swapped relative to the textbook rotation (orb.rs:371-378) */
        ptrdiff_t off1x = f64_to_isize(round(o1y * angle_cos - o1x * angle_sin));
        ptrdiff_t off1y = f64_to_isize(round(o1y * angle_sin + o1x * angle_cos));
        ptrdiff_t off2x = f64_to_isize(round(o2y * angle_cos - o2x * angle_sin));
        ptrdiff_t off2y = f64_to_isize(round(o2y * angle_sin + o2x * angle_cos));
        size_t p1x = sat_add_signed(cx, off1x), p1y = sat_add_signed(cy, off1y);
        size_t p2x = sat_add_signed(cx, off2x), p2y = sat_add_signed(cy, off2y);
        if (p1x == 0 || p2x == 0 || p1x + 1 >= w || p2x + 1 >= w || p1y + 1 >= bh || p2y + 1 >= bh) return 0;
        double p1 = blur[(size_t)w * p1y + p1x];
        if (p1 != p1) return 0;
        double p2 = blur[(size_t)w * p2y + p2x];
        if (p2 != p2) return 0;
        uint32_t tau = p1 < p2 ? 1u : 0u;
        desc[i / 32] |= tau << (i % 32);
    }
    return 1;
}

typedef struct {
    pt_t p;
    double r;
    size_t idx;
} hk_t;

static int hk_cmp_desc(const void *a, const void *b)
{
    const hk_t *x = (const hk_t *)a, *y = (const hk_t *)b;
    if (x->r > y->r) return -1;
    if (x->r < y->r) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* stable (sort_by, orb.rs:76) */
}

/* extract_points, orb.rs:50-84 */
uint32_t cvref_orb_extract(const uint8_t *img, uint32_t w, uint32_t h, uint32_t cap, uint32_t *out_xy,
                           uint32_t *out_desc)
{
    size_t npx = (size_t)w * h;
    uint8_t *adj = (uint8_t *)malloc(npx);
    memcpy(adj, img, npx);
    cvref_orb_adjust_contrast(adj, (uint32_t)npx);
    size_t n;
    pt_t *kp = find_fast_keypoints(adj, w, h, &n, NULL);
    free(adj);

    double kg[HARRIS_KERNEL_WIDTH];
    cvref_orb_gaussian_kernel(HARRIS_KERNEL_WIDTH, kg);
    hk_t *hk = (hk_t *)malloc((n ? n : 1) * sizeof(hk_t));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        double r;
        if (harris_response(img, w, h, kg, kp[i].x, kp[i].y, &r)) { /* on the ORIGINAL image */
            hk[m].p = kp[i];
            hk[m].r = r;
            hk[m].idx = m;
            m++;
        }
    }
    free(kp);
    qsort(hk, m, sizeof(hk_t), hk_cmp_desc);
    if (m > MAX_KEYPOINTS) m = MAX_KEYPOINTS;

    double *blur = (double *)malloc((size_t)w * w * sizeof(double));
    cvref_orb_gaussian_blur(img, w, h, blur);
    uint32_t out_n = 0;
    for (size_t i = 0; i < m && out_n < cap; i++) {
        uint32_t desc[8];
        if (brief_descriptor(blur, w, w, hk[i].p.x, hk[i].p.y, desc)) {
            out_xy[2 * out_n] = hk[i].p.x;
            out_xy[2 * out_n + 1] = hk[i].p.y;
            memcpy(&out_desc[8 * out_n], desc, sizeof(desc));
            out_n++;
        }
    }
    free(blur);
    free(hk);
    return out_n;
}

/* optimal_scale_steps, orb.rs:407-415 */
uint32_t cvref_orb_optimal_scale_steps(uint32_t w, uint32_t h)
{
    size_t min_dimension = h < w ? h : w;
    if (min_dimension <= KEYPOINT_SCALE_MIN_SIZE) return 0;
    return (uint32_t)floor(log2((double)min_dimension / (double)KEYPOINT_SCALE_MIN_SIZE));
}

/* KeypointMatching::match_points, pointmatching.rs:43-77 */
typedef struct {
    uint32_t m[4];
    uint32_t dist;
    size_t idx;
} pm_t;
static int pm_cmp(const void *a, const void *b)
{
    const pm_t *x = (const pm_t *)a, *y = (const pm_t *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* sort_by_key is stable */
}
uint32_t cvref_match_points(const uint32_t *xy1, const uint32_t *desc1, uint32_t n1, const uint32_t *xy2,
                            const uint32_t *desc2, uint32_t n2, uint32_t threshold, uint32_t *out_matches,
                            uint32_t *out_dist)
{
    pm_t *pm = (pm_t *)malloc((n1 ? n1 : 1) * sizeof(pm_t));
    size_t m = 0;
    for (uint32_t i = 0; i < n1; i++) {
        int have = 0;
        uint32_t best = 0, bj = 0;
        for (uint32_t j = 0; j < n2; j++) {
            uint32_t distance = 0;
            for (int k = 0; k < 8; k++) distance += (uint32_t)__builtin_popcount(desc1[8 * i + k] ^ desc2[8 * j + k]);
            if (distance <= threshold && (!have || distance < best)) { /* min_by keeps the first minimum */
                have = 1;
                best = distance;
                bj = j;
            }
        }
        if (have) {
            pm[m].m[0] = xy1[2 * i];
            pm[m].m[1] = xy1[2 * i + 1];
            pm[m].m[2] = xy2[2 * bj];
            pm[m].m[3] = xy2[2 * bj + 1];
            pm[m].dist = best;
            pm[m].idx = m;
            m++;
        }
    }
    qsort(pm, m, sizeof(pm_t), pm_cmp);
    for (size_t i = 0; i < m; i++) {
        memcpy(&out_matches[4 * i], pm[i].m, sizeof(pm[i].m));
        if (out_dist) out_dist[i] = pm[i].dist;
    }
    free(pm);
    return (uint32_t)m;
}
