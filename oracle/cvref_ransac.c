/*
 * cvref_ransac.c — CPU restatement of the hypothesis-scoring part of cybervision's RANSAC
 * (src/fundamentalmatrix.rs).  TEST INFRASTRUCTURE ONLY (see cvref.h).
 * Parity unpinned by the reference (no fixtures).
 *
 * nalgebra 0.35.0 (Cargo.lock:649, not vendored) evaluation order, restated from the crate's
 * published algorithm: 3-vector `dot` is (a0*b0 + a1*b1) + a2*b2; matrix*vector is a
 * column-by-column axpy, i.e. ((M[i][0]*v0) + M[i][1]*v1) + M[i][2]*v2.
 */
#include "cvref.h"

#include <math.h>

/* reprojection_error, fundamentalmatrix.rs:461-471. F row-major. */
double cvref_reprojection_error(const double *F, const uint32_t *m)
{
    const double p1[3] = {(double)m[0], (double)m[1], 1.0};
    const double p2[3] = {(double)m[2], (double)m[3], 1.0};
    /* p2.tr_mul(f): row vector, element j = dot(p2, F[:, j]) */
    double p2t_f[3];
    for (int j = 0; j < 3; j++) p2t_f[j] = (p2[0] * F[0 * 3 + j] + p2[1] * F[1 * 3 + j]) + p2[2] * F[2 * 3 + j];
    /* (1x3) * p1 */
    double p2t_f_p1 = p2t_f[0] * p1[0];
    p2t_f_p1 = p2t_f[1] * p1[1] + p2t_f_p1;
    p2t_f_p1 = p2t_f[2] * p1[2] + p2t_f_p1;
    /* f * p1 */
    double f_p1[3];
    for (int i = 0; i < 3; i++) {
        double acc = F[i * 3 + 0] * p1[0];
        acc = F[i * 3 + 1] * p1[1] + acc;
        acc = F[i * 3 + 2] * p1[2] + acc;
        f_p1[i] = acc;
    }
    /* f.tr_mul(&p2): element i = dot(F[:, i], p2) */
    double ft_p2[3];
    for (int i = 0; i < 3; i++) ft_p2[i] = (F[0 * 3 + i] * p2[0] + F[1 * 3 + i] * p2[1]) + F[2 * 3 + i] * p2[2];
    double nominator = p2t_f_p1 * p2t_f_p1;
    double denominator = f_p1[0] * f_p1[0] + f_p1[1] * f_p1[1] + ft_p2[0] * ft_p2[0] + ft_p2[1] * ft_p2[1];
    return nominator / denominator;
}

/* fits_model (fundamentalmatrix.rs:452-458) folded over all matches as in validate_f
 * (fundamentalmatrix.rs:210-216): (count, serial error sum) per hypothesis. */
void cvref_ransac_score(const double *F, uint32_t H, const uint32_t *matches, uint32_t N, double t,
                        uint32_t *out_count, double *out_err_sum)
{
    for (uint32_t h = 0; h < H; h++) {
        const double *f = &F[(size_t)h * 9];
        uint32_t count = 0;
        double sum = 0.0;
        for (uint32_t i = 0; i < N; i++) {
            double err = cvref_reprojection_error(f, &matches[(size_t)i * 4]);
            if (!isfinite(err) || fabs(err) > t) continue;
            count += 1;
            sum += err;
        }
        out_count[h] = count;
        out_err_sum[h] = sum;
    }
}
