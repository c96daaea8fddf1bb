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
#include <stdlib.h>

/* reprojection_error, fundamentalmatrix.rs:461-471. F row-major. */
double cvref_reprojection_error(const double *F, const uint32_t *m)
{
    const double p1[3] = {(double)m[0], (double)m[1], 1.0};
    const double p2[3] = {(double)m[2], (double)m[3], 1.0};
    /* p2.tr_mul(f): row vector, element j = dot(p2, F[:, j]) */
    double p2t_f[3];
#ifndef CVREF_ALT_ASSOC
    for (int j = 0; j < 3; j++) p2t_f[j] = (p2[0] * F[0 * 3 + j] + p2[1] * F[1 * 3 + j]) + p2[2] * F[2 * 3 + j];
#else   /* sensitivity build: the other association everywhere in this function */
    for (int j = 0; j < 3; j++) p2t_f[j] = p2[0] * F[0 * 3 + j] + (p2[1] * F[1 * 3 + j] + p2[2] * F[2 * 3 + j]);
#endif
    /* (1x3) * p1 */
#ifndef CVREF_ALT_ASSOC
    double p2t_f_p1 = p2t_f[0] * p1[0];
    p2t_f_p1 = p2t_f[1] * p1[1] + p2t_f_p1;
    p2t_f_p1 = p2t_f[2] * p1[2] + p2t_f_p1;
#else
    double p2t_f_p1 = p2t_f[0] * p1[0] + (p2t_f[1] * p1[1] + p2t_f[2] * p1[2]);
#endif
    /* f * p1 */
    double f_p1[3];
    for (int i = 0; i < 3; i++) {
#ifndef CVREF_ALT_ASSOC
        double acc = F[i * 3 + 0] * p1[0];
        acc = F[i * 3 + 1] * p1[1] + acc;
        acc = F[i * 3 + 2] * p1[2] + acc;
#else
        double acc = F[i * 3 + 0] * p1[0] + (F[i * 3 + 1] * p1[1] + F[i * 3 + 2] * p1[2]);
#endif
        f_p1[i] = acc;
    }
    /* f.tr_mul(&p2): element i = dot(F[:, i], p2) */
    double ft_p2[3];
#ifndef CVREF_ALT_ASSOC
    for (int i = 0; i < 3; i++) ft_p2[i] = (F[0 * 3 + i] * p2[0] + F[1 * 3 + i] * p2[1]) + F[2 * 3 + i] * p2[2];
#else
    for (int i = 0; i < 3; i++) ft_p2[i] = F[0 * 3 + i] * p2[0] + (F[1 * 3 + i] * p2[1] + F[2 * 3 + i] * p2[2]);
#endif
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

/* ------------------------------------------------------------------------------------------------
 * optimize_perspective_f (fundamentalmatrix.rs:391-426): the reference's own Levenberg-Marquardt on the
 * 7 free parameters of F, restated operation by operation - including what it does not do the textbook
 * way: the Jacobian (:473-512) uses c = d = f_p1[0] + f_p1[1] + ft_p2[0] + ft_p2[1] (sums, not sums of
 * squares), the step is params + delta with delta = (J'J + mu I)^-1 J'r, and "converged" compares the
 * residual reduction with 0.  nalgebra pieces restated from the crate's published code: `dot` of long
 * vectors keeps 8 accumulators (blas.rs: blocks of 8, then (0+4) (1+5) (2+6) (3+7), then the tail
 * serially), LU is partial pivoting with multipliers formed by multiplication with the reciprocal pivot,
 * solve = permute, unit-lower forward substitution by columns, upper back substitution by columns.
 * ---------------------------------------------------------------------------------------------- */
static double dot_n(const double *a, size_t sa, const double *b, size_t sb, uint32_t n)
{
    double res = 0.0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t i = 0;
    while (n - i >= 8) {
        for (int k = 0; k < 8; k++) acc[k] += a[(size_t)(i + k) * sa] * b[(size_t)(i + k) * sb];
        i += 8;
    }
    res += acc[0] + acc[4];
    res += acc[1] + acc[5];
    res += acc[2] + acc[6];
    res += acc[3] + acc[7];
    for (; i < n; i++) res += a[(size_t)i * sa] * b[(size_t)i * sb];
    return res;
}

/* f_from_perspective_params, :442-449 */
static void lm_f_from_params(const double *p, double *F)
{
    const double x = -(-p[0] * p[4] + p[6] * p[2] * p[4] + p[3] * p[1] - p[6] * p[1] * p[5]) / (-p[3] * p[2] + p[0] * p[5]);
    F[0] = p[0], F[1] = p[1], F[2] = p[2], F[3] = p[3], F[4] = p[4], F[5] = p[5], F[6] = p[6], F[7] = x, F[8] = 1.0;
}

/* f_jacobian, :473-512 */
static void lm_jacobian_row(const double *F, const uint32_t *m, double *out)
{
    const double p1[3] = {(double)m[0], (double)m[1], 1.0};
    const double p2[3] = {(double)m[2], (double)m[3], 1.0};
    double f_p1[3], ft_p2[3];
    for (int i = 0; i < 3; i++) {
#ifndef CVREF_ALT_ASSOC
        double acc = F[i * 3 + 0] * p1[0];
        acc = F[i * 3 + 1] * p1[1] + acc;
        acc = F[i * 3 + 2] * p1[2] + acc;
#else
        double acc = F[i * 3 + 0] * p1[0] + (F[i * 3 + 1] * p1[1] + F[i * 3 + 2] * p1[2]);
#endif
        f_p1[i] = acc;
        ft_p2[i] = (F[0 * 3 + i] * p2[0] + F[1 * 3 + i] * p2[1]) + F[2 * 3 + i] * p2[2];
    }
    const double c = f_p1[0] + f_p1[1] + ft_p2[0] + ft_p2[1], d = c;
    for (int i = 0; i < 7; i++) {
        const int row = i / 3, col = i % 3;
        const double a = p2[row] * p1[col]; /* p2' * (unit mask) * p1: every other term is an exact zero */
        double Fm[9];
        for (int k = 0; k < 9; k++) Fm[k] = F[k];
        Fm[row * 3 + col] = 0.0;
        double r[3];
        for (int j = 0; j < 3; j++) r[j] = (p2[0] * Fm[0 * 3 + j] + p2[1] * Fm[1 * 3 + j]) + p2[2] * Fm[2 * 3 + j];
        double b = r[0] * p1[0];
        b = r[1] * p1[1] + b;
        b = r[2] * p1[2] + b;
        const double x = F[row * 3 + col];
        out[i] = 2.0 * (a * x + b) * (a * d - b * c * c * x) / (c * c * x * x + d);
    }
}

/* nalgebra LU (partial pivoting) + solve for a 7x7 system; A row-major, destroyed.  0 = singular. */
static int lm_lu_solve7(double *A, double *b)
{
    enum { N = 7 };
    int perm_a[N], perm_b[N], np = 0;
    for (int i = 0; i < N; i++) {
        int piv = i;
        double best = fabs(A[i * N + i]);
        for (int r = i + 1; r < N; r++)
            if (fabs(A[r * N + i]) > best) best = fabs(A[r * N + i]), piv = r; /* icamax: first maximum */
        const double diag = A[piv * N + i];
        if (diag == 0.0) continue;
        if (piv != i) {
            perm_a[np] = i, perm_b[np] = piv, np++;
            for (int k = 0; k < N; k++) {
                const double t = A[i * N + k];
                A[i * N + k] = A[piv * N + k];
                A[piv * N + k] = t;
            }
        }
        const double inv = 1.0 / diag;
        for (int r = i + 1; r < N; r++) A[r * N + i] *= inv;
        for (int k = i + 1; k < N; k++) {
            const double pk = A[i * N + k];
            for (int r = i + 1; r < N; r++) A[r * N + k] = -pk * A[r * N + i] + A[r * N + k];
        }
    }
    for (int k = 0; k < np; k++) {
        const double t = b[perm_a[k]];
        b[perm_a[k]] = b[perm_b[k]];
        b[perm_b[k]] = t;
    }
    for (int i = 0; i < N; i++) {
        const double coeff = b[i];
        for (int r = i + 1; r < N; r++) b[r] = -coeff * A[r * N + i] + b[r];
    }
    for (int i = N - 1; i >= 0; i--) {
        const double diag = A[i * N + i];
        if (diag == 0.0) return 0;
        const double coeff = b[i] / diag;
        b[i] = coeff;
        for (int r = 0; r < i; r++) b[r] = -coeff * A[r * N + i] + b[r];
    }
    return 1;
}

static double lm_max7(const double *v)
{
    double m = v[0];
    for (int i = 1; i < 7; i++)
        if (v[i] > m) m = v[i];
    return m;
}

/* singular values of a 3x3 matrix (descending) through the eigenvalues of M'M (cyclic Jacobi); only compared
 * with 1e-3 (:418-423) */
static void lm_singular_values3(const double *M, double *s)
{
    double a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) a[i][j] = M[0 * 3 + i] * M[0 * 3 + j] + M[1 * 3 + i] * M[1 * 3 + j] + M[2 * 3 + i] * M[2 * 3 + j];
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
            }
    }
    double e[3] = {a[0][0], a[1][1], a[2][2]};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (e[j] > e[i]) {
                const double t = e[i];
                e[i] = e[j];
                e[j] = t;
            }
    for (int i = 0; i < 3; i++) s[i] = sqrt(e[i] > 0.0 ? e[i] : 0.0);
}

/* least_squares (:515-621) specialised to this problem.  1 = Ok(params), 0 = Err. */
static int lm_least_squares(double *params, const uint32_t *matches, uint32_t n, double *res, double *newres, double *jac)
{
    const double TAU = 1E-3, GRADIENT_EPSILON = 1E-12, DELTA_EPSILON = 1E-12, RESIDUAL_EPSILON = 1E-12,
                 RESIDUAL_REDUCTION_EPSILON = 0.0;
    double F[9], jtr[7];
    lm_f_from_params(params, F);
    for (uint32_t i = 0; i < n; i++) res[i] = cvref_reprojection_error(F, &matches[(size_t)i * 4]);
    for (uint32_t i = 0; i < n; i++) lm_jacobian_row(F, &matches[(size_t)i * 4], &jac[(size_t)i * 7]);
    for (int j = 0; j < 7; j++) jtr[j] = dot_n(&jac[j], 7, res, 1, n);
    if (fabs(lm_max7(jtr)) <= GRADIENT_EPSILON) return 1;
    double mu;
    {
        double best = dot_n(&jac[0], 7, &jac[0], 7, n);
        for (int j = 1; j < 7; j++) {
            const double v = dot_n(&jac[j], 7, &jac[j], 7, n);
            if (v >= best) best = v; /* Iterator::max_by keeps the last of equal maxima */
        }
        mu = TAU * best;
    }
    double nu = 2.0;
    int found = 0;
    for (int it = 0; it < 1000; it++) {
        double A[49], delta[7];
        for (int i = 0; i < 7; i++)
            for (int j = 0; j < 7; j++) A[i * 7 + j] = dot_n(&jac[i], 7, &jac[j], 7, n);
        for (int i = 0; i < 7; i++) A[i * 7 + i] += mu;
        for (int j = 0; j < 7; j++) delta[j] = jtr[j];
        if (!lm_lu_solve7(A, delta)) return 0; /* "Failed to compute delta vector" */
        if (sqrt(dot_n(delta, 1, delta, 1, 7)) <= DELTA_EPSILON * (sqrt(dot_n(params, 1, params, 1, 7)) + DELTA_EPSILON)) {
            found = 1;
            break;
        }
        double np[7];
        for (int j = 0; j < 7; j++) np[j] = params[j] + delta[j];
        lm_f_from_params(np, F);
        for (uint32_t i = 0; i < n; i++) newres[i] = cvref_reprojection_error(F, &matches[(size_t)i * 4]);
        const double r2 = dot_n(res, 1, res, 1, n), nr2 = dot_n(newres, 1, newres, 1, n);
        double tmp[7];
        for (int j = 0; j < 7; j++) tmp[j] = delta[j] * mu + jtr[j];
        const double rho = (r2 - nr2) / dot_n(delta, 1, tmp, 1, 7);
        if (rho > 0.0) {
            const int converged = sqrt(r2) - sqrt(nr2) < RESIDUAL_REDUCTION_EPSILON * sqrt(r2);
            for (uint32_t i = 0; i < n; i++) res[i] = newres[i];
            for (int j = 0; j < 7; j++) params[j] = np[j];
            for (uint32_t i = 0; i < n; i++) lm_jacobian_row(F, &matches[(size_t)i * 4], &jac[(size_t)i * 7]);
            for (int j = 0; j < 7; j++) jtr[j] = dot_n(&jac[j], 7, res, 1, n);
            if (converged || fabs(lm_max7(jtr)) <= GRADIENT_EPSILON) {
                found = 1;
                break;
            }
            const double q = 2.0 * rho - 1.0, alt = 1.0 - q * q * q;
            mu *= (alt > 1.0 / 3.0 ? alt : 1.0 / 3.0); /* (1/3).max(alt): a NaN loses */
            nu = 2.0;
        } else {
            mu *= nu;
            nu *= 2.0;
        }
        if (sqrt(dot_n(res, 1, res, 1, n)) <= RESIDUAL_EPSILON) {
            found = 1;
            break;
        }
    }
    return found;
}

/* optimize_perspective_f, :391-426.  Returns 1 and out_F = the optimised F, or 0 (the reference's None). */
int cvref_optimize_perspective_f(const double *F, const uint32_t *matches, uint32_t n, double *out_F)
{
    double params[7] = {F[0], F[1], F[2], F[3], F[4], F[5], F[6]};
    double *buf = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1) * 9);
    if (!buf) return 0;
    const int ok = lm_least_squares(params, matches, n, buf, buf + n, buf + 2 * (size_t)n);
    free(buf);
    if (!ok) return 0;
    double Fo[9], Ft[9], s[3];
    lm_f_from_params(params, Fo);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ft[i * 3 + j] = Fo[j * 3 + i];
    lm_singular_values3(Ft, s);
    if (fabs(s[1]) < 1e-3 || fabs(s[2]) > 1e-3) return 0; /* RANSAC_RANK_EPSILON_PERSPECTIVE, :418-423 */
    for (int k = 0; k < 9; k++) out_F[k] = Fo[k];
    return 1;
}
