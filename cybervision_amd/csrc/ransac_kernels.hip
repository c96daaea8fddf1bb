// ransac_kernels.hip — FundamentalMatrix::find_ransac (zlogic/cybervision src/fundamentalmatrix.rs:72-257) for gfx950.
//
// In file order:
//  * namespace lm: optimize_perspective_f (:391-426) - the reference's Levenberg-Marquardt loop (:515-621), Jacobian
//    (:473-512) and parameter map, ONE __host__ __device__ implementation for validate_f's per-hypothesis call
//    (device, 7 observations, register-resident) and the host form of the final refit;
//  * ransac_score_kernel: the all-matches fold of validate_f (:210-216) with fits_model / reprojection_error
//    (:452-471) for a batch of hypotheses, one lane per hypothesis, count and ordered error sum bit-identical to
//    --mode=cpu (cvhip_ransac_score);
//  * a ROUND's scoring as the device loops run it: live-slot compaction, ransac_count_kernel (one wave per two
//    hypotheses; f32 screen with rigorous bounds, f64 in the guard band: exact counts; its own copy of the match list in
//    the order that abandons hopeless hypotheses soonest), ransac_round_finish_kernel (the round's maximum, tie-break
//    sums only where a tie in the count needs them, Ord (:623-649) as a reduction, the re-sort);
//  * the generators: affine (4-point, one-sided Jacobi SVD) and perspective (7-point pencil / root / queued LM);
//  * ransac_rounds: generation in batches of rounds ahead of scoring on separate streams; optimize_result's tail on the device:
//    inlier compaction, ransac_refit_kernel (the refit, :246, on one workgroup, bit-equal to the host loop), second filter;
//  * the C entry points (include/cvhip.h).
// f64 throughout except the screen, contraction off; expression order follows nalgebra 0.35's
// published gemv / dot algorithms (column-by-column axpy; 3-vector dot = (a0*b0 + a1*b1) + a2*b2).
#include "cvhip_internal.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <new>
#include <utility>
#include <thread>
#include <vector>

namespace cvhip {

constexpr int RANSAC_TILE = 1024; // matches per LDS tile (16 KiB as 4 x u32)

// ---------------------------------------------------------------------------------------------------------
// optimize_perspective_f (fundamentalmatrix.rs:391-426) with the reference's own Levenberg-Marquardt loop
// (least_squares, :515-621) and analytic Jacobian (f_jacobian, :473-512).  ONE implementation for both of its
// call sites: validate_f runs it on every 7-point hypothesis (:201-205; on the device, inside the generator
// kernel, n = 7) and optimize_result on the winner's inliers (:246; host arithmetic like the reference's - a 7x7
// solve per iteration over <= a few thousand inliers).  The loop is kept as written there, also where it
// is not the textbook method: the Jacobian's denominator terms are plain sums c = d = (F p1)_0 + (F p1)_1 +
// (F' p2)_0 + (F' p2)_1, the step is params + (J'J + mu I)^-1 J'r, and a step that INCREASES the residual norm
// counts as converged (reduction < 0 * norm).  nalgebra 0.35 evaluation order as in the scoring kernel, plus:
// dot products over the inliers keep eight partial sums (blas.rs `dot`), LU is partial pivoting with
// multipliers scaled by the reciprocal pivot, triangular solves update column by column.
// ---------------------------------------------------------------------------------------------------------
namespace lm {

#if defined(__HIP_DEVICE_COMPILE__)
#define CVHIP_LM_INLINE __attribute__((always_inline)) inline
#else
#define CVHIP_LM_INLINE inline
#endif

struct Obs { // one match as the two homogeneous points
    double p1[3], p2[3];
};

__host__ __device__ CVHIP_LM_INLINE Obs make_obs(uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2)
{
    return Obs{{(double)x1, (double)y1, 1.0}, {(double)x2, (double)y2, 1.0}};
}

// dot product of two strided vectors (a column of the n x 7 Jacobian, or a plain vector)
__host__ __device__ CVHIP_LM_INLINE double long_dot(const double *a, uint32_t sa, const double *b, uint32_t sb, uint32_t n)
{
    double part[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    uint32_t i = 0;
    for (; n - i >= 8; i += 8) {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) part[k] += a[(size_t)(i + k) * sa] * b[(size_t)(i + k) * sb];
    }
    double total = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) total += part[k] + part[k + 4];
#pragma unroll
    for (; i < n; i++) total += a[(size_t)i * sa] * b[(size_t)i * sb];
    return total;
}

__host__ __device__ CVHIP_LM_INLINE void matrix_of(const double (&q)[7], double (&M)[9]) // f_from_perspective_params, :442-449
{
    const double last = -(-q[0] * q[4] + q[6] * q[2] * q[4] + q[3] * q[1] - q[6] * q[1] * q[5]) / (-q[3] * q[2] + q[0] * q[5]);
    for (int i = 0; i < 7; i++) M[i] = q[i];
    M[7] = last;
    M[8] = 1.0;
}

// row vector v' M and the chained product r . v, in nalgebra's orders
__host__ __device__ CVHIP_LM_INLINE void row_times(const double (&v)[3], const double (&M)[9], double (&out)[3])
{
    for (int j = 0; j < 3; j++) out[j] = (v[0] * M[j] + v[1] * M[3 + j]) + v[2] * M[6 + j];
}
__host__ __device__ CVHIP_LM_INLINE double chain3(const double (&r)[3], const double (&v)[3])
{
    double acc = r[0] * v[0];
    acc = r[1] * v[1] + acc;
    return r[2] * v[2] + acc;
}

__host__ __device__ CVHIP_LM_INLINE double residual_of(const double (&M)[9], const Obs &o) // reprojection_error, :461-471
{
    double r[3], mp1[3], mtp2[3];
    row_times(o.p2, M, r);
    const double top = chain3(r, o.p1);
    for (int i = 0; i < 3; i++) {
        const double row[3] = {M[3 * i], M[3 * i + 1], M[3 * i + 2]};
        mp1[i] = chain3(row, o.p1);
    }
    row_times(o.p2, M, mtp2); // (M' p2)_i = dot(column i of M, p2): the same numbers as p2' M
    return top * top / (mp1[0] * mp1[0] + mp1[1] * mp1[1] + mtp2[0] * mtp2[0] + mtp2[1] * mtp2[1]);
}

// f_jacobian (:473-512) in two pieces: the terms every element of a row shares, and one element
__host__ __device__ CVHIP_LM_INLINE double gradient_terms(const double (&M)[9], const Obs &o) // -> c (= d)
{
    double mp1[3], mtp2[3];
    for (int i = 0; i < 3; i++) {
        const double row[3] = {M[3 * i], M[3 * i + 1], M[3 * i + 2]};
        mp1[i] = chain3(row, o.p1);
    }
    for (int i = 0; i < 3; i++) mtp2[i] = (M[i] * o.p2[0] + M[3 + i] * o.p2[1]) + M[6 + i] * o.p2[2];
    return mp1[0] + mp1[1] + mtp2[0] + mtp2[1];
}
__host__ __device__ CVHIP_LM_INLINE double gradient_element(const double (&M)[9], const Obs &o, double c, int e)
{
    const double d = c;
    const int r = e / 3, k = e % 3;
    const double a = o.p2[r] * o.p1[k]; // p2' E_rk p1: the other eight products are exact zeros
    double rest[9];
    for (int i = 0; i < 9; i++) rest[i] = M[i];
    rest[3 * r + k] = 0.0;
    double rv[3];
    row_times(o.p2, rest, rv);
    const double b = chain3(rv, o.p1), x = M[3 * r + k];
    return 2.0 * (a * x + b) * (a * d - b * c * c * x) / (c * c * x * x + d);
}
__host__ __device__ CVHIP_LM_INLINE void gradient_of(const double (&M)[9], const Obs &o, double *out7) // f_jacobian, :473-512
{
    const double c = gradient_terms(M, o);
    for (int e = 0; e < 7; e++) out7[e] = gradient_element(M, o, c, e);
}

// (J'J + mu I) x = g, nalgebra's LU::new + LU::solve; false = "Failed to compute delta vector".  Every index below
// is a compile-time constant after unrolling (the pivot row is applied through per-row selects, not A[p]), so on the
// device the whole system stays in registers.
__host__ __device__ CVHIP_LM_INLINE bool solve7(double (&A)[49], double (&x)[7])
{
    constexpr int n = 7;
    int perm[n]; // row exchanged with row c while eliminating column c (c itself = none), applied to x in that order
#pragma unroll
    for (int c = 0; c < n; c++) {
        int p = c;
        double pivot = A[c * n + c], largest = fabs(pivot);
#pragma unroll
        for (int r = c + 1; r < n; r++) {
            const double v = A[r * n + c];
            if (fabs(v) > largest) {
                largest = fabs(v);
                pivot = v;
                p = r;
            }
        }
        perm[c] = c;
        if (pivot == 0.0) continue;
        perm[c] = p;
#pragma unroll
        for (int r = c + 1; r < n; r++)
            if (r == p) {
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const double tmp = A[c * n + k];
                    A[c * n + k] = A[r * n + k];
                    A[r * n + k] = tmp;
                }
            }
        const double rp = 1.0 / pivot;
#pragma unroll
        for (int r = c + 1; r < n; r++) A[r * n + c] *= rp;
#pragma unroll
        for (int k = c + 1; k < n; k++) {
            const double top = A[c * n + k];
#pragma unroll
            for (int r = c + 1; r < n; r++) A[r * n + k] = -top * A[r * n + c] + A[r * n + k];
        }
    }
#pragma unroll
    for (int c = 0; c < n; c++) {
#pragma unroll
        for (int r = c + 1; r < n; r++)
            if (perm[c] == r) {
                const double tmp = x[c];
                x[c] = x[r];
                x[r] = tmp;
            }
    }
#pragma unroll
    for (int c = 0; c < n; c++) {
        const double v = x[c];
#pragma unroll
        for (int r = c + 1; r < n; r++) x[r] = -v * A[r * n + c] + x[r];
    }
    bool regular = true;
#pragma unroll
    for (int c = n - 1; c >= 0; c--) {
        const double pivot = A[c * n + c];
        if (pivot == 0.0) regular = false;
        if (regular) {
            const double v = x[c] / pivot;
            x[c] = v;
#pragma unroll
            for (int r = 0; r < c; r++) x[r] = -v * A[r * n + c] + x[r];
        }
    }
    return regular;
}

// descending singular values of a 3x3 matrix from the eigenvalues of M'M (Jacobi rotations); they are only
// compared with 1e-3 (:362-366, :418-423)
__host__ __device__ inline void singular3(const double (&M)[9], double (&sv)[3])
{
    double g[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) g[i][j] = M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    // cyclic Jacobi converges quadratically: a handful of sweeps take the off-diagonal part below 1e-20 of the diagonal,
    // far beyond what a comparison with 1e-3 can see (waiting for EXACT zeros made every lane of a wave sit through
    // up to 32 sweeps of two square roots and two divisions per rotation)
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = fabs(g[0][1]) + fabs(g[0][2]) + fabs(g[1][2]);
        if (!(off > 1e-20 * (fabs(g[0][0]) + fabs(g[1][1]) + fabs(g[2][2])))) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (g[p][q] == 0.0) continue;
                const double th = (g[q][q] - g[p][p]) / (2.0 * g[p][q]);
                const double t = __builtin_copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) {
                    const double u = g[k][p], v = g[k][q];
                    g[k][p] = cs * u - sn * v;
                    g[k][q] = sn * u + cs * v;
                }
                for (int k = 0; k < 3; k++) {
                    const double u = g[p][k], v = g[q][k];
                    g[p][k] = cs * u - sn * v;
                    g[q][k] = sn * u + cs * v;
                }
            }
    }
    double a = g[0][0], b = g[1][1], c = g[2][2];
    if (a < b) { const double t = a; a = b; b = t; }
    if (b < c) { const double t = b; b = c; c = t; }
    if (a < b) { const double t = a; a = b; b = t; }
    sv[0] = sqrt(a > 0.0 ? a : 0.0);
    sv[1] = sqrt(b > 0.0 ? b : 0.0);
    sv[2] = sqrt(c > 0.0 ? c : 0.0);
}

// least_squares (:515-621) on this problem; false = Err.  Workspace: r, r_new [n], J [n x 7].  N = the number of
// observations when it is known at compile time (validate_f's 7: every loop unrolls, every index is a constant and
// the workspace lives in registers), 0 = n_obs at run time (the refit over the inliers).  Same statements either way.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed" // (the unroll requests are for N = 7; N = 0 has run-time trip counts)
// `event` (optional, for callers that continue the root elsewhere): with a `budget` below the reference's 1000 trips the
// function returns false with *event = LM_EVENT_BUDGET when the budget is used up, and with stop_on_accept it returns
// false with *event = LM_EVENT_ACCEPTED at the first ACCEPTED step (before that step's linearisation); q is unspecified then.
constexpr int LM_EVENT_BUDGET = 2, LM_EVENT_ACCEPTED = 3;
template <int N>
__host__ __device__ CVHIP_LM_INLINE bool levenberg_marquardt(double (&q)[7], const Obs *obs, uint32_t n_obs, double *r,
                                                             double *r_new, double *J, int budget = 1000, int *event = nullptr,
                                                             bool stop_on_accept = false)
{
    const uint32_t n = N > 0 ? (uint32_t)N : n_obs;
    double M[9], g[7];
    const auto evaluate = [&](const double (&at)[7], double *into) {
        matrix_of(at, M);
#pragma unroll
        for (uint32_t i = 0; i < n; i++) into[i] = residual_of(M, obs[i]);
    };
    const auto linearise = [&](const double (&at)[7], const double *res) { // Jacobian and J'r at `at`
        matrix_of(at, M);
#pragma unroll
        for (uint32_t i = 0; i < n; i++) gradient_of(M, obs[i], &J[(size_t)i * 7]);
#pragma unroll
        for (int j = 0; j < 7; j++) g[j] = long_dot(J + j, 7, res, 1, n);
    };
    const auto largest = [](const double (&v)[7]) {
        double m = v[0];
#pragma unroll
        for (int j = 1; j < 7; j++)
            if (m < v[j]) m = v[j];
        return m;
    };
    const auto norm7 = [](const double (&v)[7]) { return sqrt(long_dot(v, 1, v, 1, 7)); };

    evaluate(q, r);
    linearise(q, r);
    if (fabs(largest(g)) <= 1e-12) return true;
    double mu = 0.0;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const double djj = long_dot(J + j, 7, J + j, 7, n);
        if (j == 0 || djj >= mu) mu = djj;
    }
    mu *= 1e-3;
    double nu = 2.0;
    for (int iteration = 0; iteration < 1000; iteration++) {
        if (iteration >= budget) { // (the reference's cap is the 1000 above)
            *event = LM_EVENT_BUDGET;
            return false;
        }
        double A[49];
#pragma unroll
        for (int i = 0; i < 7; i++) {
#pragma unroll
            for (int j = 0; j < 7; j++) A[i * 7 + j] = long_dot(J + i, 7, J + j, 7, n);
        }
#pragma unroll
        for (int i = 0; i < 7; i++) A[i * 7 + i] += mu;
        double step[7];
#pragma unroll
        for (int j = 0; j < 7; j++) step[j] = g[j];
        if (!solve7(A, step)) return false;
        if (norm7(step) <= 1e-12 * (norm7(q) + 1e-12)) return true;
        double trial[7], damped[7];
#pragma unroll
        for (int j = 0; j < 7; j++) trial[j] = q[j] + step[j];
        evaluate(trial, r_new);
        const double before = long_dot(r, 1, r, 1, n);
        const double after = long_dot(r_new, 1, r_new, 1, n);
#pragma unroll
        for (int j = 0; j < 7; j++) damped[j] = step[j] * mu + g[j];
        const double rho = (before - after) / long_dot(step, 1, damped, 1, 7);
        if (rho > 0.0) {
            if (stop_on_accept) {
                *event = LM_EVENT_ACCEPTED;
                return false;
            }
            const bool converged = sqrt(before) - sqrt(after) < 0.0 * sqrt(before);
#pragma unroll
            for (uint32_t i = 0; i < n; i++) r[i] = r_new[i];
#pragma unroll
            for (int j = 0; j < 7; j++) q[j] = trial[j];
            linearise(q, r);
            if (converged || fabs(largest(g)) <= 1e-12) return true;
            const double w = 2.0 * rho - 1.0, shrink = 1.0 - w * w * w;
            mu *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
            nu = 2.0;
        } else {
            mu *= nu;
            nu *= 2.0;
        }
        if (sqrt(long_dot(r, 1, r, 1, n)) <= 1e-12) return true;
    }
    return false; // "Levenberg-Marquardt failed to converge"
}
#pragma clang diagnostic pop

// The first thing least_squares does (:542-550) is test the gradient at the start: max(J'r) <= 1e-12 returns the
// parameters unchanged.  For the 7 observations of validate_f this evaluates that test with the SAME operations in
// the same order as levenberg_marquardt (r_i, row i of J, g_j = 0 + J_0j r_0 + J_1j r_1 + ... - long_dot's order for
// n < 8) but without the r / J arrays: a 7-point solution has (nearly) zero residuals on its own sample, so on the
// device this is all the LM ever does for ~all hypotheses, and it stays in registers.
__host__ __device__ inline bool converged_at_start7(const double (&q)[7], const Obs *obs)
{
    double M[9], g[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    matrix_of(q, M);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double ri = residual_of(M, obs[i]);
        double row[7];
        gradient_of(M, obs[i], row);
#pragma unroll
        for (int j = 0; j < 7; j++) g[j] += row[j] * ri;
    }
    double m = g[0];
#pragma unroll
    for (int j = 1; j < 7; j++)
        if (m < g[j]) m = g[j];
    return fabs(m) <= 1e-12;
}

// validate_f's call (n = 7), run by the device's thread-per-root LM kernels for the roots that fail the start test (it
// re-evaluates the start: same values).  -> 1 = Ok, 0 = Err, LM_EVENT_BUDGET = still running after `budget` iterations,
// LM_EVENT_ACCEPTED = (stop_on_accept) a step was accepted (q is then unspecified: the caller restarts the root elsewhere)
__host__ __device__ CVHIP_LM_INLINE int levenberg_marquardt7(double (&q)[7], const Obs *obs, int budget, bool stop_on_accept = false)
{
    double r[7], r_new[7], J[49];
    int event = 0;
    const bool ok = levenberg_marquardt<7>(q, obs, 7, r, r_new, J, budget, &event, stop_on_accept);
    return event ? event : (ok ? 1 : 0);
}

// The same loop for validate_f's seven observations with the Jacobian never stored: row k's seven entries are folded into
// g = J'r and the 28 distinct entries of J'J as they are made, k ascending - the order of blas `dot` for seven terms (0 + t0
// + t1 + ...), and J'J's (i, j) and (j, i) are the same sum of the same products - so every value is levenberg_marquardt<7>'s,
// bit for bit (the per-sample tests compare the two).  What it saves: J'J is rebuilt only when J changed - a rejected step
// leaves it as it was, and in the thin-SVD pencil's regime two thirds of the roots only ever have rejected steps - and the
// 49 Jacobian entries do not live across the trips.  -> 1 = Ok, 0 = Err, LM_EVENT_BUDGET, LM_EVENT_ACCEPTED (stop_on_accept).
__host__ __device__ CVHIP_LM_INLINE int levenberg_marquardt7_lean(double (&q)[7], const Obs *obs, int budget, bool stop_on_accept)
{
    double M[9], g[7], r[7], jj[28];
    const auto evaluate = [&](const double (&at)[7], double (&into)[7]) {
        matrix_of(at, M);
#pragma unroll
        for (int i = 0; i < 7; i++) into[i] = residual_of(M, obs[i]);
    };
    const auto linearise = [&](const double (&at)[7], const double (&res)[7]) { // J'r and J'J at `at`
        matrix_of(at, M);
#pragma unroll
        for (int j = 0; j < 7; j++) g[j] = 0.0;
#pragma unroll
        for (int m = 0; m < 28; m++) jj[m] = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            double row[7];
            gradient_of(M, obs[k], row);
            int m = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                g[i] += row[i] * res[k];
#pragma unroll
                for (int j = i; j < 7; j++) jj[m++] += row[i] * row[j];
            }
        }
    };
    const auto largest = [](const double (&v)[7]) {
        double m = v[0];
#pragma unroll
        for (int j = 1; j < 7; j++)
            if (m < v[j]) m = v[j];
        return m;
    };
    const auto norm7 = [](const double (&v)[7]) { return sqrt(long_dot(v, 1, v, 1, 7)); };

    evaluate(q, r);
    linearise(q, r);
    if (fabs(largest(g)) <= 1e-12) return 1;
    double mu = 0.0;
    {
        int m = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const double djj = jj[m];
            if (j == 0 || djj >= mu) mu = djj;
            m += 7 - j;
        }
    }
    mu *= 1e-3;
    double nu = 2.0;
    for (int iteration = 0; iteration < 1000; iteration++) {
        if (iteration >= budget) return LM_EVENT_BUDGET;
        double A[49], step[7];
        {
            int m = 0;
#pragma unroll
            for (int i = 0; i < 7; i++)
#pragma unroll
                for (int j = i; j < 7; j++) {
                    A[i * 7 + j] = jj[m];
                    A[j * 7 + i] = jj[m++];
                }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) A[i * 7 + i] += mu;
#pragma unroll
        for (int j = 0; j < 7; j++) step[j] = g[j];
        if (!solve7(A, step)) return 0;
        if (norm7(step) <= 1e-12 * (norm7(q) + 1e-12)) return 1;
        double trial[7], damped[7], r_new[7];
#pragma unroll
        for (int j = 0; j < 7; j++) trial[j] = q[j] + step[j];
        evaluate(trial, r_new);
        const double before = long_dot(r, 1, r, 1, 7);
        const double after = long_dot(r_new, 1, r_new, 1, 7);
#pragma unroll
        for (int j = 0; j < 7; j++) damped[j] = step[j] * mu + g[j];
        const double rho = (before - after) / long_dot(step, 1, damped, 1, 7);
        if (rho > 0.0) {
            if (stop_on_accept) return LM_EVENT_ACCEPTED;
            const bool converged = sqrt(before) - sqrt(after) < 0.0 * sqrt(before);
#pragma unroll
            for (int j = 0; j < 7; j++) {
                r[j] = r_new[j];
                q[j] = trial[j];
            }
            linearise(q, r);
            if (converged || fabs(largest(g)) <= 1e-12) return 1;
            const double w = 2.0 * rho - 1.0, shrink = 1.0 - w * w * w;
            mu *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
            nu = 2.0;
        } else {
            mu *= nu;
            nu *= 2.0;
        }
        if (sqrt(long_dot(r, 1, r, 1, 7)) <= 1e-12) return 1;
    }
    return 0; // "Levenberg-Marquardt failed to converge"
}

// optimize_perspective_f (:391-426): F (normalised by F[2][2]) -> out, false = None
__host__ __device__ inline bool optimize_perspective_f(const double (&F)[9], const Obs *obs, uint32_t n, double *r,
                                                       double *r_new, double *J, double (&out)[9])
{
    double q[7];
    for (int i = 0; i < 7; i++) q[i] = F[i]; // params_from_perspective_f, :429-440
    if (!levenberg_marquardt<0>(q, obs, n, r, r_new, J)) return false;
    matrix_of(q, out);
    const double Mt[9] = {out[0], out[3], out[6], out[1], out[4], out[7], out[2], out[5], out[8]};
    double s[3];
    singular3(Mt, s);
    return !(fabs(s[1]) < 1e-3 || fabs(s[2]) > 1e-3); // :418-423
}

} // namespace lm

__device__ __forceinline__ double reprojection_error(const double (&F)[9], double p1x, double p1y, double p2x,
                                                     double p2y)
{
    // p2.tr_mul(f): element j = dot(p2, F[:, j]) = (p2x*F0j + p2y*F1j) + 1*F2j
    const double r0 = (p2x * F[0] + p2y * F[3]) + F[6];
    const double r1 = (p2x * F[1] + p2y * F[4]) + F[7];
    const double r2 = (p2x * F[2] + p2y * F[5]) + F[8];
    // (1x3) * p1, gemv order
    double n = r0 * p1x;
    n = r1 * p1y + n;
    n = r2 + n;
    // f * p1, rows 0 and 1
    double a0 = F[0] * p1x;
    a0 = F[1] * p1y + a0;
    a0 = F[2] + a0;
    double a1 = F[3] * p1x;
    a1 = F[4] * p1y + a1;
    a1 = F[5] + a1;
    // f.tr_mul(p2): element i = dot(F[:, i], p2), i = 0, 1
    const double b0 = (F[0] * p2x + F[3] * p2y) + F[6];
    const double b1 = (F[1] * p2x + F[4] * p2y) + F[7];
    const double nominator = n * n;
    const double denominator = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
    return nominator / denominator;
}

struct RansacBest {
    double f[9];
    double best_error;
    uint32_t matches_count;
    uint32_t valid;
    uint32_t err_known; // best_error has been computed (it is only ever needed to break a tie in matches_count)
    uint32_t pad;       // (ransac_pick_best_approx_kernel's note of the count its match order was made for)
    // Position of this hypothesis in the reference's iteration order (round x slots + slot), where the caller says it
    // (slot_base): batches of rounds may then be scored in any order - among hypotheses that are EQUAL under Ord the one
    // the reference meets first stays.  Scored in order it changes nothing (a later hypothesis never precedes the best).
    uint32_t origin, origin_pad;
};

// The fold of validate_f (:210-216) for a batch of hypotheses, one lane per hypothesis.
//  * live / n_live (optional): the hypotheses to score are F[live[0 .. *n_live)] - the generators leave most slots
//    empty (no real root, a failed rank / sign / sample-fit check), and an empty slot must not cost N error
//    evaluations.  Slots that are not listed keep whatever out_count holds (the caller zeroes it).
//  * bound / best (optional): a hypothesis whose count can no longer reach `min_count`, or the inlier count of the
//    best hypothesis of the PREVIOUS rounds, cannot become the result (Ord, :623-649: more matches always win), so it
//    is abandoned at the next tile boundary and reported with count 0 - the winner and every hypothesis that could
//    still beat or tie it are folded completely, so the result is the reference's.
//  * matches are converted to f64 once per tile (LDS), and the division of reprojection_error is only executed where
//    the inlier test is open: n^2 > t * den * (1 + 2^-40) implies n^2 / den > t in f64 whatever the roundings, so
//    such a match is an outlier without dividing (non-finite cases fall through to the exact expression).
constexpr int SCORE_BLOCK = 256; // hypotheses per workgroup: four waves share one staged tile of matches (32 KB), so five
                                 // workgroups = 20 waves fit a CU (one wave per 32 KB tile left the SIMDs at 1 wave each)
__global__ __launch_bounds__(SCORE_BLOCK) void ransac_score_kernel(const double *__restrict__ F, uint32_t H,
                                                           const uint4 *__restrict__ matches, uint32_t N, double t,
                                                           const uint32_t *__restrict__ live,
                                                           const uint32_t *__restrict__ n_live, uint32_t min_count,
                                                           const RansacBest *__restrict__ best,
                                                           uint32_t *__restrict__ out_count,
                                                           double *__restrict__ out_err_sum)
{
    __shared__ double tile[RANSAC_TILE][4];
    const uint32_t n_hyp = n_live ? *n_live : H;
    if (blockIdx.x * (uint32_t)SCORE_BLOCK >= n_hyp) return;
    const uint32_t j = blockIdx.x * SCORE_BLOCK + threadIdx.x;
    const bool active = j < n_hyp;
    const uint32_t h = active ? (live ? live[j] : j) : 0u;
    double f[9];
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = active ? F[(size_t)h * 9 + i] : 0.0;
    uint32_t bound = 0;
    if (best) bound = best->valid ? max(min_count, best->matches_count) : min_count;
    const double t_hi = t * (1.0 + 0x1p-40);
    uint32_t count = 0;
    double sum = 0.0;
    bool alive = active;
    for (uint32_t base = 0; base < N; base += RANSAC_TILE) {
        const uint32_t n = min((uint32_t)RANSAC_TILE, N - base);
        // count + (matches not yet seen) < bound: this hypothesis is out
        alive = alive && !(count + (N - base) < bound);
        if (!__syncthreads_or(alive ? 1 : 0)) break; // (also the barrier before the tile is overwritten)
        for (uint32_t i = threadIdx.x; i < n; i += SCORE_BLOCK) {
            const uint4 m = matches[base + i];
            tile[i][0] = (double)m.x;
            tile[i][1] = (double)m.y;
            tile[i][2] = (double)m.z;
            tile[i][3] = (double)m.w;
        }
        __syncthreads();
        if (alive) {
            for (uint32_t i = 0; i < n; i++) {
                const double p1x = tile[i][0], p1y = tile[i][1], p2x = tile[i][2], p2y = tile[i][3];
                // reprojection_error (:461-471) in nalgebra's evaluation order, as in reprojection_error() above
                const double r0 = (p2x * f[0] + p2y * f[3]) + f[6];
                const double r1 = (p2x * f[1] + p2y * f[4]) + f[7];
                const double r2 = (p2x * f[2] + p2y * f[5]) + f[8];
                double nn = r0 * p1x;
                nn = r1 * p1y + nn;
                nn = r2 + nn;
                double a0 = f[0] * p1x;
                a0 = f[1] * p1y + a0;
                a0 = f[2] + a0;
                double a1 = f[3] * p1x;
                a1 = f[4] * p1y + a1;
                a1 = f[5] + a1;
                const double nominator = nn * nn;
                const double denominator = a0 * a0 + a1 * a1 + r0 * r0 + r1 * r1; // (F' p2)_i == (p2' F)_i
                if (nominator > t_hi * denominator) continue; // certainly err > t: an outlier, no division needed
                const double err = nominator / denominator;
                // fits_model: finite and |err| <= t (fundamentalmatrix.rs:452-458)
                if (fabs(err) < __builtin_inf() && !(fabs(err) > t)) {
                    count += 1;
                    sum += err;
                }
            }
        }
    }
    if (active) {
        out_count[h] = alive ? count : 0u;
        out_err_sum[h] = alive ? sum : 0.0;
    }
}

// Ordered compaction of the live hypothesis slots (F[9 * h] is not NaN) -> live[0 .. *n_live) ascending:
// per-1024-slot counts, a single-block scan of the (~150) counts, then the scatter.
__global__ __launch_bounds__(1024) void ransac_live_count_kernel(const double *__restrict__ F, uint32_t H,
                                                                  uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wtot[16];
    const uint32_t h = blockIdx.x * 1024 + threadIdx.x;
    const double f0 = h < H ? F[(size_t)h * 9] : __builtin_nan("");
    const unsigned long long b = __ballot(f0 == f0);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; w++) tot += wtot[w];
        block_counts[blockIdx.x] = tot;
    }
}
__global__ __launch_bounds__(1024) void ransac_live_scatter_kernel(const double *__restrict__ F, uint32_t H,
                                                                    const uint32_t *__restrict__ block_offsets,
                                                                    uint32_t *__restrict__ live)
{
    __shared__ uint32_t wtot[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t h = blockIdx.x * 1024 + threadIdx.x;
    const double f0 = h < H ? F[(size_t)h * 9] : __builtin_nan("");
    const bool is_live = f0 == f0;
    const unsigned long long b = __ballot(is_live);
    if (lane == 0) wtot[wv] = (uint32_t)__popcll(b);
    __syncthreads();
    if (is_live) {
        uint32_t off = block_offsets[blockIdx.x];
        for (uint32_t w = 0; w < wv; w++) off += wtot[w];
        live[off + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = h;
    }
}

void launch_ransac_score(const double *F, uint32_t H, const uint32_t *matches, uint32_t N, double t,
                         uint32_t *out_count, double *out_err_sum, hipStream_t s)
{
    if (!H) return;
    hipLaunchKernelGGL(ransac_score_kernel, dim3((H + SCORE_BLOCK - 1) / SCORE_BLOCK), dim3(SCORE_BLOCK), 0, s, F, H,
                       reinterpret_cast<const uint4 *>(matches), N, t, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u,
                       (const RansacBest *)nullptr, out_count, out_err_sum);
}

// ---------------------------------------------------------------------------------------------
// One RANSAC round's scoring, parallel over hypotheses AND matches.  A round holds ~25 000 live hypotheses; one lane
// per hypothesis folding 29 000 matches serially (ransac_score_kernel above, kept for cvhip_ransac_score, whose
// contract is the ordered error sum of EVERY hypothesis) leaves the GPU at ~100 workgroups and bound by the latency
// of one thread's loop.  The round needs less than that contract: Ord (:623-649) looks at the inlier COUNT first and
// at the mean error only between hypotheses of equal count.  So:
//   1. ransac_count_kernel - one WAVE per live hypothesis, its lanes striding over the matches; the inlier predicate
//      is the same f64 expression as in the fold (order-free), the count a sum of ballots; a hypothesis that can no
//      longer reach min_count or the best count of the previous rounds is abandoned (count 0).
//   2. ransac_round_max_kernel - the largest count of the round and the list of hypotheses that have it.
//   3. ransac_tied_sum_kernel - only for the hypotheses AT that maximum (usually one): the reference's fold, errors
//      computed in parallel, added serially in match order - bit-equal to the serial kernel's sum.
// Every other hypothesis loses on its count, so its error sum is never looked at (it is reported as 0).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool match_fits(const double (&f)[9], uint4 m, double t, double t_hi, double &err)
{
    const double p1x = (double)m.x, p1y = (double)m.y, p2x = (double)m.z, p2y = (double)m.w;
    const double r0 = (p2x * f[0] + p2y * f[3]) + f[6];
    const double r1 = (p2x * f[1] + p2y * f[4]) + f[7];
    const double r2 = (p2x * f[2] + p2y * f[5]) + f[8];
    double nn = r0 * p1x;
    nn = r1 * p1y + nn;
    nn = r2 + nn;
    double a0 = f[0] * p1x;
    a0 = f[1] * p1y + a0;
    a0 = f[2] + a0;
    double a1 = f[3] * p1x;
    a1 = f[4] * p1y + a1;
    a1 = f[5] + a1;
    const double nominator = nn * nn;
    const double denominator = a0 * a0 + a1 * a1 + r0 * r0 + r1 * r1;
    err = 0.0;
    if (nominator > t_hi * denominator) return false; // certainly err > t (see ransac_score_kernel)
    err = nominator / denominator;
    return fabs(err) < __builtin_inf() && !(fabs(err) > t); // fits_model, :452-458
}

// Largest coordinate of the match list (one word, once per call): the scale of the f32 screen's error bounds.
// Also writes the match list once as f32 (exact below 2^24; a larger coordinate switches the screen off through W), so
// that the counting kernel's screen does not convert the same 29 000 matches for every one of 25 000 hypotheses.
// The f32 copy is laid out for the counting kernel: matches go in groups of 256, lane l of a wave handles matches
// 256 g + l + 64 h (h = 0..3), so plane c (x1, y1, x2, y2) holds the float4 {c of those four matches} at index 64 g + l:
// four 16-byte loads per lane and step.  Padded with zeros to a whole group.
constexpr uint32_t COUNT_GROUP = 256;
__host__ __device__ inline uint32_t ransac_padded(uint32_t N) { return (N + COUNT_GROUP - 1u) / COUNT_GROUP * COUNT_GROUP; }
// (index of match i's value inside a plane of ransac_padded(N) floats)
__host__ __device__ inline uint32_t ransac_plane_slot(uint32_t i) { return 4u * (64u * (i >> 8) + (i & 63u)) + ((i >> 6) & 3u); }
// how far the matrix-pipe counting phase walks (ransac_count_mfma_kernel): no hypothesis can be abandoned before match N - bound
__host__ __device__ inline uint32_t ransac_phase_len(uint32_t N, uint32_t bound)
{
    const uint32_t np = ransac_padded(N), head = N - (bound < N ? bound : N);
    const uint32_t L = ransac_padded(head) + 2u * COUNT_GROUP;
    return L < np ? L : np;
}
__global__ __launch_bounds__(1024) void ransac_coord_max_kernel(const uint4 *__restrict__ matches, uint32_t N,
                                                                 uint32_t *__restrict__ out, float4 *__restrict__ matches_f32)
{
    __shared__ uint32_t wmax[16];
    uint32_t m = 1u;
    const uint32_t np = ransac_padded(N);
    float *planes = reinterpret_cast<float *>(matches_f32); // 4 planes of np floats
    for (uint32_t i = threadIdx.x; i < np; i += 1024) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < N) {
            v = matches[i];
            m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
        }
        const uint32_t slot = ransac_plane_slot(i);
        planes[slot] = (float)v.x;
        planes[np + slot] = (float)v.y;
        planes[2u * np + slot] = (float)v.z;
        planes[3u * np + slot] = (float)v.w;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (uint32_t)__shfl_down(m, sft, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) m = max(m, wmax[w]);
        *out = m;
    }
}

// The counting kernel decides most (hypothesis, match) pairs in f32 - plain v_fma_f32, four matches per lane and step
// (packed f32 was tried first: a v_pk_fma_f32 holds the SIMD for ~8.8 cycles against 4 for each of the two plain
// instructions it replaces) - with fused multiply-adds allowed because nothing here is a
// result: the f32 values of n and of the denominator come with
// rigorous error bounds (below), "certainly an inlier" / "certainly not" are decided against those, and only a pair
// that falls inside the guard band (~1e-4 of them) is evaluated with the reference's f64 expression (match_fits).
// The count is therefore exactly the f64 count.
// Bounds (u = 2^-24; W = largest coordinate; |f| = the f64 coefficients' magnitudes, their f32 copies add one u):
//   r_j, a_j: three terms, two fmas + the coefficient rounding: |error| <= 4u * T_j,  T_j = (|f|+|f|) W + |f| >= |value|
//   n = r0 x + r1 y + r2: the operands' errors times W plus three roundings: E_n = 7u (T_r0 W + T_r1 W + T_r2)
//   den = a0^2 + a1^2 + r0^2 + r1^2 (non-negative terms): |t^2 - t_exact^2| <= E_t (2|t| + E_t) per term,
//     E_den = 8 E_max T_max + 4 E_max^2 absolute, plus 5u relative (roundings of a positive sum).
// The decision is taken on q = n^2 - t den (inlier <=> q <= 0; one multiplication, one fma):
//   |q_f32 - q_exact| <= u |q_f32| + E_n (2|n| + E_n) + t E_den + 8u t den
//     (the fma's rounding; n's error through the square; den's absolute error; den's relative error and the roundings of
//     t and of t den), and the margin the kernel compares |q| with is
//   m = 2^-18 (t den) + 2 E_n |n| + (E_n^2 + t E_den), constants inflated by 2^-20 and by the 0.1 % of E_n, E_den:
//     2^-18 against 8u = 2^-21 leaves room for the roundings of m itself, for u |q| and for the f64 expression's own 2^-50.
//   |q| > m, q < 0: certainly an inlier (then t den > t E_den: the exact denominator is positive and err is finite);
//   |q| > m, q > 0: certainly none; otherwise (and for NaN, which compares false) the f64 expression decides.
// A hypothesis with non-finite, tiny or huge coefficients is not screened at all (its pairs all take the f64 path).
__device__ __forceinline__ float wave_uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// COUNT_K live hypotheses per wave (config 5's RANSAC stage, packed-f32 form, ms: one per wave 45.7, two 42.6, four 46.6 -
// the match list is streamed from L2 once per wave, two hypotheses share it, four cost too many registers).  The kernel
// is bound by vector issue (SQ_ACTIVE_INST_VALU 0.92 - 0.98 of its cycles in this form), so what counts is the number
// of vector instructions per pair: 23.  Everything per hypothesis is wave-uniform and - the wave index being made a
// SCALAR with readfirstlane - lives in scalar registers where the compiler allows: bounds, the running count, the alive
// flag; the loop's branches are scalar branches and the lane masks of the tests stay SGPR pairs.  The list is loaded
// one step AHEAD of the arithmetic.
constexpr uint32_t TIED_CAP = 4096; // hypotheses at the round's maximum count that get an ordered error sum (usually 1)
constexpr int COUNT_K = 2;
__global__ __launch_bounds__(256) void ransac_count_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches,
                                                            uint32_t N, double t, const uint32_t *__restrict__ live,
                                                            const uint32_t *__restrict__ n_live, uint32_t min_count,
                                                            const RansacBest *__restrict__ best,
                                                            const uint32_t *__restrict__ coord_max,
                                                            const float4 *__restrict__ matches_f32,
                                                            uint32_t *__restrict__ out_count,
                                                            double *__restrict__ out_err_sum, uint32_t *__restrict__ cand,
                                                            uint32_t live_first, uint32_t live_end,
                                                            const uint32_t *__restrict__ start_count = nullptr)
{
    // start_count (phase 2 behind ransac_count_mfma_kernel): `live` is that kernel's survivor list, start_count[j] the inliers
    // survivor j has among the first ransac_phase_len(N, bound) matches, and the walk continues there.
    // live_first / live_end: the part of the live list this launch scores (the first round goes in two parts: a small
    // head that gives the rest a bound and an order to be abandoned by)
    // cand (the device loops; optional): [0] the largest count completed so far in this launch, [1] the number of
    // candidates, then (slot, count) pairs - every hypothesis whose count was at least the largest seen when it finished,
    // which includes everyone at the final maximum.  The running maximum also raises the abandonment bound: a hypothesis
    // that cannot reach a count somebody already HAS cannot be the round's winner, nor tie with it.
    const uint32_t n_hyp = min(*n_live, live_end), lane = threadIdx.x & 63;
    const uint32_t j0 = live_first + (blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * COUNT_K;
    if (j0 >= n_hyp || N == 0) return; // (an empty list: the counts stay at the zeros they were cleared to)
    uint32_t bound = best->valid ? max(min_count, best->matches_count) : min_count;
    const uint32_t base0 = start_count ? ransac_phase_len(N, bound) : 0u; // (a multiple of COUNT_GROUP, at most the padded length)
    const double t_hi = t * (1.0 + 0x1p-40);
    const double W = (double)*coord_max, u = 0x1p-24;
    const float T_f = wave_uniform((float)t);
    // per hypothesis of the group (all wave-uniform)
    float ff[COUNT_K][9], En2[COUNT_K], C0[COUNT_K];
    bool screen[COUNT_K], alive[COUNT_K];
    uint32_t slot[COUNT_K], count[COUNT_K];
#pragma unroll
    for (int k = 0; k < COUNT_K; k++) {
        alive[k] = j0 + k < n_hyp;
        slot[k] = live[alive[k] ? j0 + k : j0];
        count[k] = start_count ? start_count[alive[k] ? j0 + k : j0] : 0u;
        double af[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double fi = F[(size_t)slot[k] * 9 + i];
            af[i] = fabs(fi);
            ff[k][i] = wave_uniform((float)fi);
        }
        // the screen's constants (derivation above)
        const double Tr0 = (af[0] + af[3]) * W + af[6], Tr1 = (af[1] + af[4]) * W + af[7], Tr2 = (af[2] + af[5]) * W + af[8];
        const double Ta0 = (af[0] + af[1]) * W + af[2], Ta1 = (af[3] + af[4]) * W + af[5];
        const double Tmax = fmax(fmax(Tr0, Tr1), fmax(Ta0, Ta1)), Emax = 4.0 * u * Tmax;
        const double En_d = 1.001 * 7.0 * u * (Tr0 * W + Tr1 * W + Tr2), Eden_d = 1.001 * (8.0 * Emax * Tmax + 4.0 * Emax * Emax);
        // (as f32, rounded up by one more 2^-20)
        En2[k] = wave_uniform((float)(2.0 * En_d * (1.0 + 0x1p-20)));
        C0[k] = wave_uniform((float)((En_d * En_d + t * Eden_d) * (1.0 + 0x1p-20)));
        // the screen runs only where every bound and every product it is compared with stays a NORMAL f32 number
        // (E_den >= 32u Tmax^2 >= 2e-18, times t >= 1e-6; squares below 1e31); false for NaN / inf coefficients too
        screen[k] = Tmax >= 1e-6 && Tmax * W <= 1e12 && Tr2 <= 1e12 && t >= 1e-6 && t <= 1e12 && W < 16777216.0;
    }
    const uint32_t np = ransac_padded(N);
    const float4 *const px1 = matches_f32, *const py1 = px1 + np / 4, *const px2 = px1 + np / 2, *const py2 = px1 + 3 * (np / 4);
    const uint32_t q0 = ((base0 < N ? base0 : 0u) >> 2) + lane;
    float4 p1x = px1[q0], p1y = py1[q0], p2x = px2[q0], p2y = py2[q0]; // the first step's matches
    for (uint32_t base = base0; base < N; base += COUNT_GROUP) {
        bool any_alive = false;
#pragma unroll
        for (int k = 0; k < COUNT_K; k++) {
            // a hypothesis that cannot reach the bound any more is out, whatever the remaining matches do
            alive[k] = alive[k] && !(count[k] + (N - base) < bound);
            any_alive = any_alive || alive[k];
        }
        if (!any_alive) break;
        // the next step's matches are on their way while this step's are tested
        // (the last step loads itself again: no branch, nothing read past the end)
        const uint32_t qn = ((base + COUNT_GROUP < N ? base + COUNT_GROUP : base) >> 2) + lane;
        const float4 n1x = px1[qn], n1y = py1[qn], n2x = px2[qn], n2y = py2[qn];
        // (the running maximum as of now, for the next step's test - looked at every eighth step only: it is ONE word that
        // every wave of the launch reads past its L1, and at every step the L2 channel that holds it was the bottleneck)
        uint32_t running = 0;
        if (cand && ((base / COUNT_GROUP) & 7u) == 7u) running = __hip_atomic_load(&cand[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the ragged last group: lanes past the end count for nothing
        unsigned long long valid[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (base + COUNT_GROUP > N) {
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const uint32_t first = base + 64u * h;
                valid[h] = first >= N ? 0ull : (N - first >= 64u ? ~0ull : (1ull << (N - first)) - 1ull);
            }
        }
#pragma unroll
        for (int k = 0; k < COUNT_K; k++) {
            if (!alive[k]) continue; // (scalar)
            // lane masks (scalar registers) per quarter of the group: certainly in, still open
            unsigned long long in[4] = {0ull, 0ull, 0ull, 0ull}, open[4] = {valid[0], valid[1], valid[2], valid[3]};
            if (screen[k]) {
                const float *c = ff[k];
#pragma unroll
                for (int h = 0; h < 4; h++) { // the lane's four matches, one after the other (see the note on packed f32 above)
                    const float x1 = p1x[h], y1 = p1y[h], x2 = p2x[h], y2 = p2y[h];
                    const float r0 = __builtin_fmaf(x2, c[0], __builtin_fmaf(y2, c[3], c[6]));
                    const float r1 = __builtin_fmaf(x2, c[1], __builtin_fmaf(y2, c[4], c[7]));
                    const float r2 = __builtin_fmaf(x2, c[2], __builtin_fmaf(y2, c[5], c[8]));
                    const float nn = __builtin_fmaf(r0, x1, __builtin_fmaf(r1, y1, r2));
                    const float a0 = __builtin_fmaf(x1, c[0], __builtin_fmaf(y1, c[1], c[2]));
                    const float a1 = __builtin_fmaf(x1, c[3], __builtin_fmaf(y1, c[4], c[5]));
                    const float den = __builtin_fmaf(a0, a0, __builtin_fmaf(a1, a1, __builtin_fmaf(r0, r0, r1 * r1)));
                    const float d = T_f * den;
                    const float q = __builtin_fmaf(nn, nn, -d);
                    const float m = __builtin_fmaf(d, 0x1p-18f, __builtin_fmaf(__builtin_fabsf(nn), En2[k], C0[k]));
                    in[h] = __builtin_amdgcn_ballot_w64(q < -m) & valid[h];
                    open[h] = ~(in[h] | __builtin_amdgcn_ballot_w64(q > m)) & valid[h];
                }
            }
            count[k] += (uint32_t)__popcll(in[0]) + (uint32_t)__popcll(in[1]) + (uint32_t)__popcll(in[2]) + (uint32_t)__popcll(in[3]);
            if ((open[0] | open[1] | open[2] | open[3]) != 0ull) { // rare (never, when the screen is off: always): the reference's f64 expression
                double f[9], err;
#pragma unroll
                for (int i = 0; i < 9; i++) f[i] = F[(size_t)slot[k] * 9 + i];
                const uint32_t mine = (uint32_t)((open[0] >> lane) & 1ull) | (uint32_t)((open[1] >> lane) & 1ull) << 1 |
                                      (uint32_t)((open[2] >> lane) & 1ull) << 2 | (uint32_t)((open[3] >> lane) & 1ull) << 3;
#pragma unroll 1
                for (uint32_t h = 0; h < 4; h++) {
                    bool fits = false;
                    if ((mine >> h) & 1u) fits = match_fits(f, matches[base + 64u * h + lane], t, t_hi, err);
                    count[k] += (uint32_t)__popcll(__ballot(fits));
                }
            }
        }
        p1x = n1x, p1y = n1y, p2x = n2x, p2y = n2y;
        bound = max(bound, (uint32_t)__builtin_amdgcn_readfirstlane((int)running));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < COUNT_K; k++)
            if (j0 + k < n_hyp) {
                const uint32_t final_count = alive[k] ? count[k] : 0u;
                out_count[slot[k]] = final_count;
                out_err_sum[slot[k]] = 0.0;
                // (a look before the atomic: same-address atomics queue up in L2 - 25 000 of them cost round 0 a quarter of a
                // millisecond - and all but a few dozen hypotheses are below the maximum that is already there)
                if (cand && final_count >= max(min_count, 1u) &&
                    final_count >= __hip_atomic_load(&cand[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    const uint32_t seen = atomicMax(&cand[0], final_count);
                    if (final_count >= seen) {
                        const uint32_t at = atomicAdd(&cand[1], 1u);
                        if (at < TIED_CAP) {
                            cand[2 + 2 * at] = slot[k];
                            cand[3 + 2 * at] = final_count;
                        }
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The counting screen on the matrix pipe (round 5).  Per (hypothesis, match) pair the screen needs two numbers:
//   n   = p2' F p1             = sum_k F_k phi_k(match),     phi = (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1)
//   t d = t (a0^2 + a1^2 + r0^2 + r1^2) = sum_k G_k psi_k(match),  psi = (x1^2, x1 y1, y1^2, x1, y1, 1, x2^2, x2 y2, y2^2, x2, y2, 1)
// with G the twelve coefficients of the two quadratic forms (from F, in f64, times t) - two [hypotheses x 12] x [12 x matches]
// products, i.e. GEMM-shaped work: v_mfma_f32_16x16x4_f32, sixteen hypotheses x sixteen matches x four terms per instruction, six
// instructions per 256 pairs (192 matrix-pipe cycles per SIMD against ~350 vector-pipe cycles for the sixteen fmas per pair of
// ransac_count_kernel), leaving the vector pipe the decision: q = n^2 - t d against its error bound, two comparisons, one count.
// f32 MFMA accumulates with fma roundings in k order, so the bounds are those of a 9- / 12-term fma chain on exact operands
// (phi, psi: integers below 2^24 for coordinates below 4096; one more rounding each above) with f32-rounded coefficients:
//   |n_f32 - n| <= E_n = 14 u T_n,   T_n = (|F0|+|F1|+|F3|+|F4|) W^2 + (|F2|+|F5|+|F6|+|F7|) W + |F8|
//   |d_f32 - t d| <= E_d = 18 u T_d, T_d = the same sum over |G_k| psi_k's largest values
//   q = fma(n, n, -d):  |q_f32 - q| <= u |q_f32| + E_n (2 |n| + E_n) + E_d; the relative term cannot change q's sign, so with
//   m = (2 E_n |n| + E_n^2 + E_d) (1 + 2^-20):  q < -m certainly an inlier (then d > E_d: the exact denominator is positive),
//   q > m certainly none, anything else (NaN included) takes the reference's f64 expression (match_fits) - as before.
// The kernel walks only the head of the list: a hypothesis can be abandoned no earlier than match N - bound, most are soon after,
// and a wave holds sixteen of them - so phase 1 (this kernel) covers [0, L), L = N - bound rounded up + 512, for all live
// hypotheses, and the few that are still alive there continue in ransac_count_kernel from L with the count they have (phase 2;
// they are also the only ones that can reach the round's maximum, so the candidate logic stays there).
// ---------------------------------------------------------------------------------------------------------------
typedef float mf_float4 __attribute__((ext_vector_type(4)));
// phi / psi of the (re-sorted) match list, in the B-operand layout: per chunk of 16 matches six blocks of 64 floats, block s < 3:
// phi[4 s + (lane >> 4)] of match 16 c + (lane & 15), block 3 + s: psi likewise - one coalesced dword per lane and block
__global__ __launch_bounds__(256) void ransac_phi_kernel(const float *__restrict__ planes, uint32_t N, uint32_t np, float *__restrict__ phi)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, c = gid >> 6, l = gid & 63;
    if (c * 16u >= np) return;
    const uint32_t i = c * 16u + (l & 15u), kq = l >> 4;
    float x1 = 0.0f, y1 = 0.0f, x2 = 0.0f, y2 = 0.0f, one = 0.0f;
    if (i < N) {
        const uint32_t slot = ransac_plane_slot(i);
        x1 = planes[slot];
        y1 = planes[np + slot];
        x2 = planes[2u * np + slot];
        y2 = planes[3u * np + slot];
        one = 1.0f;
    }
    const float ph[3][4] = {{x2 * x1, x2 * y1, x2, y2 * x1}, {y2 * y1, y2, x1, y1}, {one, 0.0f, 0.0f, 0.0f}};
    const float ps[3][4] = {{x1 * x1, x1 * y1, y1 * y1, x1}, {y1, one, x2 * x2, x2 * y2}, {y2 * y2, x2, y2, one}};
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const float a = kq == 0 ? ph[s][0] : (kq == 1 ? ph[s][1] : (kq == 2 ? ph[s][2] : ph[s][3]));
        const float b = kq == 0 ? ps[s][0] : (kq == 1 ? ps[s][1] : (kq == 2 ? ps[s][2] : ps[s][3]));
        phi[((size_t)c * 6u + s) * 64u + l] = a;
        phi[((size_t)c * 6u + 3u + s) * 64u + l] = b;
    }
}

// a pair the screen left open: the reference's expression (out of line: sixteen call sites in a rarely taken branch)
__device__ __attribute__((noinline)) uint32_t mfma_open_pair(const double *__restrict__ F, uint32_t slot, uint4 mm, double t, double t_hi, bool mine)
{
    if (!mine) return 0u;
    double ff[9], err;
#pragma unroll
    for (int k = 0; k < 9; k++) ff[k] = F[(size_t)slot * 9 + k];
    return match_fits(ff, mm, t, t_hi, err) ? 1u : 0u;
}

__device__ __forceinline__ uint32_t row16_sum(uint32_t v) // sum over the 16 lanes of a DPP row, in every lane of the row
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2, 3, 0, 1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); // row_mirror
    return v;
}

// surv: [0] = survivors (atomic), [4 .. 4 + cap) their slots, [4 + cap .. 4 + 2 cap) their counts over [0, L)
__global__ __launch_bounds__(256) void ransac_count_mfma_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches, uint32_t N, double t,
                                                                 const uint32_t *__restrict__ live, const uint32_t *__restrict__ n_live,
                                                                 uint32_t min_count, const RansacBest *__restrict__ best,
                                                                 const uint32_t *__restrict__ coord_max, const float *__restrict__ phi,
                                                                 uint32_t *__restrict__ out_count, double *__restrict__ out_err_sum,
                                                                 uint32_t *__restrict__ surv, uint32_t surv_cap, uint32_t live_first, uint32_t live_end)
{
    const uint32_t n_hyp = min(*n_live, live_end), lane = threadIdx.x & 63;
    const uint32_t jb = live_first + blockIdx.x * 64u; // the workgroup's first hypothesis: its four waves take sixteen each
    const uint32_t j0 = jb + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 16u;
    if (jb >= n_hyp || N == 0) return; // (per WORKGROUP: a wave without hypotheses still takes part in the loads and barriers)
    const uint32_t bound = best->valid ? max(min_count, best->matches_count) : min_count;
    const uint32_t L = ransac_phase_len(N, bound);
    const uint32_t i = lane & 15u, kq = lane >> 4;
    const double t_hi = t * (1.0 + 0x1p-40);
    // ---- this lane's hypothesis in its role as an A-operand row: coefficients, bounds
    const bool have_i = j0 + i < n_hyp;
    const uint32_t slot_i = live[have_i ? j0 + i : jb];
    double f[9], af[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        f[k] = F[(size_t)slot_i * 9 + k];
        af[k] = fabs(f[k]);
    }
    const double W = (double)*coord_max, u = 0x1p-24;
    // t (a0^2 + a1^2): a0 = F0 x1 + F1 y1 + F2, a1 = F3 x1 + F4 y1 + F5;  t (r0^2 + r1^2): r0 = F0 x2 + F3 y2 + F6, r1 = F1 x2 + F4 y2 + F7
    double G[12];
    G[0] = t * (f[0] * f[0] + f[3] * f[3]);
    G[1] = t * (2.0 * (f[0] * f[1] + f[3] * f[4]));
    G[2] = t * (f[1] * f[1] + f[4] * f[4]);
    G[3] = t * (2.0 * (f[0] * f[2] + f[3] * f[5]));
    G[4] = t * (2.0 * (f[1] * f[2] + f[4] * f[5]));
    G[5] = t * (f[2] * f[2] + f[5] * f[5]);
    G[6] = t * (f[0] * f[0] + f[1] * f[1]);
    G[7] = t * (2.0 * (f[0] * f[3] + f[1] * f[4]));
    G[8] = t * (f[3] * f[3] + f[4] * f[4]);
    G[9] = t * (2.0 * (f[0] * f[6] + f[1] * f[7]));
    G[10] = t * (2.0 * (f[3] * f[6] + f[4] * f[7]));
    G[11] = t * (f[6] * f[6] + f[7] * f[7]);
    const double Tn = (af[0] + af[1] + af[3] + af[4]) * W * W + (af[2] + af[5] + af[6] + af[7]) * W + af[8];
    // (the expanded quadratic forms: every |G_k| psi_k is at most the square sums' own bound, signs ignored)
    const double Td = (fabs(G[0]) + fabs(G[1]) + fabs(G[2]) + fabs(G[6]) + fabs(G[7]) + fabs(G[8])) * W * W +
                      (fabs(G[3]) + fabs(G[4]) + fabs(G[9]) + fabs(G[10])) * W + fabs(G[5]) + fabs(G[11]);
    const double En = 1.001 * 14.0 * u * Tn, Ed = 1.001 * 18.0 * u * Td;
    const float En2_i = (float)(2.0 * En * (1.0 + 0x1p-20)), C0_i = (float)((En * En + Ed) * (1.0 + 0x1p-20));
    // the screen runs only where every bound and every product stays a NORMAL f32 number (false for NaN / inf coefficients)
    const bool screen_i = Tn >= 1e-6 && Tn <= 1e15 && Td >= 1e-12 && Td <= 1e30 && t >= 1e-6 && t <= 1e12 && W < 16777216.0;
    float an[3], ad[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        double vn = 0.0, vd = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((int)kq == k) {
                vn = 4 * s + k < 9 ? f[4 * s + k < 9 ? 4 * s + k : 0] : 0.0;
                vd = G[4 * s + k];
            }
        an[s] = (float)vn;
        ad[s] = (float)vd;
    }
    // ---- the four hypotheses whose results this lane holds (accumulator rows 4 kq + r): constants, flags, slots
    float En2r[4], C0r[4];
    uint32_t slot_r[4], cnt[4] = {0u, 0u, 0u, 0u};
    bool have_r[4];
    unsigned long long alive_m[4]; // lane masks (scalar registers): the row's hypothesis is still alive
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int src = (int)(4u * kq) + r;
        En2r[r] = __shfl(En2_i, src, 64);
        C0r[r] = __shfl(C0_i, src, 64);
        slot_r[r] = (uint32_t)__shfl((int)slot_i, src, 64);
        if (__shfl(screen_i ? 1 : 0, src, 64) == 0) C0r[r] = __builtin_inff(); // not screened: every pair of the row stays open
        have_r[r] = __shfl(have_i ? 1 : 0, src, 64) != 0;
        alive_m[r] = __builtin_amdgcn_ballot_w64(have_r[r]);
    }
    // Pairs the screen leaves open (a few per 10 000) are not decided where they turn up - the f64 expression with its loads
    // of the hypothesis would stall the wave every other trip - but parked on a small list per wave (row | match << 4) and
    // decided together when the walk is over; until then they count as POSSIBLE inliers wherever a hypothesis is tested for
    // abandonment (pendc), so nobody is abandoned who could still reach the bound.
    constexpr uint32_t PCAP = 224;
    __shared__ uint32_t s_pend[4][PCAP], s_pend_cnt[4][16];
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane < 16u) s_pend_cnt[wv][lane] = 0u;
    uint32_t pend_n = 0u; // (wave-uniform)
    uint32_t pendc[4] = {0u, 0u, 0u, 0u};
    // Four chunks (64 matches) per trip: their 24 operand dwords are loaded a trip ahead, the 24 MFMAs are eight independent
    // accumulator chains (no dependent-issue stalls), and the decision code is branch-free - masks are scalar.
    constexpr uint32_t CH = 4;
    // The four waves of a workgroup walk the SAME matches (different hypotheses), in step: a trip's operands (4 chunks x 6
    // blocks x 64 floats = 6 KB) are loaded once per workgroup - six dwords per thread, a trip ahead - and handed round
    // through a double buffer in LDS.  (Each wave loading its own: 24 loads per trip and wave, and Little's law on the L2
    // latency held the kernel at 44 % of the matrix pipe.)
    constexpr uint32_t TRIP_FLOATS = CH * 6u * 64u;
    // (a ring of three buffers: the loads issued at the top of trip t - the operands of trip t + 2 - have the whole trip to arrive
    // before they are written into the ring at the top of trip t + 1; with two buffers the trip itself had to outlast the L2
    // latency, and the matrix pipe stood at 46 %)
    __shared__ float s_b[3][TRIP_FLOATS];
    const uint32_t last_c = (L >> 4) - CH; // first chunk of the last trip
    float pre[6];
#pragma unroll
    for (int j = 0; j < 6; j++) s_b[0][threadIdx.x + 256u * j] = phi[threadIdx.x + 256u * j];
    {
        const float *const p1 = phi + (size_t)min(CH, last_c) * (6u * 64u);
#pragma unroll
        for (int j = 0; j < 6; j++) pre[j] = p1[threadIdx.x + 256u * j];
    }
    for (uint32_t base = 0, trip = 0, slot3 = 0; base < L; base += 16u * CH, trip++, slot3 = slot3 == 2u ? 0u : slot3 + 1u) {
        // (L is a multiple of 256, the same for every wave)
        const uint32_t next3 = slot3 == 2u ? 0u : slot3 + 1u;
#pragma unroll
        for (int j = 0; j < 6; j++) s_b[next3][threadIdx.x + 256u * j] = pre[j]; // trip + 1's operands (loaded during the last trip)
        const float *const pn = phi + (size_t)min((trip + 2u) * CH, last_c) * (6u * 64u); // (past the end: the last trip's again)
#pragma unroll
        for (int j = 0; j < 6; j++) pre[j] = pn[threadIdx.x + 256u * j];
        __syncthreads(); // this trip's buffer is complete (written a trip ago); the one being written was last read two trips ago
        float b[CH][6];
#pragma unroll
        for (uint32_t c = 0; c < CH; c++)
#pragma unroll
            for (int s = 0; s < 6; s++) b[c][s] = s_b[slot3][(c * 6u + s) * 64u + lane];
        mf_float4 accn[CH], accd[CH];
#pragma unroll
        for (uint32_t c = 0; c < CH; c++) {
            accn[c] = mf_float4{0.0f, 0.0f, 0.0f, 0.0f};
            accd[c] = mf_float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (uint32_t c = 0; c < CH; c++) {
                accn[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[s], b[c][s], accn[c], 0, 0, 0);
                accd[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[s], b[c][3 + s], accd[c], 0, 0, 0);
            }
        // The decisions, in vector registers only (six instructions per pair, no mask arithmetic on the scalar unit): the
        // count takes every certain inlier - a column beyond the list has n = d = 0 and is never one, a dead row's count is
        // never read, an unscreened row has m = inf -, and one bit per pair notes what is open; which of those matter (the
        // row alive, the column inside the list) is sorted out only in the trips that have any.
        uint32_t openbits = 0u; // bit 15 - (4 c + r): this lane's pair (chunk c, row r) is open
#pragma unroll
        for (uint32_t c = 0; c < CH; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float nn = accn[c][r], d = accd[c][r];
                const float q = __builtin_fmaf(nn, nn, -d);
                const float m = __builtin_fmaf(__builtin_fabsf(nn), En2r[r], C0r[r]);
                asm("v_cmp_lt_f32 vcc, %1, -%2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt[r]) : "v"(q), "v"(m) : "vcc");
                asm("v_cmp_ngt_f32 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(openbits) : "v"(q), "v"(m) : "vcc");
            }
        if (__builtin_amdgcn_ballot_w64(openbits != 0u)) {
#pragma unroll
            for (uint32_t c = 0; c < CH; c++) {
                const unsigned long long colv = __builtin_amdgcn_ballot_w64(base + 16u * c + i < N);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned long long om = __builtin_amdgcn_ballot_w64(((openbits >> (15u - (4u * c + (uint32_t)r))) & 1u) != 0u) & colv & alive_m[r];
                    if (!om) continue;
                    const uint32_t n_open = (uint32_t)__builtin_popcountll(om);
                    const bool mine = ((om >> lane) & 1ull) != 0ull;
                    const uint32_t mi = base + 16u * c + i;
                    if (pend_n + n_open <= PCAP) {
                        const uint32_t at = pend_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
                        if (mine) s_pend[wv][at] = (4u * kq + (uint32_t)r) | (mi << 4);
                        asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(pendc[r]) : "s"(om) : "vcc");
                        pend_n += n_open;
                    } else { // the list is full (never seen): decided on the spot
                        cnt[r] += mfma_open_pair(F, slot_r[r], matches[mi < N ? mi : 0u], t, t_hi, mine);
                    }
                }
            }
        }
        if (((base >> 4) & 15u) == 16u - CH) { // every 256 matches: who cannot reach the bound any more?
            unsigned long long any_alive = 0ull;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t tot = row16_sum(cnt[r] + pendc[r]); // (the parked pairs as if they were all inliers)
                alive_m[r] &= ~__builtin_amdgcn_ballot_w64(tot + (N - min(N, base + 16u * CH)) < bound);
                any_alive |= alive_m[r];
            }
            // (the workgroup leaves together: its waves share the operand stream and its barriers)
            if (!__syncthreads_or(any_alive != 0ull ? 1 : 0)) break;
        }
    }
    // ---- the parked pairs, one per lane: the reference's f64 expression (match_fits), tallied per row in LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t e0 = 0; e0 < pend_n; e0 += 64u) {
        const bool mine = e0 + lane < pend_n;
        const uint32_t entry = mine ? s_pend[wv][e0 + lane] : 0u;
        const uint32_t row = entry & 15u, mi = entry >> 4;
        const uint32_t slot = (uint32_t)__shfl((int)slot_i, (int)row, 64);
        if (mine) {
            double ff[9], err;
#pragma unroll
            for (int k = 0; k < 9; k++) ff[k] = F[(size_t)slot * 9 + k];
            if (match_fits(ff, matches[mi < N ? mi : 0u], t, t_hi, err)) atomicAdd(&s_pend_cnt[wv][row], 1u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t tot = row16_sum(cnt[r]) + s_pend_cnt[wv][4u * kq + (uint32_t)r];
        if (i == 0u && have_r[r]) {
            if (!((alive_m[r] >> lane) & 1ull)) {
                out_count[slot_r[r]] = 0u;
                out_err_sum[slot_r[r]] = 0.0;
            } else {
                const uint32_t at = atomicAdd(&surv[0], 1u);
                if (at < surv_cap) {
                    surv[4u + at] = slot_r[r];
                    surv[4u + surv_cap + at] = tot;
                }
            }
        }
    }
}

// the round's largest count, and the list of the live hypotheses that have it: tied[0] = their number, then the slots
__global__ __launch_bounds__(1024) void ransac_round_max_kernel(const uint32_t *__restrict__ counts,
                                                                 const uint32_t *__restrict__ live,
                                                                 const uint32_t *__restrict__ n_live, uint32_t min_count,
                                                                 uint32_t *__restrict__ tied)
{
    __shared__ uint32_t wmax[16];
    __shared__ uint32_t s_max, s_n;
    uint32_t m = 0;
    const uint32_t n_hyp = *n_live;
    for (uint32_t j = threadIdx.x; j < n_hyp; j += 1024) m = max(m, counts[live[j]]);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (uint32_t)__shfl_down(m, sft, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) m = max(m, wmax[w]);
        s_max = m;
        s_n = 0;
    }
    __syncthreads();
    const uint32_t top = s_max;
    if (top >= min_count && top > 0)
        for (uint32_t j = threadIdx.x; j < n_hyp; j += 1024) {
            const uint32_t h = live[j];
            if (counts[h] == top) {
                const uint32_t k = atomicAdd(&s_n, 1u);
                if (k < TIED_CAP) tied[1 + k] = h;
            }
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        tied[0] = s_n; // > TIED_CAP (never seen): the sum kernel then scans the live list for counts == top itself
        tied[1 + TIED_CAP] = top;
    }
}

// The reference's fold for the hypotheses at the round's maximum: errors in parallel, added serially in match order
// (an outlier contributes +0.0, which leaves the sum's bits unchanged: the sum is a sum of non-negative terms).
// Ord (:623-649) looks at the error only between hypotheses of EQUAL count, so the fold - 29 000 dependent f64
// additions, ~0.25 ms - runs only when it can decide something: several hypotheses share the round's maximum, or the
// maximum equals the count of the best hypothesis of the earlier rounds (whose own sum is then computed too, from
// best->f, if it never was).  In every other round the kernel returns at once and the winner's error stays unknown.
__device__ __forceinline__ bool ransac_round_needs_errors(const uint32_t *__restrict__ tied, const RansacBest *best)
{
    return tied[0] > 1u || (best->valid && tied[1 + TIED_CAP] == best->matches_count);
}
__global__ __launch_bounds__(1024) void ransac_tied_sum_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches,
                                                                uint32_t N, double t, const uint32_t *__restrict__ tied,
                                                                const uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ live,
                                                                const uint32_t *__restrict__ n_live, RansacBest *best,
                                                                double *__restrict__ out_err_sum)
{
    __shared__ double errs[1024];
    if (tied[0] == 0u || !ransac_round_needs_errors(tied, best)) return;
    const bool listed = tied[0] <= TIED_CAP;
    const uint32_t n_items = listed ? tied[0] : *n_live, top = tied[1 + TIED_CAP];
    // item n_items: the best hypothesis of the earlier rounds, when it ties with this round's maximum
    const uint32_t n_all = n_items + ((best->valid && top == best->matches_count) ? 1u : 0u);
    for (uint32_t b = blockIdx.x; b < n_all; b += gridDim.x) {
        const bool carried = b == n_items;
        if (carried && best->err_known) continue; // (only this workgroup writes err_known)
        const uint32_t h = carried ? 0u : (listed ? tied[1 + b] : live[b]);
        if (!carried && !listed && counts[h] != top) continue; // (uniform per workgroup)
        double f[9];
#pragma unroll
        for (int i = 0; i < 9; i++) f[i] = carried ? best->f[i] : F[(size_t)h * 9 + i];
        const double t_hi = t * (1.0 + 0x1p-40);
        double sum = 0.0;
        for (uint32_t base = 0; base < N; base += 1024) {
            const uint32_t i = base + threadIdx.x;
            double err = 0.0;
            const bool in = i < N && match_fits(f, matches[i], t, t_hi, err);
            __syncthreads();
            errs[threadIdx.x] = in ? err : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) {
#pragma unroll 8
                for (uint32_t q = 0; q < 1024; q++) sum += errs[q];
            }
        }
        if (threadIdx.x == 0) {
            if (carried) {
                best->best_error = sum / (double)best->matches_count;
                best->err_known = 1u;
            } else {
                out_err_sum[h] = sum;
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ bool ransac_better(uint32_t ca, double ea, uint32_t cb, double eb); // Ord, defined below

// ---- the device loops' form of the tie-break: parallel sums first, the ordered fold only where it decides ----------
// Ord (:623-649) compares the MEAN ERRORS of two hypotheses only when their counts are equal, and all that matters is
// which of the two serial sums is smaller.  A parallel sum S' of the same non-negative terms (each thread its strided
// share, then a fixed tree) and the reference's left-to-right sum S both lie within n u S (n = 29 000 terms, u = 2^-53:
// 3.3e-12 relative) of the exact sum, so S'_a < S'_b (1 - 1e-10) implies S_a < S_b.  ransac_tied_approx_kernel computes
// S' for the hypotheses at the round's maximum count (and for the carried best where it ties with them) with the whole
// GPU instead of one thread's 29 000 dependent additions; ransac_pick_best_approx_kernel orders them by S' and computes
// the reference's ordered fold - block_ordered_sum, the same fold as ransac_tied_sum_kernel - only for candidates whose
// S' lie within 1e-10 of the smallest (exact duplicates of a sample, in practice).
__device__ double block_parallel_sum(const double (&f)[9], const uint4 *__restrict__ matches, uint32_t N, double t, double *scratch16)
{
    const double t_hi = t * (1.0 + 0x1p-40);
    double part = 0.0;
    for (uint32_t i = threadIdx.x; i < N; i += 1024) {
        double err = 0.0;
        if (match_fits(f, matches[i], t, t_hi, err)) part += err;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_down(part, sft, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch16[threadIdx.x >> 6] = part;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < 16; w++) total += scratch16[w];
    return total; // (every thread returns the same value)
}
// the reference's fold (errors in parallel, added left to right by one thread); every thread returns the sum
__device__ double block_ordered_sum(const double (&f)[9], const uint4 *__restrict__ matches, uint32_t N, double t, double *errs1024,
                                    double *result)
{
    const double t_hi = t * (1.0 + 0x1p-40);
    double sum = 0.0;
    for (uint32_t base = 0; base < N; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        double err = 0.0;
        const bool in = i < N && match_fits(f, matches[i], t, t_hi, err);
        __syncthreads();
        errs1024[threadIdx.x] = in ? err : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll 8
            for (uint32_t q = 0; q < 1024; q++) sum += errs1024[q];
        }
    }
    if (threadIdx.x == 0) *result = sum;
    __syncthreads();
    return *result;
}

__device__ void ransac_tied_approx(const double *__restrict__ F, const uint4 *__restrict__ matches, uint32_t N, double t,
                                   const uint32_t *tied, RansacBest *best, double *__restrict__ out_err_sum, uint32_t block,
                                   uint32_t blocks)
{
    __shared__ double scratch[16];
    if (tied[0] == 0u || tied[0] > TIED_CAP || !ransac_round_needs_errors(tied, best)) return;
    const uint32_t n_items = tied[0], top = tied[1 + TIED_CAP];
    const uint32_t n_all = n_items + ((best->valid && top == best->matches_count && !best->err_known) ? 1u : 0u);
    for (uint32_t b = block; b < n_all; b += blocks) {
        const bool carried = b == n_items;
        const uint32_t h = carried ? 0u : tied[1 + b];
        double f[9];
#pragma unroll
        for (int i = 0; i < 9; i++) f[i] = carried ? best->f[i] : F[(size_t)h * 9 + i];
        const double sum = block_parallel_sum(f, matches, N, t, scratch);
        if (threadIdx.x == 0) {
            if (carried) {
                best->best_error = sum / (double)best->matches_count;
                best->err_known = 1u; // 1 = from a parallel sum, 2 = the reference's ordered fold
            } else {
                out_err_sum[h] = sum;
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void ransac_tied_approx_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches,
                                                                   uint32_t N, double t, const uint32_t *__restrict__ tied,
                                                                   RansacBest *best, double *__restrict__ out_err_sum)
{
    ransac_tied_approx(F, matches, N, t, tied, best, out_err_sum, blockIdx.x, gridDim.x);
}

// The counting kernel's copy of the match list, reordered whenever the best hypothesis changes: the matches the best
// hypothesis REJECTS first, then the ones it accepts - those it fits worst in front.  Counting is order-free, and a
// hypothesis is abandoned as soon as its misses exceed N - (best count).  With the list in the matcher's order a
// hypothesis that is nearly as good as the best one (most are: an epipolar constraint that is roughly right accepts most
// true correspondences) collects those misses over ~0.7 of the list; with the best hypothesis' outliers in front, which
// nearly every hypothesis misses, and its marginal inliers next, it has used up its allowance at 0.37 - the floor
// is (N - best count) / N = 0.355 on config 5's pairs (scripts/ransac_order_study.py).  The winner and every hypothesis
// that ties it are never abandoned, so the round's result is unchanged; the tie-break sums run over the list in its
// original order.  Nothing here has to be exact - any order counts the same - so the classes come from an f32
// evaluation, a counting sort over REORDER_CLASSES error classes with the order inside a class as the LDS atomics fall,
// and the list is only reordered again when the best count has grown by 1/64 since the last time.
// One workgroup of 1024 threads; scratch: 2 * REORDER_CLASSES words of LDS.
constexpr uint32_t REORDER_CLASSES = 9; // rejected; accepted with err / t in (7/8, 1], (6/8, 7/8], ..., [0, 1/8]
__device__ void ransac_reorder_matches(const double *best_f, const uint4 *__restrict__ matches, uint32_t N, double t,
                                       uint4 *__restrict__ order_u32, float *__restrict__ order_planes, uint32_t *scratch)
{
    float c[9];
#pragma unroll
    for (int i = 0; i < 9; i++) c[i] = (float)best_f[i];
    const float tf = (float)t;
    uint32_t *const total = scratch, *const cursor = scratch + REORDER_CLASSES;
    if (threadIdx.x < 2 * REORDER_CLASSES) scratch[threadIdx.x] = 0u;
    __syncthreads();
    const auto error_class = [&](uint4 v) -> uint32_t {
        const float x1 = (float)v.x, y1 = (float)v.y, x2 = (float)v.z, y2 = (float)v.w;
        const float r0 = __builtin_fmaf(x2, c[0], __builtin_fmaf(y2, c[3], c[6]));
        const float r1 = __builtin_fmaf(x2, c[1], __builtin_fmaf(y2, c[4], c[7]));
        const float r2 = __builtin_fmaf(x2, c[2], __builtin_fmaf(y2, c[5], c[8]));
        const float nn = __builtin_fmaf(r0, x1, __builtin_fmaf(r1, y1, r2));
        const float a0 = __builtin_fmaf(x1, c[0], __builtin_fmaf(y1, c[1], c[2]));
        const float a1 = __builtin_fmaf(x1, c[3], __builtin_fmaf(y1, c[4], c[5]));
        const float den = __builtin_fmaf(a0, a0, __builtin_fmaf(a1, a1, __builtin_fmaf(r0, r0, r1 * r1)));
        const float x = nn * nn, y = tf * den;
        if (!(x <= y)) return 0u; // rejected (or not a number)
        const float r = __fdividef(x, y); // in [0, 1], NaN for 0 / 0
        return 1u + min(7u, (uint32_t)fmaxf((1.0f - r) * 8.0f, 0.0f));
    };
    for (uint32_t i = threadIdx.x; i < N; i += 1024u) atomicAdd(&total[error_class(matches[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { // class c starts where the classes before it end
        uint32_t run = 0;
        for (uint32_t k = 0; k < REORDER_CLASSES; k++) {
            cursor[k] = run;
            run += total[k];
        }
    }
    __syncthreads();
    const uint32_t np = ransac_padded(N);
    for (uint32_t i = threadIdx.x; i < N; i += 1024u) {
        const uint4 v = matches[i];
        const uint32_t pos = atomicAdd(&cursor[error_class(v)], 1u);
        order_u32[pos] = v;
        const uint32_t slot = ransac_plane_slot(pos);
        order_planes[slot] = (float)v.x;
        order_planes[np + slot] = (float)v.y;
        order_planes[2u * np + slot] = (float)v.z;
        order_planes[3u * np + slot] = (float)v.w;
    }
}

__device__ void ransac_pick_best_approx(const double *__restrict__ F, const uint4 *__restrict__ matches, uint32_t N, double t,
                                        double *__restrict__ err_sums, uint32_t min_count, const uint32_t *tied,
                                        RansacBest *best, uint4 *__restrict__ order_u32, float *__restrict__ order_planes,
                                        uint32_t slot_base = 0u)
{
    __shared__ double errs[1024];
    __shared__ uint32_t s_replaced;
    __shared__ double s_result;
    __shared__ uint32_t s_close[64]; // candidates within the margin of the smallest parallel sum (slots; ~0u = the carried best)
    __shared__ uint32_t s_nclose, s_winner, s_exact;
    __shared__ double s_winner_err;
    const uint32_t n = tied[0], top = tied[1 + TIED_CAP];
    if (n == 0u || n > TIED_CAP || top < min_count) return; // (more than TIED_CAP ties: never seen; the round is then skipped)
    const bool errors = ransac_round_needs_errors(tied, best); // were this round's sums computed?
    const bool carried_ties = best->valid && top == best->matches_count;
    constexpr double MARGIN = 1e-10;
    if (threadIdx.x == 0) {
        // smallest parallel sum among the round's candidates (smaller slot first among equals)
        uint32_t bi = 0xFFFFFFFFu;
        double be = 0.0;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t h = tied[1 + k];
            const double e = errors ? err_sums[h] / (double)top : 0.0;
            const bool ef = fabs(e) < __builtin_inf(), bf = fabs(be) < __builtin_inf();
            if (bi == 0xFFFFFFFFu || (ef && !bf) || (ef == bf && (e < be || (e == be && h < bi)))) {
                bi = h;
                be = e;
            }
        }
        s_winner = bi;
        s_winner_err = be;
        s_exact = 0u;
        uint32_t nc = 0;
        if (errors) {
            // who else is within the margin of it - the other candidates of the round, and the carried best
            const double lim = be + MARGIN * fabs(be);
            for (uint32_t k = 0; k < n && nc < 63u; k++) {
                const uint32_t h = tied[1 + k];
                if (h != bi && err_sums[h] / (double)top <= lim) s_close[nc++] = h;
            }
            if (carried_ties && fabs(best->best_error - be) <= MARGIN * fmax(fabs(best->best_error), fabs(be)) && best->err_known != 2u)
                s_close[nc++] = 0xFFFFFFFFu;
            if (nc > 0) s_close[nc++] = bi; // the winner itself needs its ordered sum too
        }
        s_nclose = nc;
    }
    __syncthreads();
    if (s_nclose > 0) { // (rare) the reference's ordered fold for everybody inside the margin
        for (uint32_t k = 0; k < s_nclose; k++) {
            const uint32_t h = s_close[k];
            double f[9];
#pragma unroll
            for (int i = 0; i < 9; i++) f[i] = h == 0xFFFFFFFFu ? best->f[i] : F[(size_t)h * 9 + i];
            const double sum = block_ordered_sum(f, matches, N, t, errs, &s_result);
            if (threadIdx.x == 0) {
                if (h == 0xFFFFFFFFu) {
                    best->best_error = sum / (double)best->matches_count;
                    best->err_known = 2u;
                } else {
                    err_sums[h] = sum;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { // the winner among them by the ordered sums
            uint32_t bi = 0xFFFFFFFFu;
            double be = 0.0;
            for (uint32_t k = 0; k < s_nclose; k++) {
                const uint32_t h = s_close[k];
                if (h == 0xFFFFFFFFu) continue;
                const double e = err_sums[h] / (double)top;
                if (bi == 0xFFFFFFFFu || e < be || (e == be && h < bi)) {
                    bi = h;
                    be = e;
                }
            }
            s_winner = bi;
            s_winner_err = be;
            s_exact = 1u;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t w = s_winner;
        s_replaced = 0u;
        // (equal under Ord - same count, and error sums that the ordered fold made bit-equal, or both not finite: the one
        // met first in the reference's iteration order stays; only an out-of-order caller can bring an earlier one later)
        const bool better = !best->valid || ransac_better(top, s_winner_err, best->matches_count, best->best_error);
        const bool equal = best->valid && !better && !ransac_better(best->matches_count, best->best_error, top, s_winner_err);
        if (better || (equal && slot_base + w < best->origin)) {
            for (int i = 0; i < 9; i++) best->f[i] = F[(size_t)w * 9 + i];
            best->matches_count = top;
            best->best_error = s_winner_err;
            best->valid = 1;
            best->err_known = errors ? (s_exact ? 2u : 1u) : 0u;
            best->origin = slot_base + w;
            s_replaced = 1u;
        }
    }
    __syncthreads();
    // (best->pad: the best count the counting kernel's list was last ordered by; 0 = not yet)
    if (s_replaced && order_u32 && (best->pad == 0u || top >= best->pad + max(best->pad >> 6, 1u))) {
        ransac_reorder_matches(best->f, matches, N, t, order_u32, order_planes, reinterpret_cast<uint32_t *>(errs));
        if (threadIdx.x == 0) best->pad = top;
    }
}
__global__ __launch_bounds__(1024) void ransac_pick_best_approx_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches,
                                                                        uint32_t N, double t, const uint32_t *__restrict__ counts,
                                                                        double *__restrict__ err_sums, uint32_t min_count,
                                                                        const uint32_t *__restrict__ tied, RansacBest *best,
                                                                        uint4 *__restrict__ order_u32, float *__restrict__ order_planes,
                                                                        uint32_t slot_base)
{
    (void)counts;
    ransac_pick_best_approx(F, matches, N, t, err_sums, min_count, tied, best, order_u32, order_planes, slot_base);
}

// The device loops' round end in ONE launch behind the counting kernel: the round's maximum list from the counting
// kernel's candidates (its own atomic maximum, and whoever reached it), the tie-break sums where they decide, the pick, the
// reordering of the counting list.  Three launches and their gaps less on a chain of twenty rounds (~30 us each).
// cand: as ransac_count_kernel leaves it, cleared here for the next round; tied: [2 + TIED_CAP] words of workspace.
// the round's maximum list from the counting kernel's candidates; clears `cand` for the next round.  One workgroup.
__device__ void ransac_round_tied_list(uint32_t *__restrict__ cand, uint32_t *tied)
{
    // (the candidates that reached the final maximum, picked out by all threads - one thread alone pays a dependent global
    // load per candidate - then ordered by slot: the list's order does not depend on who finished first)
    __shared__ uint32_t s_n, s_slots[TIED_CAP];
    const uint32_t top = cand[0], listed = cand[1];
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    if (listed <= TIED_CAP)
        for (uint32_t i = threadIdx.x; i < listed; i += blockDim.x)
            if (cand[3 + 2 * i] == top) s_slots[atomicAdd(&s_n, 1u)] = cand[2 + 2 * i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        if (listed > TIED_CAP) {
            n = TIED_CAP + 1u; // (never seen: more record holders than the list takes; the round is skipped, as above)
        } else {
            for (uint32_t i = 0; i < s_n; i++) { // insertion by slot
                const uint32_t h = s_slots[i];
                uint32_t at = n++;
                for (; at > 0 && tied[at] > h; at--) tied[1 + at] = tied[at];
                tied[1 + at] = h;
            }
        }
        tied[0] = n;
        tied[1 + TIED_CAP] = top;
        cand[0] = 0u;
        cand[1] = 0u;
    }
}
__global__ __launch_bounds__(1024) void ransac_round_finish_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches,
                                                                    uint32_t N, double t, double *__restrict__ err_sums,
                                                                    uint32_t min_count, uint32_t *__restrict__ cand,
                                                                    uint32_t *tied, RansacBest *best,
                                                                    uint4 *__restrict__ order_u32, float *__restrict__ order_planes,
                                                                    uint32_t slot_base)
{
    ransac_round_tied_list(cand, tied);
    __syncthreads();
    ransac_tied_approx(F, matches, N, t, tied, best, err_sums, 0u, 1u);
    __syncthreads();
    ransac_pick_best_approx(F, matches, N, t, err_sums, min_count, tied, best, order_u32, order_planes, slot_base);
}
// the same in three launches, for rounds with several hypotheses at the maximum count (a batch of rounds scored as one:
// each costs ~10 us of one workgroup's time in the kernel above, ~100 us per batch): the list, the sums spread over
// workgroups (20 - 35 us), the pick
__global__ __launch_bounds__(1024) void ransac_round_tied_list_kernel(uint32_t *__restrict__ cand, uint32_t *tied)
{
    ransac_round_tied_list(cand, tied);
}

// The live slots of a round's hypothesis buffer, in slot order (count, scan, scatter).  Depends on the hypotheses only,
// so the device loops run it on the GENERATOR's stream right behind the generation - off the scoring chain.
// scratch: ceil(H / 1024) words.
// workspace of the matrix-pipe counting phase (ransac_count_mfma_kernel): the B operands of the match list (24 floats per match of
// the padded list) and the survivor list [4 + 2 cap] (cap = the most live hypotheses one scoring launch can see)
struct CountMfmaWs {
    float *phi = nullptr;
    uint32_t *surv = nullptr;
    uint32_t cap = 0;
};
template <typename Mem> static hipError_t alloc_count_mfma(Mem &mem, uint32_t N, uint32_t cap, CountMfmaWs &ws)
{
    hipError_t e = mem.alloc(&ws.phi, (size_t)ransac_padded(std::max(N, 1u)) * 24u);
    if (e == hipSuccess) e = mem.alloc(&ws.surv, 4 + 2 * (size_t)cap);
    ws.cap = cap;
    return e;
}
static void launch_ransac_live(const double *F, uint32_t H, uint32_t *live, uint32_t *n_live, uint32_t *scratch, hipStream_t s)
{
    const uint32_t nblocks = (H + 1023) / 1024;
    hipLaunchKernelGGL(ransac_live_count_kernel, dim3(nblocks), dim3(1024), 0, s, F, H, scratch);
    launch_scan_u32(scratch, nblocks, n_live, s);
    hipLaunchKernelGGL(ransac_live_scatter_kernel, dim3(nblocks), dim3(1024), 0, s, F, H, (const uint32_t *)scratch, live);
}

// tied: [2 + TIED_CAP] words (number, slots, the maximum itself); coord_max: one word (ransac_coord_max_kernel);
// live_ready: the live list was already built (launch_ransac_live)
static void launch_ransac_score_round(const double *F, uint32_t H, const uint32_t *matches, const uint32_t *count_matches, const float4 *matches_f32,
                                      uint32_t N, double t, uint32_t *live, uint32_t *n_live, uint32_t *tied,
                                      const uint32_t *coord_max, bool live_ready, bool approx_sums, uint32_t min_count,
                                      RansacBest *best, uint32_t *out_count, double *out_err_sum, hipStream_t s, uint32_t *cand = nullptr,
                                      uint32_t live_first = 0, uint32_t live_end = 0xFFFFFFFFu, const CountMfmaWs *ws = nullptr)
{
    const uint4 *m4 = reinterpret_cast<const uint4 *>(matches);
    // (the device loops read the counts through the live list only, and the counting kernel writes every live slot when
    // there is a match at all: only the entry point that hands ALL counts back needs the others zeroed)
    if (!live_ready || N == 0) (void)hipMemsetAsync(out_count, 0, (size_t)H * sizeof(uint32_t), s);
    // (scratch of the compaction: out_err_sum's first words, until the count kernel overwrites them)
    if (!live_ready) launch_ransac_live(F, H, live, n_live, reinterpret_cast<uint32_t *>(out_err_sum), s);
    // (grids are sized for the case that every slot is live; waves / workgroups beyond *n_live leave at once)
    // (count_matches / matches_f32: the counting kernel's own copy of the list - same matches, any order)
    const uint32_t span = std::min(H, live_end) > live_first ? std::min(H, live_end) - live_first : 0u;
    if (span && ws && ws->phi && N > 0 && span <= ws->cap) {
        // phase 1 on the matrix pipe (every live hypothesis over the head of the list), phase 2 for whoever is still alive
        const uint32_t np = ransac_padded(N);
        (void)hipMemsetAsync(ws->surv, 0, 4 * sizeof(uint32_t), s);
        hipLaunchKernelGGL(ransac_phi_kernel, dim3((np * 4u + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const float *>(matches_f32), N, np, ws->phi);
        hipLaunchKernelGGL(ransac_count_mfma_kernel, dim3((span + 63u) / 64u), dim3(256), 0, s, F, reinterpret_cast<const uint4 *>(count_matches), N, t,
                           (const uint32_t *)live, (const uint32_t *)n_live, min_count, (const RansacBest *)best, coord_max, (const float *)ws->phi,
                           out_count, out_err_sum, ws->surv, ws->cap, live_first, live_end);
        hipLaunchKernelGGL(ransac_count_kernel, dim3((span + 4 * COUNT_K - 1) / (4 * COUNT_K)), dim3(256), 0, s, F,
                           reinterpret_cast<const uint4 *>(count_matches), N, t, (const uint32_t *)(ws->surv + 4), (const uint32_t *)ws->surv, min_count,
                           (const RansacBest *)best, coord_max, matches_f32, out_count, out_err_sum, cand, 0u, 0xFFFFFFFFu,
                           (const uint32_t *)(ws->surv + 4 + ws->cap));
    } else if (span)
        hipLaunchKernelGGL(ransac_count_kernel, dim3((span + 4 * COUNT_K - 1) / (4 * COUNT_K)), dim3(256), 0, s, F,
                           reinterpret_cast<const uint4 *>(count_matches), N, t, (const uint32_t *)live, (const uint32_t *)n_live, min_count,
                           (const RansacBest *)best, coord_max, matches_f32, out_count, out_err_sum, cand, live_first, live_end);
    if (cand) return; // the device loops: ransac_round_finish_kernel takes it from the candidates
    hipLaunchKernelGGL(ransac_round_max_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t *)out_count, (const uint32_t *)live,
                       (const uint32_t *)n_live, min_count, tied);
    if (approx_sums) // the device loops: parallel sums now, the ordered fold inside ransac_pick_best_approx_kernel where it decides
        hipLaunchKernelGGL(ransac_tied_approx_kernel, dim3(16), dim3(1024), 0, s, F, m4, N, t, (const uint32_t *)tied, best, out_err_sum);
    else
        hipLaunchKernelGGL(ransac_tied_sum_kernel, dim3(16), dim3(1024), 0, s, F, m4, N, t, (const uint32_t *)tied,
                           (const uint32_t *)out_count, (const uint32_t *)live, (const uint32_t *)n_live, best, out_err_sum);
}

// ---------------------------------------------------------------------------------------------
// Whole affine RANSAC on the device: hypothesis generation moves next
// to the scoring kernel, so the 10^6-iteration loop of FundamentalMatrix::find_ransac
// (fundamentalmatrix.rs:103-147) never leaves the GPU except for one 4-byte early-exit check per
// 50 000-iteration round.  Per hypothesis (one thread): choose_inliers (:155-175, rejection
// sampling from the top 5000 matches, >= 10 px apart), calculate_model_affine (:260-286: mean-centred
// 4x4, right-singular vector of the smallest singular value — here via Jacobi on A^T A), validate_f's
// finiteness and sample-fit checks (:197-209).  The reference seeds its RNG from the OS and is not
// reproducible run to run, so parity for this row is statistical (tests compare with the known model).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ bool affine_model_from_sample(const uint4 (&sm)[4], double (&f)[9])
{
    double a[4][4], mean[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) { // rows (x2, y2, x1, y1), fundamentalmatrix.rs:262-268
        a[i][0] = (double)sm[i].z;
        a[i][1] = (double)sm[i].w;
        a[i][2] = (double)sm[i].x;
        a[i][3] = (double)sm[i].y;
#pragma unroll
        for (int j = 0; j < 4; j++) mean[j] += a[i][j] / 4.0; // row_mean(): exact for integer coordinates
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) a[i][j] -= mean[j];
    // SVD of the centred 4x4 by one-sided (Hestenes) Jacobi: rotate pairs of COLUMNS of A until they are mutually
    // orthogonal; then A = U S with column norms = singular values and the accumulated rotations V.  Unlike an
    // eigen-decomposition of A'A this keeps the small singular directions to full relative accuracy.
    double v[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; sweep++) { // fixed pivot order (static indices)
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    alpha += a[k][p] * a[k][p];
                    beta += a[k][q] * a[k][q];
                    gamma += a[k][p] * a[k][q];
                }
                if (!(fabs(gamma) > 1e-300) || !(fabs(gamma) > 1e-17 * sqrt(alpha * beta))) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tt = __builtin_copysign(1.0, zeta) / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    double nrm[4];
#pragma unroll
    for (int j = 0; j < 4; j++) nrm[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j] + a[3][j] * a[3][j]);
    // smallest singular value -> null vector (last row of V'); the second largest must be >= 1e-3 (:272-275)
    int last = 0;
    double lo = nrm[0], hi1 = -1.0, hi2 = -1.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (nrm[i] < lo) {
            lo = nrm[i];
            last = i;
        }
        if (nrm[i] > hi1) {
            hi2 = hi1;
            hi1 = nrm[i];
        } else if (nrm[i] > hi2) {
            hi2 = nrm[i];
        }
    }
    if (hi2 < 0.001) return false;
    double vt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) vt[k] = last == 0 ? v[k][0] : (last == 1 ? v[k][1] : (last == 2 ? v[k][2] : v[k][3]));
    const double e = vt[0] * mean[0] + vt[1] * mean[1] + vt[2] * mean[2] + vt[3] * mean[3];
    const double raw[9] = {0.0, 0.0, vt[0], 0.0, 0.0, vt[1], vt[2], vt[3], -e};
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = raw[i] / raw[8]; // f / f[(2, 2)], :285
    return true;
}

__global__ __launch_bounds__(64) void ransac_generate_affine_kernel(const uint4 *__restrict__ matches, uint32_t limit,
                                                                     double t, unsigned long long seed,
                                                                     uint32_t round, uint32_t H,
                                                                     const uint32_t *__restrict__ sample_idx,
                                                                     double *__restrict__ F)
{
    const uint32_t h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    unsigned long long state = mix64(seed ^ mix64(((unsigned long long)round << 32) | h));
    uint4 sm[4];
    int have = 0;
    if (sample_idx) { // the caller's samples (test hook)
#pragma unroll
        for (int i = 0; i < 4; i++) sm[i] = matches[sample_idx[(size_t)h * 4 + i]];
        have = 4;
    }
    for (int tries = 0; tries < 256 && have < 4; tries++) { // choose_inliers, :155-175 (bounded here)
        state = mix64(state + 0x9E3779B97F4A7C15ull);
        const uint32_t idx = (uint32_t)(((state >> 32) * (unsigned long long)limit) >> 32);
        const uint4 nm = matches[idx];
        bool close = false;
        for (int i = 0; i < have; i++) {
            const uint4 c = sm[i];
            auto dist = [](uint32_t a, uint32_t b) { return a > b ? a - b : b - a; };
            close = close || dist(nm.x, c.x) < 10u || dist(nm.y, c.y) < 10u || dist(nm.z, c.z) < 10u || dist(nm.w, c.w) < 10u;
        }
        if (!close) {
            if (have == 0) sm[0] = nm;
            else if (have == 1) sm[1] = nm;
            else if (have == 2) sm[2] = nm;
            else sm[3] = nm;
            have++;
        }
    }
    double f[9];
    bool ok = have == 4 && affine_model_from_sample(sm, f);
    if (ok) {
#pragma unroll
        for (int i = 0; i < 9; i++) ok = ok && fabs(f[i]) < __builtin_inf(); // validate_f, :197-199
#pragma unroll
        for (int i = 0; i < 4; i++) { // all sample points must fit, :206-209
            const double err = reprojection_error(f, (double)sm[i].x, (double)sm[i].y, (double)sm[i].z, (double)sm[i].w);
            ok = ok && fabs(err) < __builtin_inf() && !(fabs(err) > t);
        }
    }
    const double nan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)h * 9 + i] = ok ? f[i] : nan; // NaN hypotheses score 0 inliers
}

// ---------------------------------------------------------------------------------------------
// Perspective model: hypothesis generation on the device.  Per sample (one thread):
// calculate_model_perspective (fundamentalmatrix.rs:289-389) - the two-dimensional null space of the
// 7x9 epipolar system, the cubic det(a F1 + (1 - a) F2) = 0, and per real root the reference's rank test
// (second singular value >= 1e-3, third <= 1e-3), normalisation by F[2][2] and sign-consistency test -
// then validate_f's finiteness and sample-fit checks (:197-209).  The null space comes from a Householder
// QR of A^T (its last two Q columns) instead of an SVD: any basis of the null space gives the same pencil,
// hence the same F's.  validate_f's per-hypothesis optimize_perspective_f (:201-205) runs here too (lm:: above,
// n = 7), including its rank test on the re-parametrised matrix.  Statistical parity, as for the affine model;
// tests compare this generator with an independent numpy restatement (LAPACK underneath) on identical samples.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double (&m)[9])
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// real roots of c0 x^3 + c1 x^2 + c2 x + c3 (c0 != 0), polished by Newton steps; returns their number
__device__ int cubic_real_roots(double c0, double c1, double c2, double c3, double (&r)[3])
{
    const double b = c1 / c0, c = c2 / c0, d = c3 / c0;
    const double p = c - b * b / 3.0, q = 2.0 * b * b * b / 27.0 - b * c / 3.0 + d;
    const double disc = q * q / 4.0 + p * p * p / 27.0;
    int n;
    if (disc > 0.0) {
        const double sq = sqrt(disc);
        r[0] = cbrt(-q / 2.0 + sq) + cbrt(-q / 2.0 - sq) - b / 3.0;
        n = 1;
    } else if (p == 0.0) {
        r[0] = -b / 3.0;
        n = 1;
    } else {
        const double rr = 2.0 * sqrt(-p / 3.0);
        double arg = 3.0 * q / (p * rr);
        arg = fmin(1.0, fmax(-1.0, arg));
        const double phi = acos(arg) / 3.0;
#pragma unroll
        for (int k = 0; k < 3; k++) r[k] = rr * cos(phi - 2.0943951023931953 * (double)k) - b / 3.0;
        n = 3;
    }
    for (int k = 0; k < n; k++) {
        double x = r[k];
        for (int it = 0; it < 3; it++) {
            const double fx = ((c0 * x + c1) * x + c2) * x + c3, dfx = (3.0 * c0 * x + 2.0 * c1) * x + c2;
            if (dfx != 0.0 && fabs(fx) < __builtin_inf()) x -= fx / dfx;
        }
        r[k] = x;
    }
    return n;
}

// The pencil's basis as the reference WRITES it (fundamentalmatrix.rs:309-322): `a.svd(false, true)` on an
// SMatrix<f64, 7, 9> is nalgebra's THIN decomposition - v_t is DimMinimum<7, 9> x 9 = 7 x 9, singular values descending -
// so v_t.row(nrows - 2) and v_t.row(nrows - 1) are rows 5 and 6: the right singular vectors of the two SMALLEST of the
// seven singular values, not the null space of A (that would be rows 7 and 8 of a full V').  The pencil's members
// therefore have det 0 but do not fit the sample; validate_f's optimize_perspective_f (:201-205) starts from them.
// One-sided (Hestenes) Jacobi on W = A' (9 x 7): pairs of columns are rotated until mutually orthogonal, then
// W J = V S - column norms = singular values, normalised columns = right singular vectors of A, each to full relative
// accuracy (an eigen-decomposition of A A' would square a condition number of ~1e7).  n1 = the vector of the second
// smallest singular value (row 5), n2 = of the smallest (row 6).  SIGN: a singular vector is defined up to sign and
// nalgebra's choice falls out of its bidiagonalisation (source not in this image); the roots of the cubic in
// `a` and the scale the rank test :365-370 sees depend on it (projectively the pencil is the same).  Convention here
// (the CPU checker of the tests uses the same one): the entry of largest magnitude, the first of equals, is positive.
__device__ void perspective_basis_thin_svd(const uint4 (&sm)[7], double (&n1)[9], double (&n2)[9])
{
    double W[9][7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double x1 = (double)sm[i].x, y1 = (double)sm[i].y, x2 = (double)sm[i].z, y2 = (double)sm[i].w;
        W[0][i] = x2 * x1;
        W[1][i] = x2 * y1;
        W[2][i] = x2;
        W[3][i] = y2 * x1;
        W[4][i] = y2 * y1;
        W[5][i] = y2;
        W[6][i] = x1;
        W[7][i] = y1;
        W[8][i] = 1.0;
    }
    for (int sweep = 0; sweep < 30; sweep++) { // fixed pivot order (static indices: W stays in registers)
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 7; p++)
#pragma unroll
            for (int q = p + 1; q < 7; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    alpha += W[k][p] * W[k][p];
                    beta += W[k][q] * W[k][q];
                    gamma += W[k][p] * W[k][q];
                }
                if (!(fabs(gamma) > 1e-300) || !(fabs(gamma) > 1e-15 * sqrt(alpha * beta))) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tt = __builtin_copysign(1.0, zeta) / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const double wp = W[k][p], wq = W[k][q];
                    W[k][p] = c * wp - sn * wq;
                    W[k][q] = sn * wp + c * wq;
                }
            }
        if (!rotated) break;
    }
    double nrm[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) ss += W[k][j] * W[k][j];
        nrm[j] = sqrt(ss);
    }
    int i7 = 0; // smallest, then the smallest of the rest (a later column only replaces an earlier one if strictly smaller)
#pragma unroll
    for (int j = 1; j < 7; j++)
        if (nrm[j] < nrm[i7]) i7 = j;
    int i6 = i7 == 0 ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 7; j++)
        if (j != i7 && nrm[j] < nrm[i6]) i6 = j;
    double s6 = 0.0, s7 = 0.0;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        if (j == i6) s6 = nrm[j];
        if (j == i7) s7 = nrm[j];
    }
    double big1 = -1.0, big2 = -1.0, sg1 = 1.0, sg2 = 1.0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (j == i6) a = W[k][j];
            if (j == i7) b = W[k][j];
        }
        n1[k] = a / s6;
        n2[k] = b / s7;
        if (fabs(n1[k]) > big1) {
            big1 = fabs(n1[k]);
            sg1 = n1[k] < 0.0 ? -1.0 : 1.0;
        }
        if (fabs(n2[k]) > big2) {
            big2 = fabs(n2[k]);
            sg2 = n2[k] < 0.0 ? -1.0 : 1.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        n1[k] *= sg1;
        n2[k] *= sg2;
    }
}

// The textbook basis (cvhip_ransac_set_pencil(dev, CVHIP_PENCIL_NULL_SPACE)): the two-dimensional null space of A.
__device__ void perspective_basis_null_space(const uint4 (&sm)[7], double (&n1)[9], double (&n2)[9])
{
    // M = A^T (9 x 7), fundamentalmatrix.rs:293-309; Householder QR, reflectors kept in place
    double M[9][7], beta[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double x1 = (double)sm[i].x, y1 = (double)sm[i].y, x2 = (double)sm[i].z, y2 = (double)sm[i].w;
        M[0][i] = x2 * x1;
        M[1][i] = x2 * y1;
        M[2][i] = x2;
        M[3][i] = y2 * x1;
        M[4][i] = y2 * y1;
        M[5][i] = y2;
        M[6][i] = x1;
        M[7][i] = y1;
        M[8][i] = 1.0;
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        double nrm = 0.0;
#pragma unroll
        for (int r = k; r < 9; r++) nrm += M[r][k] * M[r][k];
        nrm = sqrt(nrm);
        const double alpha = M[k][k] >= 0.0 ? -nrm : nrm;
        M[k][k] -= alpha; // v = x - alpha e_k, stored in rows k..8 of column k
        double vv = 0.0;
#pragma unroll
        for (int r = k; r < 9; r++) vv += M[r][k] * M[r][k];
        beta[k] = vv > 0.0 ? 2.0 / vv : 0.0;
#pragma unroll
        for (int c = k + 1; c < 7; c++) {
            double dot = 0.0;
#pragma unroll
            for (int r = k; r < 9; r++) dot += M[r][k] * M[r][c];
            dot *= beta[k];
#pragma unroll
            for (int r = k; r < 9; r++) M[r][c] -= dot * M[r][k];
        }
    }
    // null space of A = last two columns of Q = H0 H1 ... H6 applied to e7, e8: NOT what fundamentalmatrix.rs:309-322
    // takes (see perspective_basis_thin_svd) - the 7-point algorithm as published, whose pencil fits the sample exactly.
#pragma unroll
    for (int r = 0; r < 9; r++) {
        n1[r] = r == 7 ? 1.0 : 0.0;
        n2[r] = r == 8 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 6; k >= 0; k--) {
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int r = k; r < 9; r++) {
            d1 += M[r][k] * n1[r];
            d2 += M[r][k] * n2[r];
        }
        d1 *= beta[k];
        d2 *= beta[k];
#pragma unroll
        for (int r = k; r < 9; r++) {
            n1[r] -= d1 * M[r][k];
            n2[r] -= d2 * M[r][k];
        }
    }
}

// The pencil of a sample: its basis (n1, n2) in the mode the handle is in (CVHIP_PENCIL_THIN_SVD, the reference's and the
// default; CVHIP_PENCIL_NULL_SPACE) and the real roots of det(a n1 + (1 - a) n2) = 0 (:324-358); returns the number of roots.
template <bool THIN>
__device__ int perspective_pencil(const uint4 (&sm)[7], double (&n1)[9], double (&n2)[9], double (&roots)[3])
{
    if (THIN) perspective_basis_thin_svd(sm, n1, n2);
    else perspective_basis_null_space(sm, n1, n2);
    // d[i][j][k] = det([F_i col 0, F_j col 1, F_k col 2]) (vgg_singF_from_FF, :326-337)
    double d[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                double m[9];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    m[r * 3 + 0] = i ? n2[r * 3 + 0] : n1[r * 3 + 0];
                    m[r * 3 + 1] = j ? n2[r * 3 + 1] : n1[r * 3 + 1];
                    m[r * 3 + 2] = k ? n2[r * 3 + 2] : n1[r * 3 + 2];
                }
                d[i][j][k] = det3(m);
            }
    const double c0 = -d[1][0][0] + d[0][1][1] + d[0][0][0] + d[1][1][0] + d[1][0][1] - d[0][1][0] - d[0][0][1] - d[1][1][1];
    const double c1 = d[0][0][1] - 2.0 * d[0][1][1] - 2.0 * d[1][0][1] + d[1][0][0] - 2.0 * d[1][1][0] + d[0][1][0] + 3.0 * d[1][1][1];
    const double c2 = d[1][1][0] + d[0][1][1] + d[1][0][1] - 3.0 * d[1][1][1];
    const double c3 = d[1][1][1];
    if (!(fabs(c0) > 1e-300) || !(fabs(c0) < __builtin_inf())) return 0;
    return cubic_real_roots(c0, c1, c2, c3, roots);
}

// One root of the pencil -> the normalised matrix validate_f receives, or false (:359-389 and validate_f :197-199):
// rank test, normalisation, sign consistency, finiteness.
__device__ bool perspective_root_matrix(const uint4 (&sm)[7], const double (&n1)[9], const double (&n2)[9], double a,
                                        double (&fo)[9])
{
    double f[9];
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = a * n1[i] + (1.0 - a) * n2[i]; // :359
    double sv[3];
    {
        const double ft[9] = {f[0], f[3], f[6], f[1], f[4], f[7], f[2], f[5], f[8]};
        lm::singular3(ft, sv); // f.transpose().svd, :361
    }
    bool good = !(sv[1] < 0.001) && !(sv[2] > 0.001); // :362-366
    // e1: null vector of F^T (last right singular vector of svd(F^T), :372-373) = normal of F's columns
    double e1[3], best = -1.0;
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
        const int ca = pr == 2 ? 1 : 0, cb = pr == 0 ? 1 : 2;
        const double ax = f[0 + ca], ay = f[3 + ca], az = f[6 + ca], bx = f[0 + cb], by = f[3 + cb], bz = f[6 + cb];
        const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const double nn = cx * cx + cy * cy + cz * cz;
        if (nn > best) {
            best = nn;
            e1[0] = cx;
            e1[1] = cy;
            e1[2] = cz;
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) fo[i] = f[i] / f[8]; // :369
    // sign consistency (:374-383): s = sum over the seven points of (F x2) .* (e1 x x1), per component
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double x1 = (double)sm[i].x, y1 = (double)sm[i].y, x2 = (double)sm[i].z, y2 = (double)sm[i].w;
        const double l0 = e1[1] * 1.0 - e1[2] * y1, l1 = e1[2] * x1 - e1[0] * 1.0, l2 = e1[0] * y1 - e1[1] * x1;
        const double g0 = fo[0] * x2 + fo[1] * y2 + fo[2], g1 = fo[3] * x2 + fo[4] * y2 + fo[5], g2 = fo[6] * x2 + fo[7] * y2 + fo[8];
        s0 += g0 * l0;
        s1 += g1 * l1;
        s2 += g2 * l2;
    }
    good = good && ((s0 > 0.0 && s1 > 0.0 && s2 > 0.0) || (s0 < 0.0 && s1 < 0.0 && s2 < 0.0));
#pragma unroll
    for (int i = 0; i < 9; i++) good = good && fabs(fo[i]) < __builtin_inf(); // validate_f, :197-199
    return good;
}

// The tail of validate_f once optimize_perspective_f's LM has returned parameters q (:412-423, :206-209): the hypothesis
// that goes on is f_from_perspective_params(q) (F[2][1] rebuilt from det = 0, F[2][2] = 1 exactly); it has to pass the
// rank test on THAT matrix, and all seven sample points must fit it.
__device__ bool perspective_root_accept(const double (&q)[7], const uint4 (&sm)[7], double t, double (&fo)[9])
{
    lm::matrix_of(q, fo);
    const double Mt[9] = {fo[0], fo[3], fo[6], fo[1], fo[4], fo[7], fo[2], fo[5], fo[8]};
    double s[3];
    lm::singular3(Mt, s);
    bool good = !(fabs(s[1]) < 1e-3 || fabs(s[2]) > 1e-3); // :418-423
#pragma unroll
    for (int i = 0; i < 7; i++) { // all sample points must fit, :206-209
        const double err = reprojection_error(fo, (double)sm[i].x, (double)sm[i].y, (double)sm[i].z, (double)sm[i].w);
        good = good && fabs(err) < __builtin_inf() && !(fabs(err) > t);
    }
    return good;
}

// Hypothesis generation in two kernels: one thread per SAMPLE draws it and solves for the pencil (QR, cubic), one
// thread per ROOT turns a root into a validated hypothesis - three times the parallelism for the expensive half, and
// neither kernel has to hold the other's registers (as one kernel this was 512 VGPRs + 1 KB of scratch per thread at
// one wave per SIMD: 1.4 ms per round).  Three hypothesis slots per sample; slots without a surviving root hold NaN.
// sample_idx != nullptr: the caller's samples (7 match indices each) instead of choose_inliers (test hook).
struct LateRoot { // a root whose loop outlasted the thread kernel's budget, for the call-level list of the batched rounds
    uint4 sm[7];
    double q[7];
};

struct PerspPencil {
    uint4 sm[7];
    double n1[9], n2[9], roots[3];
    int nr, pad;
};

template <bool THIN>
__global__ __launch_bounds__(64) void ransac_perspective_pencil_kernel(const uint4 *__restrict__ matches, uint32_t limit,
                                                                        unsigned long long seed, uint32_t round0, uint32_t per_round,
                                                                        uint32_t H, const uint32_t *__restrict__ sample_idx,
                                                                        PerspPencil *__restrict__ pencils)
{
    // H = per_round x (the rounds of this launch): sample g belongs to round round0 + g / per_round, and its random
    // numbers depend on the round and on its index within the round only
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= H) return;
    const uint32_t round = round0 + g / per_round, h = g - (g / per_round) * per_round;
    uint4 sm[7];
    int have = 0;
    if (sample_idx) {
#pragma unroll
        for (int i = 0; i < 7; i++) sm[i] = matches[sample_idx[(size_t)g * 7 + i]];
        have = 7;
    } else {
        unsigned long long state = mix64(seed ^ mix64(((unsigned long long)round << 32) | h));
        for (int tries = 0; tries < 512 && have < 7; tries++) { // choose_inliers, :155-175 (bounded here)
            state = mix64(state + 0x9E3779B97F4A7C15ull);
            const uint32_t idx = (uint32_t)(((state >> 32) * (unsigned long long)limit) >> 32);
            const uint4 nm = matches[idx];
            bool close = false;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const uint4 c = sm[i];
                auto dist = [](uint32_t a, uint32_t b) { return a > b ? a - b : b - a; };
                close = close || (i < have && (dist(nm.x, c.x) < 10u || dist(nm.y, c.y) < 10u || dist(nm.z, c.z) < 10u || dist(nm.w, c.w) < 10u));
            }
            if (!close) {
#pragma unroll
                for (int i = 0; i < 7; i++)
                    if (i == have) sm[i] = nm;
                have++;
            }
        }
    }
    PerspPencil &out = pencils[g];
    double n1[9], n2[9], roots[3] = {0.0, 0.0, 0.0};
    int nr = 0;
    if (have == 7) nr = perspective_pencil<THIN>(sm, n1, n2, roots);
    out.nr = nr;
    if (nr > 0) {
#pragma unroll
        for (int i = 0; i < 7; i++) out.sm[i] = sm[i];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            out.n1[i] = n1[i];
            out.n2[i] = n2[i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) out.roots[i] = roots[i];
    }
}

// validate_f's optimize_perspective_f over the sample itself (:201-205).  A 7-point solution has (nearly) zero
// reprojection error on its own sample, so for ~92 % of the roots least_squares returns at its first test (the gradient
// at the start) - those finish here.  The others (degenerate samples: gradients up to 1e11) run the full loop; a few
// such lanes per wave used to hold every wave of this kernel for the whole loop, so they are queued instead and the
// loop runs densely packed in ransac_perspective_lm_kernel.  queue[0] = count, queue[1 + i] = slot.
__global__ __launch_bounds__(64) void ransac_perspective_root_kernel(const PerspPencil *__restrict__ pencils, uint32_t H,
                                                                      double t, double *__restrict__ F,
                                                                      uint32_t *__restrict__ queue)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x; // slot = sample * 3 + root
    if (g >= 3u * H) return;
    const uint32_t h = g / 3u, k = g - 3u * h;
    const PerspPencil &pc = pencils[h];
    double f[9];
    bool ok = false, queued = false;
    if ((int)k < pc.nr) {
        uint4 sm[7];
        double n1[9], n2[9];
#pragma unroll
        for (int i = 0; i < 7; i++) sm[i] = pc.sm[i];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            n1[i] = pc.n1[i];
            n2[i] = pc.n2[i];
        }
        double fo[9];
        if (perspective_root_matrix(sm, n1, n2, pc.roots[k], fo)) {
            lm::Obs obs[7];
#pragma unroll
            for (int i = 0; i < 7; i++) obs[i] = lm::make_obs(sm[i].x, sm[i].y, sm[i].z, sm[i].w);
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = fo[i]; // params_from_perspective_f, :429-440
            if (lm::converged_at_start7(q, obs)) {
                ok = perspective_root_accept(q, sm, t, f);
            } else {
                queued = true;
#pragma unroll
                for (int i = 0; i < 7; i++) f[i] = q[i];
                f[7] = f[8] = 0.0;
            }
        }
    }
    if (queued) queue[1u + atomicAdd(&queue[0], 1u)] = g;
    const double nan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = (ok || queued) ? f[i] : nan;
}

// The queued roots: least_squares in full, then the same acceptance tail.  One WAVE per queued root (a persistent
// grid strides over the queue).  A scalar thread running lm::levenberg_marquardt<7> spends ~12 000 dependent cycles per
// iteration, and the kernel lasted as long as the round's longest loop (hundreds of iterations: ~1 ms, which made the
// generator the critical path of the rounds).  Here everything that loop computes element by element - the 7
// residuals, the 49 Jacobian entries, the 49 entries of J'J, the 7 entries of J'r - is computed by one lane per
// element, with lm's own functions (same operations, same order: the values are the scalar loop's), and made
// wave-uniform through LDS; the short serial parts (the 7x7 LU, norms, control flow) run redundantly in every lane on
// uniform values.  J'J is kept (in LDS) while steps are rejected (J does not change then).  ~1 500 cycles per iteration.
// late_list == nullptr: the roots of `queue` (slots of F, samples in `pencils`); otherwise the call's list of stragglers
// (LateRoot entries behind late_list + 64; late_list[0] = count, clamped to late_cap), results to F[entry].
__global__ __launch_bounds__(64) void ransac_perspective_lm_kernel(const PerspPencil *__restrict__ pencils, double t,
                                                                    double *__restrict__ F, const uint32_t *__restrict__ queue,
                                                                    const uint32_t *__restrict__ late_list, uint32_t late_cap)
{
    __shared__ double sJ[49], sA[49], sV[8];
    const uint32_t lane = threadIdx.x, n_queued = late_list ? min(late_list[0], late_cap) : queue[0];
    const int k = lane < 49u ? (int)(lane / 7u) : 6, e = lane < 49u ? (int)(lane % 7u) : 6; // this lane's observation / parameter
    const LateRoot *const roots = reinterpret_cast<const LateRoot *>(late_list + 64);
    for (uint32_t item = blockIdx.x; item < n_queued; item += gridDim.x) {
        const uint32_t g = late_list ? item : queue[1u + item];
        const uint4 *const smp = late_list ? roots[item].sm : pencils[g / 3u].sm;
        uint4 sm[7];
#pragma unroll
        for (int kk = 0; kk < 7; kk++) sm[kk] = smp[kk];
        const uint4 mine = smp[k];
        const lm::Obs ob = lm::make_obs(mine.x, mine.y, mine.z, mine.w);
        double q[7], r[7], gv[7], M[9];
#pragma unroll
        for (int j = 0; j < 7; j++) q[j] = late_list ? roots[item].q[j] : F[(size_t)g * 9 + j];
        __syncthreads(); // (the previous item's LDS reads are done)

        // r = residuals at `at` (lane (k, 0) computes r_k), uniform in every lane afterwards
        const auto evaluate = [&](const double (&at)[7], double (&into)[7]) {
            lm::matrix_of(at, M);
            const double rk = lm::residual_of(M, ob);
            __syncthreads();
            if (lane < 49u && e == 0) sV[k] = rk;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 7; j++) into[j] = sV[j];
        };
        // J at `at` -> sJ (lane (k, e) computes J[k][e]); g = J'res (lane j computes g_j: long_dot's order for n = 7,
        // 0 + J_0j res_0 + J_1j res_1 + ...), uniform; AJ = J'J (lane (i, j): the same serial order), uniform
        const auto linearise = [&](const double (&at)[7], const double (&res)[7]) {
            lm::matrix_of(at, M);
            const double c = lm::gradient_terms(M, ob);
            const double jke = lm::gradient_element(M, ob, c, e);
            __syncthreads();
            if (lane < 49u) sJ[k * 7 + e] = jke;
            __syncthreads();
            double gj = 0.0, a = 0.0;
            if (lane < 49u) {
#pragma unroll
                for (int kk = 0; kk < 7; kk++) a += sJ[kk * 7 + k] * sJ[kk * 7 + e]; // (k, e) = (row, column) of J'J here
            }
            if (lane < 7u) {
#pragma unroll
                for (int kk = 0; kk < 7; kk++) gj += sJ[kk * 7 + (int)lane] * res[kk];
            }
            if (lane < 49u) sA[lane] = a;
            if (lane < 7u) sV[lane] = gj;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 7; j++) gv[j] = sV[j]; // (J'J stays in sA until the next linearise)
        };
        const auto largest = [](const double (&v)[7]) {
            double m = v[0];
#pragma unroll
            for (int j = 1; j < 7; j++)
                if (m < v[j]) m = v[j];
            return m;
        };
        const auto norm7 = [](const double (&v)[7]) { return sqrt(lm::long_dot(v, 1, v, 1, 7)); };

        // ---- least_squares (:515-621), statement for statement as lm::levenberg_marquardt<7>
        int status = 0; // 0 = running, 1 = Ok, 2 = Err
        evaluate(q, r);
        linearise(q, r);
        if (fabs(largest(gv)) <= 1e-12) status = 1;
        double mu = 0.0, nu = 2.0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const double djj = sA[j * 7 + j];
            if (j == 0 || djj >= mu) mu = djj;
        }
        mu *= 1e-3;
        for (int iteration = 0; status == 0 && iteration < 1000; iteration++) {
            double A[49], step[7];
#pragma unroll
            for (int m = 0; m < 49; m++) A[m] = sA[m];
#pragma unroll
            for (int i = 0; i < 7; i++) A[i * 7 + i] += mu;
#pragma unroll
            for (int j = 0; j < 7; j++) step[j] = gv[j];
            if (!lm::solve7(A, step)) {
                status = 2;
                break;
            }
            if (norm7(step) <= 1e-12 * (norm7(q) + 1e-12)) {
                status = 1;
                break;
            }
            double trial[7], damped[7], r_new[7];
#pragma unroll
            for (int j = 0; j < 7; j++) trial[j] = q[j] + step[j];
            evaluate(trial, r_new);
            const double before = lm::long_dot(r, 1, r, 1, 7);
            const double after = lm::long_dot(r_new, 1, r_new, 1, 7);
#pragma unroll
            for (int j = 0; j < 7; j++) damped[j] = step[j] * mu + gv[j];
            const double rho = (before - after) / lm::long_dot(step, 1, damped, 1, 7);
            if (rho > 0.0) {
                const bool converged = sqrt(before) - sqrt(after) < 0.0 * sqrt(before);
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    r[j] = r_new[j];
                    q[j] = trial[j];
                }
                linearise(q, r);
                if (converged || fabs(largest(gv)) <= 1e-12) {
                    status = 1;
                    break;
                }
                const double w = 2.0 * rho - 1.0, shrink = 1.0 - w * w * w;
                mu *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
                nu = 2.0;
            } else {
                mu *= nu;
                nu *= 2.0;
            }
            if (sqrt(lm::long_dot(r, 1, r, 1, 7)) <= 1e-12) status = 1;
        }
        double f[9];
        const bool ok = status == 1 && perspective_root_accept(q, sm, t, f);
        const double nan = __builtin_nan("");
        if (lane < 9u) F[(size_t)g * 9 + lane] = ok ? f[lane] : nan;
    }
}

// The same loop, one THREAD per queued root: the form for the reference's own pencil (CVHIP_PENCIL_THIN_SVD), where
// EVERY root that passes the rank and sign tests enters least_squares away from a solution - ~80 000 roots per round
// instead of ~6 000 - and the loops are short: the Jacobian as the reference writes it (:473-512; c = d = a plain sum of
// the four linear forms) is not the derivative of the residual, steps are rejected until mu has grown enough for the
// step to vanish (mu x 2, x 4, x 8, ...: `delta.norm() <= 1e-12 (params.norm() + 1e-12)` after ~5-25 iterations, "found").
// With that many independent short loops the chip is filled by roots, not by the inside of one loop: lm::levenberg_marquardt7
// (register-resident, the scalar loop itself) on 64 roots per wave.  A root that has not finished within `budget`
// iterations is handed to the wave-per-root kernel through `late` (which runs it from its start: same values), so one
// long loop does not hold 63 finished lanes.
__global__ __launch_bounds__(64) void ransac_perspective_lm_thread_kernel(const PerspPencil *__restrict__ pencils, double t,
                                                                           double *__restrict__ F, const uint32_t *__restrict__ queue,
                                                                           uint32_t *__restrict__ late, int budget)
{
    const uint32_t item = blockIdx.x * 64 + threadIdx.x;
    if (item >= queue[0]) return;
    const uint32_t g = queue[1u + item];
    const PerspPencil &pc = pencils[g / 3u];
    uint4 sm[7];
    lm::Obs obs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sm[i] = pc.sm[i];
        obs[i] = lm::make_obs(sm[i].x, sm[i].y, sm[i].z, sm[i].w);
    }
    double q[7];
#pragma unroll
    for (int j = 0; j < 7; j++) q[j] = F[(size_t)g * 9 + j];
    const int status = lm::levenberg_marquardt7(q, obs, budget); // 1 = Ok, 0 = Err, 2 = budget exhausted
    if (status == 2) {
        late[1u + atomicAdd(&late[0], 1u)] = g; // (F[g] still holds the start parameters)
        return;
    }
    double f[9];
    const bool ok = status == 1 && perspective_root_accept(q, sm, t, f);
    const double nan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = ok ? f[i] : nan;
}

// The thin-SVD pencil's roots fall into two classes (census on config 5's matches, scripts/lm_census.py; oracle: the same):
// ~2/3 never have a step accepted - 0-7 trips until the step vanishes - and ~1/3 accept step after step (8-25 trips, mean
// 14).  One launch over all of them runs every wave for its slowest lane: ~25 trips for a mean of ~6.  So the classes are
// separated: ransac_lm_start_kernel runs the scalar loop for EVERY queued root but leaves it at the first accepted step
// (the first class finishes here, in uniform short loops; the others go onto a second queue), ransac_lm_run_kernel runs
// the second class, densely packed, from its start.  Same function, same values (lm::levenberg_marquardt<7>).
// (Tried and dropped: the loop cut into a linearise and an iterate piece with the root's state - q, r, J'r, J'J - in
// memory between them and the sample re-read per observation, at two waves per SIMD instead of one: bit-identical, but
// every trip then waited for ~10 dependent loads: 19 ns per root against 8.)
__global__ __launch_bounds__(64) void ransac_lm_start_kernel(const PerspPencil *__restrict__ pencils, double t, double *__restrict__ F,
                                                              const uint32_t *__restrict__ queue, uint32_t *__restrict__ next)
{
    const uint32_t item = blockIdx.x * 64 + threadIdx.x;
    if (item >= queue[0]) return;
    const uint32_t g = queue[1u + item];
    const PerspPencil &pc = pencils[g / 3u];
    uint4 sm[7];
    lm::Obs obs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sm[i] = pc.sm[i];
        obs[i] = lm::make_obs(sm[i].x, sm[i].y, sm[i].z, sm[i].w);
    }
    double q[7];
#pragma unroll
    for (int j = 0; j < 7; j++) q[j] = F[(size_t)g * 9 + j];
    const int status = lm::levenberg_marquardt7_lean(q, obs, 1000, true);
    if (status == lm::LM_EVENT_ACCEPTED) {
        next[1u + atomicAdd(&next[0], 1u)] = g; // (F[g] still holds the start parameters)
        return;
    }
    double f[9];
    const bool ok = status == 1 && perspective_root_accept(q, sm, t, f);
    const double nan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = ok ? f[i] : nan;
}

// late_list != nullptr: a straggler is copied (sample, start parameters) onto the CALL's list - late_list[0] = count,
// entries behind 256 bytes, late_cap of them at most (beyond: counted, not stored; the host fails the call) - and its slot
// holds NaN; otherwise its slot goes onto the batch's `late` queue.
__global__ __launch_bounds__(64) void ransac_lm_run_kernel(const PerspPencil *__restrict__ pencils, double t, double *__restrict__ F,
                                                            const uint32_t *__restrict__ queue, uint32_t *__restrict__ late, int budget,
                                                            uint32_t *__restrict__ late_list, uint32_t late_cap)
{
    const uint32_t item = blockIdx.x * 64 + threadIdx.x;
    if (item >= queue[0]) return;
    const uint32_t g = queue[1u + item];
    const PerspPencil &pc = pencils[g / 3u];
    uint4 sm[7];
    lm::Obs obs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sm[i] = pc.sm[i];
        obs[i] = lm::make_obs(sm[i].x, sm[i].y, sm[i].z, sm[i].w);
    }
    double q[7];
#pragma unroll
    for (int j = 0; j < 7; j++) q[j] = F[(size_t)g * 9 + j];
    const int status = lm::levenberg_marquardt7_lean(q, obs, budget, false);
    const double nan = __builtin_nan("");
    if (status == lm::LM_EVENT_BUDGET) {
        if (!late_list) {
            late[1u + atomicAdd(&late[0], 1u)] = g; // (F[g] still holds the start parameters)
            return;
        }
        const uint32_t at = atomicAdd(&late_list[0], 1u);
        if (at < late_cap) {
            LateRoot &lr = reinterpret_cast<LateRoot *>(late_list + 64)[at];
#pragma unroll
            for (int i = 0; i < 7; i++) {
                lr.sm[i] = sm[i];
                lr.q[i] = F[(size_t)g * 9 + i];
            }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = nan;
        return;
    }
    double f[9];
    const bool ok = status == 1 && perspective_root_accept(q, sm, t, f);
#pragma unroll
    for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = ok ? f[i] : nan;
}

// The same two passes with the lanes REFILLED (round 5; cvhip_ransac_set_lm_pipeline(dev, 2), the default).  Above, a wave runs
// for its slowest root - 8-25 trips in the second pass for a mean of 14, 0-7 in the first - on one wave per SIMD (the scalar
// loop holds a whole register file), and a launch is a grid of waves most of which find the queue empty: the counters showed
// 0.6-0.85 waves per SIMD while these kernels run, each issuing well (6-8 cycles per vector instruction) - what was missing
// was work in the lanes, not issue slots.  Here the grid is persistent (a wave per SIMD) and a lane whose root has left the
// loop takes the next one off the queue: roots are fetched for all idle lanes of a wave at once, when enough of them are
// idle (the start of a root - residuals, Jacobian, J'J at the start parameters - is a code path of its own, paid per fetch,
// not per root), and so is the acceptance tail of the finished ones.  A root's arithmetic is lm::levenberg_marquardt7_lean's,
// statement for statement - only WHEN a lane executes it has changed - so the values are the scalar loop's, bit for bit
// (test_device_reference_pencil_matches_thin_svd_rows_per_sample runs all three forms against each other).
// STOP = true: the first pass (every queued root; a root leaves at its first accepted step, onto `out_queue`);
// STOP = false: the second pass (the roots of `queue` from their start; a root still running after `budget` trips goes
// to the call's straggler list, or onto `out_queue` when there is none).  cursor: one zeroed word per launch.
template <bool STOP>
__global__ __launch_bounds__(64) void ransac_lm_refill_kernel(const PerspPencil *__restrict__ pencils, double t, double *__restrict__ F,
                                                               const uint32_t *__restrict__ queue, uint32_t *__restrict__ out_queue, int budget,
                                                               uint32_t *__restrict__ late_list, uint32_t late_cap, uint32_t *__restrict__ cursor)
{
    constexpr uint32_t REFILL_AT = STOP ? 24u : 12u; // idle lanes that trigger a fetch (all of them once the queue is empty)
    const uint32_t n_items = queue[0];
    const uint32_t lane = threadIdx.x;
    const double nan = __builtin_nan("");
    // lane state: 0 = idle, 1 = in the loop, 2 = left the loop (status in `left`), waiting for the next flush
    int state = 0, left = 0, iteration = 0;
    uint32_t g = 0;
    uint4 sm[7]; // (the sample as integers: 28 registers across the trips; an observation's doubles are made where they are used)
    // (J'J between the trips lives in LDS, a column per lane: written once per linearisation, read once per trip - 56
    // registers less across the loop, where the compiler otherwise went to scratch)
    __shared__ double s_jj[28][64];
    double q[7], r[7], gv[7], M[9], mu = 0.0, nu = 2.0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sm[i] = make_uint4(0u, 0u, 0u, 0u);
        q[i] = r[i] = gv[i] = 0.0;
    }
    const auto evaluate = [&](const double (&at)[7], double (&into)[7]) {
        lm::matrix_of(at, M);
#pragma unroll
        for (int i = 0; i < 7; i++) into[i] = lm::residual_of(M, lm::make_obs(sm[i].x, sm[i].y, sm[i].z, sm[i].w));
    };
    const auto linearise = [&](const double (&at)[7], const double (&res)[7]) { // J'r and J'J at `at` (levenberg_marquardt7_lean's)
        lm::matrix_of(at, M);
        double jj[28];
#pragma unroll
        for (int j = 0; j < 7; j++) gv[j] = 0.0;
#pragma unroll
        for (int m = 0; m < 28; m++) jj[m] = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            double row[7];
            lm::gradient_of(M, lm::make_obs(sm[k].x, sm[k].y, sm[k].z, sm[k].w), row);
            int m = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                gv[i] += row[i] * res[k];
#pragma unroll
                for (int j = i; j < 7; j++) jj[m++] += row[i] * row[j];
            }
        }
#pragma unroll
        for (int m = 0; m < 28; m++) s_jj[m][lane] = jj[m];
    };
    const auto largest = [](const double (&v)[7]) {
        double m = v[0];
#pragma unroll
        for (int j = 1; j < 7; j++)
            if (m < v[j]) m = v[j];
        return m;
    };
    const auto norm7 = [](const double (&v)[7]) { return sqrt(lm::long_dot(v, 1, v, 1, 7)); };

    bool exhausted = false; // (wave-uniform) the queue has been handed out
    for (;;) {
        const unsigned long long busy = __builtin_amdgcn_ballot_w64(state == 1);
        const uint32_t n_free = 64u - (uint32_t)__builtin_popcountll(busy);
        const bool refill = !exhausted && (n_free >= REFILL_AT || busy == 0ull);
        const bool drain = exhausted && busy == 0ull; // the queue is handed out and every root has left the loop: the last flush
        if (refill || drain) {
            // ---- flush: the acceptance tail of the roots that have left the loop (validate_f after the LM, :206-209, 412-423)
            if (state == 2) {
                if (left == lm::LM_EVENT_ACCEPTED) {
                    out_queue[1u + atomicAdd(&out_queue[0], 1u)] = g; // (F[g] still holds the start parameters)
                } else if (left == lm::LM_EVENT_BUDGET) {
                    if (!late_list) {
                        out_queue[1u + atomicAdd(&out_queue[0], 1u)] = g;
                    } else {
                        const uint32_t at = atomicAdd(&late_list[0], 1u);
                        if (at < late_cap) {
                            LateRoot &lr = reinterpret_cast<LateRoot *>(late_list + 64)[at];
#pragma unroll
                            for (int i = 0; i < 7; i++) {
                                lr.sm[i] = sm[i];
                                lr.q[i] = F[(size_t)g * 9 + i];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = nan;
                    }
                } else {
                    double f[9];
                    const bool ok = left == 1 && perspective_root_accept(q, sm, t, f);
#pragma unroll
                    for (int i = 0; i < 9; i++) F[(size_t)g * 9 + i] = ok ? f[i] : nan;
                }
                state = 0;
            }
            // ---- fetch: the next roots of the queue for the idle lanes, and the start of least_squares for them (:542-558)
            if (!exhausted) {
                const unsigned long long idle = __builtin_amdgcn_ballot_w64(state == 0);
                const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                uint32_t base = 0u;
                if (lane == 0u) base = atomicAdd(cursor, n_idle);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t item = base + rank;
                exhausted = base + n_idle >= n_items;
                if (state == 0 && item < n_items) {
                    g = queue[1u + item];
                    const PerspPencil &pc = pencils[g / 3u];
#pragma unroll
                    for (int i = 0; i < 7; i++) sm[i] = pc.sm[i];
#pragma unroll
                    for (int j = 0; j < 7; j++) q[j] = F[(size_t)g * 9 + j];
                    evaluate(q, r);
                    linearise(q, r);
                    state = 1;
                    iteration = 0;
                    if (fabs(largest(gv)) <= 1e-12) {
                        state = 2;
                        left = 1;
                    }
                    mu = 0.0;
                    {
                        int m = 0;
#pragma unroll
                        for (int j = 0; j < 7; j++) {
                            const double djj = s_jj[m][lane];
                            if (j == 0 || djj >= mu) mu = djj;
                            m += 7 - j;
                        }
                    }
                    mu *= 1e-3;
                    nu = 2.0;
                }
            }
            if (drain) break;
            continue;
        }
        // ---- one trip of the loop (:560-617) for the lanes that are in it
        if (state == 1) {
            if (iteration >= budget) { // (the reference's cap is 1000)
                state = 2;
                left = lm::LM_EVENT_BUDGET;
            } else {
                double A[49], step[7];
                {
                    int m = 0;
#pragma unroll
                    for (int i = 0; i < 7; i++)
#pragma unroll
                        for (int j = i; j < 7; j++) {
                            const double v = s_jj[m++][lane];
                            A[i * 7 + j] = v;
                            A[j * 7 + i] = v;
                        }
                }
#pragma unroll
                for (int i = 0; i < 7; i++) A[i * 7 + i] += mu;
#pragma unroll
                for (int j = 0; j < 7; j++) step[j] = gv[j];
                if (!lm::solve7(A, step)) {
                    state = 2;
                    left = 0;
                } else if (norm7(step) <= 1e-12 * (norm7(q) + 1e-12)) {
                    state = 2;
                    left = 1;
                } else {
                    double trial[7], damped[7], r_new[7];
#pragma unroll
                    for (int j = 0; j < 7; j++) trial[j] = q[j] + step[j];
                    evaluate(trial, r_new);
                    const double before = lm::long_dot(r, 1, r, 1, 7);
                    const double after = lm::long_dot(r_new, 1, r_new, 1, 7);
#pragma unroll
                    for (int j = 0; j < 7; j++) damped[j] = step[j] * mu + gv[j];
                    const double rho = (before - after) / lm::long_dot(step, 1, damped, 1, 7);
                    if (rho > 0.0) {
                        if (STOP) {
                            state = 2;
                            left = lm::LM_EVENT_ACCEPTED;
                        } else {
                            const bool converged = sqrt(before) - sqrt(after) < 0.0 * sqrt(before);
#pragma unroll
                            for (int j = 0; j < 7; j++) {
                                r[j] = r_new[j];
                                q[j] = trial[j];
                            }
                            linearise(q, r);
                            if (converged || fabs(largest(gv)) <= 1e-12) {
                                state = 2;
                                left = 1;
                            }
                            const double w = 2.0 * rho - 1.0, shrink = 1.0 - w * w * w;
                            mu *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
                            nu = 2.0;
                        }
                    } else {
                        mu *= nu;
                        nu *= 2.0;
                    }
                    if (state == 1 && sqrt(lm::long_dot(r, 1, r, 1, 7)) <= 1e-12) {
                        state = 2;
                        left = 1;
                    }
                    iteration++;
                    if (state == 1 && iteration >= 1000) { // "Levenberg-Marquardt failed to converge"
                        state = 2;
                        left = 0;
                    }
                }
            }
        }
    }
}

// queues: LM_QUEUES lists of 1 + 3 H words, `stride` words apart (the root kernel's queue, the roots whose first step was
// accepted, the stragglers)
// H samples: `per_round` each of the rounds round0, round0 + 1, ... (H a multiple of per_round)
constexpr uint32_t LM_REFILL_WAVES = 1024; // persistent waves of the refilled LM kernels: one per SIMD (256 CUs x 4)
constexpr int LM_THREAD_BUDGET = 48; // iterations a root gets in a thread-per-root kernel (thin-SVD pencil: the oracle's longest of 3 000 was 30)
constexpr int LM_QUEUES = 4; // (the last list is not a list: its first words are the refilled kernels' queue cursors)
static void launch_generate_perspective(int pencil, int lm_pipeline, const uint4 *m4, uint32_t limit, double t, unsigned long long seed, uint32_t round0,
                                        uint32_t per_round, uint32_t H, const uint32_t *sample_idx, PerspPencil *pencils,
                                        uint32_t *queues, size_t stride, uint32_t *late_list, uint32_t late_cap, double *d_F, hipStream_t s)
{
    const bool thin = pencil == CVHIP_PENCIL_THIN_SVD;
    const dim3 roots_grid((3 * H + 63) / 64);
    const PerspPencil *pc = pencils;
    uint32_t *queue = queues;
    (void)hipMemsetAsync(queue, 0, sizeof(uint32_t), s);
    if (thin) {
        for (int k = 1; k < LM_QUEUES; k++) (void)hipMemsetAsync(queues + k * stride, 0, sizeof(uint32_t), s);
        hipLaunchKernelGGL(ransac_perspective_pencil_kernel<true>, dim3((H + 63) / 64), dim3(64), 0, s, m4, limit, seed, round0, per_round, H,
                           sample_idx, pencils);
    } else {
        hipLaunchKernelGGL(ransac_perspective_pencil_kernel<false>, dim3((H + 63) / 64), dim3(64), 0, s, m4, limit, seed, round0, per_round, H,
                           sample_idx, pencils);
    }
    hipLaunchKernelGGL(ransac_perspective_root_kernel, roots_grid, dim3(64), 0, s, pc, H, t, d_F, queue);
    if (thin) {
        // every queued root runs least_squares: the start and the rejected steps after it for all of them, the roots that
        // have a step accepted densely packed after that, the stragglers on a wave of their own
        uint32_t *late = queues + 2 * stride;
        if (lm_pipeline == 2) {
            // the two passes on refilled lanes, a persistent wave per SIMD (the loop's registers leave room for no more)
            uint32_t *cursors = queues + 3 * stride;
            (void)hipMemsetAsync(cursors, 0, 128, s);
            const uint32_t waves = std::min<uint32_t>((3 * H + 63) / 64, LM_REFILL_WAVES);
            hipLaunchKernelGGL(ransac_lm_refill_kernel<true>, dim3(waves), dim3(64), 0, s, pc, t, d_F, (const uint32_t *)queue, queue + stride, 1000,
                               (uint32_t *)nullptr, 0u, cursors);
            hipLaunchKernelGGL(ransac_lm_refill_kernel<false>, dim3(waves), dim3(64), 0, s, pc, t, d_F, (const uint32_t *)(queue + stride), late,
                               LM_THREAD_BUDGET, late_list, late_cap, cursors + 16);
            if (late_list) return; // (the call's stragglers run once, behind its last batch: ransac_rounds)
        } else if (lm_pipeline) {
            hipLaunchKernelGGL(ransac_lm_start_kernel, roots_grid, dim3(64), 0, s, pc, t, d_F, (const uint32_t *)queue, queue + stride);
            hipLaunchKernelGGL(ransac_lm_run_kernel, roots_grid, dim3(64), 0, s, pc, t, d_F, (const uint32_t *)(queue + stride), late, LM_THREAD_BUDGET,
                               late_list, late_cap);
            if (late_list) return; // (the call's stragglers run once, behind its last batch: ransac_rounds)
        } else {
            hipLaunchKernelGGL(ransac_perspective_lm_thread_kernel, roots_grid, dim3(64), 0, s, pc, t, d_F, (const uint32_t *)queue, late, LM_THREAD_BUDGET);
        }
        queue = late;
    }
    // a persistent grid of waves strides over the queue (one wave per queued root)
    hipLaunchKernelGGL(ransac_perspective_lm_kernel, dim3(std::min<uint32_t>(3 * H, 4096u)), dim3(64), 0, s, pc, t, d_F, (const uint32_t *)queue,
                       (const uint32_t *)nullptr, 0u);
}

// Ord for RansacIterationResult (fundamentalmatrix.rs:623-649)
__device__ __forceinline__ bool ransac_better(uint32_t ca, double ea, uint32_t cb, double eb)
{
    if (ca != cb) return ca > cb;
    const bool af = fabs(ea) < __builtin_inf(), bf = fabs(eb) < __builtin_inf();
    if (af != bf) return af;
    if (!af) return false;
    return ea < eb;
}

// `tied` (optional): the round's maximum list of ransac_round_max_kernel - tells whether this round's error sums were
// computed (ransac_round_needs_errors); without it they are taken as given (cvhip_ransac_score-style full sums).
__global__ __launch_bounds__(1024) void ransac_pick_best_kernel(const double *__restrict__ F,
                                                                 const uint32_t *__restrict__ counts,
                                                                 const double *__restrict__ err_sums, uint32_t H,
                                                                 uint32_t min_count, const uint32_t *__restrict__ tied,
                                                                 RansacBest *best)
{
    // (read before thread 0 updates *best below; every thread evaluates the same values)
    const uint32_t errors_known = tied == nullptr || (tied[0] != 0u && ransac_round_needs_errors(tied, best)) ? 1u : 0u;
    __syncthreads();
    __shared__ uint32_t s_cnt[1024];
    __shared__ double s_err[1024];
    __shared__ uint32_t s_idx[1024];
    uint32_t bc = 0, bi = 0xFFFFFFFFu;
    double be = __builtin_inf();
    // Equal (count, error): the smaller slot, whatever order the candidates are visited in (the full scan visits them
    // in slot order, the round's maximum list is in the order its atomics landed).
    const auto better_slot = [](uint32_t ca, double ea, uint32_t ia, uint32_t cb, double eb, uint32_t ib) {
        return ransac_better(ca, ea, cb, eb) || (!ransac_better(cb, eb, ca, ea) && ia < ib);
    };
    const bool listed = tied != nullptr && tied[0] <= TIED_CAP;
    if (listed) {
        // the round's best hypothesis is one with the round's largest count: only those need looking at (a handful)
        for (uint32_t k = threadIdx.x; k < tied[0]; k += 1024) {
            const uint32_t h = tied[1 + k], c = counts[h];
            if (c < min_count) continue; // :218-220
            const double e = err_sums[h] / (double)c;
            if (bi == 0xFFFFFFFFu || better_slot(c, e, h, bc, be, bi)) {
                bc = c;
                be = e;
                bi = h;
            }
        }
    } else {
        for (uint32_t h = threadIdx.x; h < H; h += 1024) {
            const uint32_t c = counts[h];
            if (c < min_count) continue; // :218-220
            const double e = err_sums[h] / (double)c;
            if (bi == 0xFFFFFFFFu || ransac_better(c, e, bc, be)) {
                bc = c;
                be = e;
                bi = h;
            }
        }
    }
    s_cnt[threadIdx.x] = bc;
    s_err[threadIdx.x] = be;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (uint32_t s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const uint32_t o = threadIdx.x + s;
            if (s_idx[o] != 0xFFFFFFFFu &&
                (s_idx[threadIdx.x] == 0xFFFFFFFFu ||
                 better_slot(s_cnt[o], s_err[o], s_idx[o], s_cnt[threadIdx.x], s_err[threadIdx.x], s_idx[threadIdx.x]))) {
                s_cnt[threadIdx.x] = s_cnt[o];
                s_err[threadIdx.x] = s_err[o];
                s_idx[threadIdx.x] = s_idx[o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_idx[0] != 0xFFFFFFFFu) {
        if (!best->valid || ransac_better(s_cnt[0], s_err[0], best->matches_count, best->best_error)) {
            for (int i = 0; i < 9; i++) best->f[i] = F[(size_t)s_idx[0] * 9 + i];
            best->matches_count = s_cnt[0];
            best->best_error = s_err[0];
            best->valid = 1;
            best->err_known = errors_known;
        }
    }
}

__global__ void ransac_inlier_mask_kernel(const RansacBest *__restrict__ best, const uint4 *__restrict__ matches,
                                          uint32_t N, double t, uint8_t *__restrict__ mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = best->f[k];
    const uint4 m = matches[i];
    const double err = reprojection_error(f, (double)m.x, (double)m.y, (double)m.z, (double)m.w);
    mask[i] = (fabs(err) < __builtin_inf() && !(fabs(err) > t)) ? 1 : 0; // optimize_result, :231-239
}

// ---- optimize_result's tail (:233-254) without leaving the device: the winner's inliers in list order, the refit on
// them (ransac_refit_kernel), the inliers of the refitted matrix.  One workgroup of 1024 threads compacts (the order is
// the refit's summation order, so it is the list's).
__global__ __launch_bounds__(1024) void ransac_compact_inliers_kernel(const uint8_t *__restrict__ mask, const uint4 *__restrict__ matches,
                                                                       uint32_t N, uint4 *__restrict__ inliers, uint32_t *__restrict__ n_out)
{
    __shared__ uint32_t wave_total[16];
    const uint32_t chunk = (N + 1023u) / 1024u, i0 = min(threadIdx.x * chunk, N), i1 = min(i0 + chunk, N);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; i++) mine += mask[i] ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, sft, 64);
        if ((threadIdx.x & 63) >= (uint32_t)sft) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_total[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t at = incl - mine, total = 0;
    for (uint32_t w = 0; w < 16; w++) {
        if (w < (threadIdx.x >> 6)) at += wave_total[w];
        total += wave_total[w];
    }
    for (uint32_t i = i0; i < i1; i++)
        if (mask[i]) inliers[at++] = matches[i];
    if (threadIdx.x == 0) *n_out = total;
}
// fits_model of the refitted matrix as optimize_result's second filter evaluates it (:248-254; the arithmetic of the
// host loop it replaces: lm::residual_of)
__global__ void ransac_refit_mask_kernel(const double *__restrict__ F, const uint4 *__restrict__ matches, uint32_t N, double t,
                                         uint8_t *__restrict__ mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double M[9];
#pragma unroll
    for (int k = 0; k < 9; k++) M[k] = F[k];
    const uint4 m = matches[i];
    const double err = lm::residual_of(M, lm::make_obs(m.x, m.y, m.z, m.w));
    mask[i] = (fabs(err) < __builtin_inf() && !(fabs(err) > t)) ? 1 : 0;
}

} // namespace cvhip

using namespace cvhip;

namespace {
bool dev_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
} // namespace

extern "C" int cvhip_ransac_score(cvhip_device *dev, const double *F, uint32_t H, const uint32_t *matches,
                                  uint32_t N, double t, uint32_t *out_count, double *out_err_sum)
{
    if (!dev || (!F && H) || (!matches && N) || !out_count || !out_err_sum)
        return fail(CVHIP_ERR_INVALID, "null argument");
    if (H == 0) return CVHIP_OK;
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    // the kernel reads matches as 16-byte vectors: a device pointer that is not 16-byte aligned is copied too
    const bool f_dev = dev_ptr(F),
               m_dev = N ? (dev_ptr(matches) && (reinterpret_cast<uintptr_t>(matches) & 15u) == 0) : true,
               c_dev = dev_ptr(out_count),
               e_dev = dev_ptr(out_err_sum);
    double *d_F = const_cast<double *>(F), *d_err = out_err_sum;
    uint32_t *d_m = const_cast<uint32_t *>(matches), *d_cnt = out_count;
    hipError_t e = hipSuccess;
    if (!f_dev) {
        e = hipMalloc(&d_F, (size_t)H * 9 * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(d_F, F, (size_t)H * 9 * sizeof(double), hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && !m_dev) {
        e = hipMalloc(&d_m, (size_t)N * 4 * sizeof(uint32_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_m, matches, (size_t)N * 4 * sizeof(uint32_t),
                               dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && !c_dev) e = hipMalloc(&d_cnt, (size_t)H * sizeof(uint32_t));
    if (e == hipSuccess && !e_dev) e = hipMalloc(&d_err, (size_t)H * sizeof(double));
    if (e == hipSuccess) {
        launch_ransac_score(d_F, H, d_m, N, t, d_cnt, d_err, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !c_dev)
        e = hipMemcpyAsync(out_count, d_cnt, (size_t)H * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && !e_dev)
        e = hipMemcpyAsync(out_err_sum, d_err, (size_t)H * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && !(c_dev && e_dev && f_dev && m_dev)) e = hipStreamSynchronize(s);
    if (!f_dev && d_F != F) (void)hipFree(d_F);
    if (!m_dev && d_m != matches) (void)hipFree(d_m);
    if (!c_dev && d_cnt != out_count) (void)hipFree(d_cnt);
    if (!e_dev && d_err != out_err_sum) (void)hipFree(d_err);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("ransac_score: ") + hipGetErrorString(e));
    return CVHIP_OK;
}

// Test hook: what ONE ROUND's scoring computes for caller-given hypotheses - the inlier counts of the counting kernel
// (packed-f32 screen, f64 in the guard band) and the ordered error sums of the hypotheses tied at the maximum count
// (0 for every other one).  No pruning (no best of earlier rounds, min_count 0): counts must equal cvhip_ransac_score's.
extern "C" int cvhip_ransac_round_score(cvhip_device *dev, const double *F, uint32_t H, const uint32_t *matches, uint32_t N,
                                        double t, uint32_t *out_count, double *out_err_sum)
{
    if (!dev || (!F && H) || (!matches && N) || !out_count || !out_err_sum) return fail(CVHIP_ERR_INVALID, "null argument");
    if (H == 0) return CVHIP_OK;
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    double *d_F = nullptr, *d_err = nullptr;
    uint32_t *d_m = nullptr, *d_cnt = nullptr, *d_live = nullptr;
    RansacBest *d_best = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_F, (size_t)H * 9));
    CVHIP_TRY_HIP(mem.alloc(&d_err, std::max<size_t>(H, 64)));
    CVHIP_TRY_HIP(mem.alloc(&d_m, (size_t)std::max(N, 1u) * 4));
    CVHIP_TRY_HIP(mem.alloc(&d_cnt, H));
    CVHIP_TRY_HIP(mem.alloc(&d_live, (size_t)H + 4 + TIED_CAP));
    CVHIP_TRY_HIP(mem.alloc(&d_best, 1));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_F, F, (size_t)H * 9 * sizeof(double), dev_ptr(F) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (N) CVHIP_TRY_HIP(hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    CVHIP_TRY_HIP(hipMemsetAsync(d_best, 0, sizeof(RansacBest), s));
    CVHIP_TRY_HIP(hipMemsetAsync(d_err, 0, std::max<size_t>(H, 64) * sizeof(double), s));
    float4 *d_mf = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_mf, ransac_padded(std::max(N, 1u))));
    hipLaunchKernelGGL(ransac_coord_max_kernel, dim3(1), dim3(1024), 0, s, reinterpret_cast<const uint4 *>(d_m), N,
                       d_live + H + 3 + TIED_CAP, d_mf);
    CountMfmaWs ws;
    if (dev->d.ransac_count_mfma) CVHIP_TRY_HIP(alloc_count_mfma(mem, N, H, ws));
    launch_ransac_score_round(d_F, H, d_m, d_m, d_mf, N, t, d_live, d_live + H, d_live + H + 1, d_live + H + 3 + TIED_CAP, false, false, 0u, d_best,
                              d_cnt, d_err, s, nullptr, 0u, 0xFFFFFFFFu, dev->d.ransac_count_mfma ? &ws : nullptr);
    CVHIP_TRY_HIP(hipGetLastError());
    CVHIP_TRY_HIP(hipMemcpyAsync(out_count, d_cnt, (size_t)H * sizeof(uint32_t), dev_ptr(out_count) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipMemcpyAsync(out_err_sum, d_err, (size_t)H * sizeof(double), dev_ptr(out_err_sum) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipStreamSynchronize(s));
    return CVHIP_OK;
}

// Test hook: the device loops' ROUNDS on caller-given hypotheses - `rounds` consecutive slices of F, each scored as
// ransac_rounds scores a generated round: live-slot compaction, the counting kernel on its own, re-sorted copy of the
// match list with the abandonment bound of the best so far and the launch's running maximum, the candidate list,
// ransac_round_finish_kernel (maximum list, tie-break sums, pick, re-sort).  -> the best hypothesis after the last
// round as the loops would carry it: its matrix, inlier count and mean error (NaN where it was never needed), and
// its index in F (-1: no hypothesis reached min_count).  Must equal Ord's (:623-649) maximum over the hypotheses'
// (count, mean error) of cvhip_ransac_score, first of equals by index.
extern "C" int cvhip_ransac_rounds_pick(cvhip_device *dev, const double *F, uint32_t H, uint32_t rounds, const uint32_t *matches,
                                        uint32_t N, double t, uint32_t min_count, double *out_F, uint32_t *out_count,
                                        double *out_mean_error, int64_t *out_index)
{
    if (!dev || !F || !matches || !out_F || !out_count || !out_mean_error || !out_index) return fail(CVHIP_ERR_INVALID, "null argument");
    if (H == 0 || N == 0 || rounds == 0 || rounds > H) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_rounds_pick: empty input");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    const uint32_t per = (H + rounds - 1) / rounds; // hypotheses per round (the last round may be shorter)
    double *d_F = nullptr, *d_err = nullptr;
    uint32_t *d_m = nullptr, *d_mo = nullptr, *d_cnt = nullptr, *d_live = nullptr, *d_cand = nullptr;
    RansacBest *d_best = nullptr;
    float4 *d_mf = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_F, (size_t)H * 9));
    CVHIP_TRY_HIP(mem.alloc(&d_err, std::max<size_t>(per, 64)));
    CVHIP_TRY_HIP(mem.alloc(&d_m, (size_t)N * 4));
    CVHIP_TRY_HIP(mem.alloc(&d_mo, (size_t)N * 4));
    CVHIP_TRY_HIP(mem.alloc(&d_cnt, per));
    const size_t live_words = (size_t)per + 1 + (per + 1023) / 1024;
    CVHIP_TRY_HIP(mem.alloc(&d_live, live_words + 3 + TIED_CAP));
    CVHIP_TRY_HIP(mem.alloc(&d_cand, 2 + 2 * (size_t)TIED_CAP));
    CVHIP_TRY_HIP(mem.alloc(&d_best, 1));
    CVHIP_TRY_HIP(mem.alloc(&d_mf, ransac_padded(N)));
    CountMfmaWs ws;
    if (dev->d.ransac_count_mfma) CVHIP_TRY_HIP(alloc_count_mfma(mem, N, per, ws));
    uint32_t *const d_tied = d_live + live_words, *const d_coord_max = d_tied + 2 + TIED_CAP;
    CVHIP_TRY_HIP(hipMemcpyAsync(d_F, F, (size_t)H * 9 * sizeof(double), dev_ptr(F) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_mo, d_m, (size_t)N * 16, hipMemcpyDeviceToDevice, s));
    CVHIP_TRY_HIP(hipMemsetAsync(d_best, 0, sizeof(RansacBest), s));
    CVHIP_TRY_HIP(hipMemsetAsync(d_cand, 0, 2 * sizeof(uint32_t), s));
    const uint4 *m4 = reinterpret_cast<const uint4 *>(d_m);
    hipLaunchKernelGGL(ransac_coord_max_kernel, dim3(1), dim3(1024), 0, s, m4, N, d_coord_max, d_mf);
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t first = r * per, n = std::min(per, H - std::min(H, first));
        if (n == 0) break;
        const double *F_round = d_F + (size_t)first * 9;
        launch_ransac_live(F_round, n, d_live, d_live + per, d_live + per + 1, s);
        launch_ransac_score_round(F_round, n, d_m, d_mo, d_mf, N, t, d_live, d_live + per, d_tied, d_coord_max, true, true, min_count, d_best,
                                  d_cnt, d_err, s, d_cand, 0u, 0xFFFFFFFFu, dev->d.ransac_count_mfma ? &ws : nullptr);
        hipLaunchKernelGGL(ransac_round_finish_kernel, dim3(1), dim3(1024), 0, s, F_round, m4, N, t, d_err, min_count, d_cand, d_tied, d_best,
                           reinterpret_cast<uint4 *>(d_mo), reinterpret_cast<float *>(d_mf), first);
    }
    CVHIP_TRY_HIP(hipGetLastError());
    RansacBest h_best;
    CVHIP_TRY_HIP(hipMemcpyAsync(&h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipStreamSynchronize(s));
    *out_index = -1;
    *out_count = 0;
    *out_mean_error = std::nan("");
    for (int i = 0; i < 9; i++) out_F[i] = 0.0;
    if (!h_best.valid) return CVHIP_OK;
    std::memcpy(out_F, h_best.f, sizeof(h_best.f));
    *out_count = h_best.matches_count;
    if (h_best.err_known) *out_mean_error = h_best.best_error;
    if (!dev_ptr(F)) // (the index: the first hypothesis with exactly these nine doubles)
        for (uint32_t h = 0; h < H && *out_index < 0; h++)
            if (std::memcmp(F + (size_t)h * 9, h_best.f, sizeof(h_best.f)) == 0) *out_index = (int64_t)h;
    return CVHIP_OK;
}

// The listener of the cvhip_find_ransac call in progress on this thread (fundamentalmatrix.rs:41-47, 103): the model
// entry points below are reached through it and report once per round.
namespace {
struct RansacListener {
    cvhip_progress_fn progress = nullptr;
    cvhip_matches_fn matches = nullptr;
    void *user = nullptr;
    uint64_t max_matches = 0;
    bool wants_counts() const { return matches != nullptr; }
    void count_seen(uint64_t best_count) // max_matches.fetch_max(count), :126-131
    {
        if (!matches) return;
        max_matches = std::max(max_matches, best_count);
        matches(user, max_matches);
    }
    void round_done(uint32_t finished, uint32_t total, bool have_count, uint64_t best_count)
    {
        if (progress) progress(user, (float)finished / (float)total); // counter / ransac_k, :119-123
        if (matches && have_count) {                                  // max_matches.fetch_max(count), :126-131
            max_matches = std::max(max_matches, best_count);
            matches(user, max_matches);
        }
    }
};
thread_local RansacListener g_listener;
} // namespace

extern "C" int cvhip_ransac_affine(cvhip_device *dev, const uint32_t *matches, uint32_t N, uint64_t seed,
                                   double *out_F, uint32_t *out_inlier_count, uint8_t *out_inlier_mask)
{
    // constants of the affine model, fundamentalmatrix.rs:16-30
    constexpr uint32_t RANSAC_K = 1000000, CHECK_INTERVAL = 50000, RANSAC_N = 4, RANSAC_D = 10, EARLY_EXIT = 1000,
                       TOP_INLIERS = 5000;
    constexpr double RANSAC_T = 0.1;
    if (!dev || !matches || !out_F) return fail(CVHIP_ERR_INVALID, "null argument");
    if (N < RANSAC_D + RANSAC_N) return fail(CVHIP_ERR_NO_MODEL, "Not enough matches"); // :107-109
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    uint32_t *d_m = nullptr, *d_cnt = nullptr;
    double *d_F = nullptr, *d_err = nullptr;
    RansacBest *d_best = nullptr;
    uint8_t *d_mask = nullptr;
    hipError_t e = hipMalloc(&d_m, (size_t)N * 16);
    if (e == hipSuccess) e = hipMalloc(&d_F, (size_t)CHECK_INTERVAL * 9 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (size_t)CHECK_INTERVAL * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_err, (size_t)CHECK_INTERVAL * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_best, sizeof(RansacBest));
    if (e == hipSuccess) e = hipMalloc(&d_mask, N);
    uint32_t *d_live = nullptr;
    if (e == hipSuccess) e = hipMalloc(&d_live, ((size_t)CHECK_INTERVAL + 4 + TIED_CAP) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_best, 0, sizeof(RansacBest), s);
    float4 *d_mf = nullptr;
    if (e == hipSuccess) e = hipMalloc(&d_mf, (size_t)ransac_padded(N) * sizeof(float4));
    if (e == hipSuccess)
        hipLaunchKernelGGL(ransac_coord_max_kernel, dim3(1), dim3(1024), 0, s, reinterpret_cast<const uint4 *>(d_m), N,
                           d_live + CHECK_INTERVAL + 3 + TIED_CAP, d_mf);
    RansacBest h_best;
    std::memset(&h_best, 0, sizeof(h_best));
    const uint4 *m4 = reinterpret_cast<const uint4 *>(d_m);
    for (uint32_t round = 0; e == hipSuccess && round < RANSAC_K / CHECK_INTERVAL; round++) {
        hipLaunchKernelGGL(ransac_generate_affine_kernel, dim3((CHECK_INTERVAL + 63) / 64), dim3(64), 0, s, m4,
                           std::min(N, TOP_INLIERS), RANSAC_T, (unsigned long long)seed, round, CHECK_INTERVAL,
                           (const uint32_t *)nullptr, d_F);
        launch_ransac_score_round(d_F, CHECK_INTERVAL, d_m, d_m, d_mf, N, RANSAC_T, d_live, d_live + CHECK_INTERVAL,
                                  d_live + CHECK_INTERVAL + 1, d_live + CHECK_INTERVAL + 3 + TIED_CAP, false, true, RANSAC_D + RANSAC_N, d_best,
                                  d_cnt, d_err, s);
        hipLaunchKernelGGL(ransac_pick_best_approx_kernel, dim3(1), dim3(1024), 0, s, d_F, m4, N, RANSAC_T, (const uint32_t *)d_cnt, d_err,
                           RANSAC_D + RANSAC_N, (const uint32_t *)(d_live + CHECK_INTERVAL + 1), d_best, (uint4 *)nullptr, (float *)nullptr,
                           round * CHECK_INTERVAL);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) g_listener.round_done(round + 1, RANSAC_K / CHECK_INTERVAL, true, h_best.valid ? h_best.matches_count : 0);
        if (h_best.valid && h_best.matches_count > EARLY_EXIT) break; // :135-141
    }
    int rc = CVHIP_OK;
    if (e == hipSuccess && !h_best.valid) rc = fail(CVHIP_ERR_NO_MODEL, "No reliable matches found"); // :145
    if (e == hipSuccess && rc == CVHIP_OK) {
        std::memcpy(out_F, h_best.f, sizeof(h_best.f));
        hipLaunchKernelGGL(ransac_inlier_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_best, m4, N, RANSAC_T,
                           d_mask);
        e = hipGetLastError();
        std::vector<uint8_t> h_mask(N);
        if (e == hipSuccess) e = hipMemcpyAsync(h_mask.data(), d_mask, N, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < N; i++) cnt += h_mask[i];
            if (out_inlier_count) *out_inlier_count = cnt;
            if (out_inlier_mask) std::memcpy(out_inlier_mask, h_mask.data(), N);
        }
    }
    (void)hipFree(d_m);
    (void)hipFree(d_F);
    (void)hipFree(d_cnt);
    (void)hipFree(d_err);
    (void)hipFree(d_best);
    (void)hipFree(d_mask);
    (void)hipFree(d_live);
    (void)hipFree(d_mf);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("ransac_affine: ") + hipGetErrorString(e));
    return rc;
}

// Shared driver of the two RANSAC models: rounds of `per_round` samples x `slots` hypotheses each.
namespace {
// optimize_result's tail on the device (defined behind ransac_refit_kernel): d_mask holds the winner's inlier mask on entry
// and the refitted matrix' on return, d_F_out [9] the refitted matrix (the winner itself where the refit declines)
hipError_t launch_refit_tail(DevAllocs &mem, const uint4 *m4, uint32_t N, double t, uint8_t *d_mask, const double *d_F_in,
                             double *d_F_out, hipStream_t s);
// Hypotheses are generated GEN_BATCH rounds at a time (the generator kernels are bound by the latency of their serial f64
// work on too few threads for the chip - 50 000 samples are 782 waves - so two rounds in one launch take what one
// takes), into one of GEN_DEPTH buffers, on one of GEN_STREAMS side streams of the handle.  Round 4: four streams and a
// buffer per batch of the reference's twenty rounds - the generators' long Levenberg-Marquardt tails run side by side
// instead of one behind the other, and the call's last batch is generated ~3 ms earlier, which is what its stragglers
// (LateStage, below) need to finish under the scoring chain: config 5's RANSAC stage 50.9 -> 45.2 ms with the reference's
// pencil, 15.8 -> 15.4 with the null-space one (streams / buffers / rounds per batch: 2/3/4 50.9, 3/4/4 48.0, 3/5/4 47.2,
// 4/5/4 45.2, 5/5/4 45.3, 6/6/4 45.4, 5/6/2 48.5, 4/4/5 48.8 ms; five buffers on two streams had changed nothing).
#ifndef CVHIP_GEN_BATCH
#define CVHIP_GEN_BATCH 4
#endif
#ifndef CVHIP_GEN_DEPTH
#define CVHIP_GEN_DEPTH 5
#endif
#ifndef CVHIP_GEN_STREAMS
#define CVHIP_GEN_STREAMS 4
#endif
constexpr uint32_t GEN_BATCH = CVHIP_GEN_BATCH, GEN_DEPTH = CVHIP_GEN_DEPTH;
// The call's stragglers (thin-SVD pencil; batched scoring only): roots whose Levenberg-Marquardt loop outlasts the thread
// kernels' budget - a handful per 100 000, but each keeps ONE wave busy for up to 1000 trips (~5 ms), and run inside their
// batch they held that batch's generator stream, and with it every later batch, for that long (2.6 of the 6.7 ms a batch
// spent on its stream).  They are collected on one list per call (sample + start parameters), run ONCE behind the last
// batch's generation - under the scoring of the last batches - and their hypotheses are scored as one more small round.
// Ord's maximum does not depend on when a hypothesis is met (see below); among EQUAL ones a straggler counts as met last.
struct LateStage {
    uint32_t *list = nullptr; // [0] = count; LateRoot entries behind 256 bytes
    uint32_t cap = 0;
    double *F = nullptr;      // cap x 9: the stragglers' hypotheses (NaN = none)
    std::function<void(hipStream_t)> run;
};
template <typename Generate>
int ransac_rounds(cvhip_device *dev, DevAllocs &mem, const uint32_t *matches, uint32_t N, uint32_t rounds, uint32_t per_round, uint32_t slots,
                  double t, uint32_t min_count, uint32_t early_exit, double *out_F, uint32_t *out_inlier_count,
                  uint8_t *out_inlier_mask, const char *what, bool refit_tail, Generate generate, const LateStage *late = nullptr)
{
    hipStream_t s = dev->d.stream;
    const uint32_t H = per_round * slots;
    uint32_t *d_m = nullptr, *d_cnt = nullptr;
    double *d_F = nullptr, *d_err = nullptr;
    RansacBest *d_best = nullptr;
    uint8_t *d_mask = nullptr;
    hipError_t e = mem.alloc(&d_m, (size_t)N * 4);
    if (e == hipSuccess) e = mem.alloc(&d_F, GEN_DEPTH * GEN_BATCH * (size_t)H * 9); // the hypotheses of the rounds in flight
    if (e == hipSuccess) e = mem.alloc(&d_cnt, GEN_BATCH * (size_t)H);
    if (e == hipSuccess) e = mem.alloc(&d_err, GEN_BATCH * (size_t)H);
    if (e == hipSuccess) e = mem.alloc(&d_best, 1);
    if (e == hipSuccess) e = mem.alloc(&d_mask, N);
    // per hypothesis buffer: the live list [H], its length [1] and the compaction's scratch; then the round's maximum list
    // [2 + TIED_CAP] and the largest coordinate [1]
    // (a batch scored as ONE round - see below - has one list of GEN_BATCH x H slots: it fits the space of its rounds' lists)
    const size_t live_words = (size_t)H + 1 + (H + 1023) / 1024;
    static_assert(GEN_BATCH >= 1, "");
    uint32_t *d_live = nullptr;
    if (e == hipSuccess) e = mem.alloc(&d_live, GEN_DEPTH * GEN_BATCH * live_words + 3 + TIED_CAP);
    uint32_t *const d_tied = d_live + GEN_DEPTH * GEN_BATCH * live_words, *const d_coord_max = d_tied + 2 + TIED_CAP;
    uint32_t *d_late_live = nullptr; // the stragglers' round: live list [cap], its length, the compaction's scratch
    if (e == hipSuccess && late) e = mem.alloc(&d_late_live, (size_t)late->cap + 1 + (late->cap + 1023) / 1024);
    float4 *d_mf = nullptr; // the counting kernel's copy of the list (f32 planes + u32), reordered as the best hypothesis changes
    uint32_t *d_mo = nullptr;
    if (e == hipSuccess) e = mem.alloc(&d_mf, ransac_padded(N));
    if (e == hipSuccess) e = mem.alloc(&d_mo, (size_t)N * 4);
    uint32_t *d_cand = nullptr; // the counting kernel's running maximum and candidate list (ransac_round_finish_kernel clears it per round)
    if (e == hipSuccess) e = mem.alloc(&d_cand, 2 + 2 * (size_t)TIED_CAP);
    CountMfmaWs mfma_ws; // the counting screen's phase on the matrix pipe (cvhip_ransac_set_count_mfma)
    const CountMfmaWs *const ws = dev->d.ransac_count_mfma ? &mfma_ws : nullptr;
    if (e == hipSuccess && ws) e = alloc_count_mfma(mem, N, std::max<uint32_t>(GEN_BATCH * H, late ? late->cap : 0u), mfma_ws);
    if (e == hipSuccess) e = hipMemsetAsync(d_cand, 0, 2 * sizeof(uint32_t), s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    if (e == hipSuccess && late) e = hipMemsetAsync(late->list, 0, sizeof(uint32_t), s);
    if (e == hipSuccess && late) e = hipMemsetAsync(late->F, 0xFF, (size_t)late->cap * 9 * sizeof(double), s); // (all ones: NaN)
    // (the generators need the list and nothing else of what follows: their event goes right behind the upload)
    if (e == hipSuccess && !dev->d.rq.uploaded) e = hipEventCreateWithFlags(&dev->d.rq.uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(dev->d.rq.uploaded, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_best, 0, sizeof(RansacBest), s);
    if (e == hipSuccess) // the scale of the counting kernel's f32 screen: one word behind the round's maximum list
        hipLaunchKernelGGL(ransac_coord_max_kernel, dim3(1), dim3(1024), 0, s, reinterpret_cast<const uint4 *>(d_m), N,
                           d_coord_max, d_mf);
    if (e == hipSuccess) e = hipMemcpyAsync(d_mo, d_m, (size_t)N * 16, hipMemcpyDeviceToDevice, s);
    RansacBest h_best;
    std::memset(&h_best, 0, sizeof(h_best));
    uint32_t late_count = 0;
    const uint4 *m4 = reinterpret_cast<const uint4 *>(d_m);
    // GEN_STREAMS side streams, GEN_DEPTH hypothesis buffers of GEN_BATCH rounds: the batches after the one being SCORED (the handle's
    // stream; the best-so-far chain lives there) are GENERATED on the side streams (samples depend on the seed and the
    // round number only), one batch per stream at a time.  The generator's tail - a few long Levenberg-Marquardt loops
    // on a few hundred waves that need a whole SIMD's registers each - cannot get onto the chip while the counting kernel
    // fills it; generated ahead, it is already queued when a counting kernel drains, starts in the gap that follows
    // (ordered sums, best pick, the host's early-exit read) and finishes under the next counting kernels.  With single
    // rounds generated two ahead the scoring chain waited for its generators in most rounds (a round's generation took
    // ~330 us, more with a long LM tail, against ~190 us of scoring: 7.3 ms per pair for 3.8 ms of scoring).  An early
    // exit discards at most two generated batches.
    // The early exit (:135-141) needs more than `early_exit` inliers: with fewer matches than that it can never fire, and
    // the rounds are enqueued back to back with ONE host synchronisation at the end instead of a 4-byte read per round.
    const bool may_exit_early = N > early_exit;
    // Without the early exit (and without a listener that wants the count after every round) nothing observes the state
    // between the rounds of a batch, and Ord's result over a batch is the same whether its rounds are scored one after
    // the other or as ONE round of GEN_BATCH x H slots (a later hypothesis replaces the best only if strictly better, the
    // smaller slot stays among equals - and round q's slots follow round q - 1's): half as many counting launches and
    // round ends (the latter are single-workgroup kernels: 20 - 70 us of an otherwise idle chip each).
    const bool score_batches = !may_exit_early;
    constexpr uint32_t GEN_STREAMS = CVHIP_GEN_STREAMS;
    static_assert(GEN_DEPTH + 2 <= sizeof(Device::RansacQueues::ready) / sizeof(hipEvent_t), // (ready[6], ready[7]: the stragglers' events)
                  "RansacQueues holds eight events per kind");
    static_assert(GEN_STREAMS <= sizeof(Device::aux) / sizeof(hipStream_t), "Device holds the side streams");
    Device::RansacQueues &rq = dev->d.rq; // (kept on the handle: created once)
    hipStream_t g[GEN_STREAMS] = {};
    for (uint32_t k = 0; k < GEN_STREAMS && e == hipSuccess; k++) e = aux_stream(dev->d, (int)k, &g[k]);
    for (uint32_t b = 0; b < GEN_DEPTH && e == hipSuccess; b++) {
        if (!rq.ready[b]) e = hipEventCreateWithFlags(&rq.ready[b], hipEventDisableTiming);
        if (e == hipSuccess && !rq.scored[b]) e = hipEventCreateWithFlags(&rq.scored[b], hipEventDisableTiming);
    }
    hipEvent_t *const ready = rq.ready, *const scored = rq.scored;
    const hipEvent_t uploaded = rq.uploaded;
    for (uint32_t k = 0; k < GEN_STREAMS && e == hipSuccess; k++) e = hipStreamWaitEvent(g[k], uploaded, 0);
    const uint32_t units = (rounds + GEN_BATCH - 1) / GEN_BATCH;
    // batch u into buffer b on generator stream k, once the buffer's last reader is done (wait_scored: it had one)
    const auto generate_unit_on = [&](uint32_t u, uint32_t b, uint32_t k, bool wait_scored) {
        const uint32_t r0 = u * GEN_BATCH, nr = std::min(GEN_BATCH, rounds - r0);
        hipStream_t gs = g[k];
        hipError_t ge = wait_scored ? hipStreamWaitEvent(gs, scored[b], 0) : hipSuccess;
        if (ge != hipSuccess) return ge;
        double *F_unit = d_F + (size_t)b * GEN_BATCH * H * 9;
        generate(m4, r0, nr, (int)b, F_unit, gs, score_batches && late != nullptr);
        // the rounds' live lists, right behind their generation: three small launches less on the scoring chain
        if (score_batches) {
            uint32_t *lv = d_live + (size_t)b * GEN_BATCH * live_words;
            launch_ransac_live(F_unit, nr * H, lv, lv + nr * H, lv + nr * H + 1, gs);
        } else {
            for (uint32_t q = 0; q < nr; q++) {
                uint32_t *lv = d_live + ((size_t)b * GEN_BATCH + q) * live_words;
                launch_ransac_live(F_unit + (size_t)q * H * 9, H, lv, lv + H, lv + H + 1, gs);
            }
        }
        return hipEventRecord(ready[b], gs);
    };
    const auto generate_unit = [&](uint32_t u) { return generate_unit_on(u, u % GEN_DEPTH, u % GEN_STREAMS, u >= GEN_DEPTH); };
    if (score_batches) {
        // Batches are scored AS THEY BECOME READY, not in order.  A batch whose generator is held up - one
        // Levenberg-Marquardt loop that runs to the reference's cap of 1000 iterations keeps its wave busy for 1.6 ms -
        // is passed over until it is done or its buffer is needed; scored in order, the chain stood still behind it
        // (~0.5 ms per pair on config 5, where such a loop sits in the second batch).  Ord's maximum does not depend on
        // the order: more matches win, then the smaller mean error, and among EQUAL hypotheses the one the reference
        // meets first - RansacBest::origin, the hypothesis' position in iteration order, decides that
        // (ransac_pick_best_approx).  The counting kernel's bound only needs SOME best-so-far.  The host enqueues a
        // batch as soon as it is ready - usually while the previous one is still being scored.
        // Buffers and generator streams are handed out as they come free: the stream behind a long loop is not given
        // the next batch while the other one is idle, and the batch that is passed over does not block the buffer of the
        // batch three further on.
        std::vector<char> done(units, 0);
        std::vector<uint32_t> buf_of(units, 0);
        uint32_t next_gen = 0, n_done = 0, reported = 0, turn = 0;
        int buf_unit[GEN_DEPTH], stream_unit[GEN_STREAMS]; // the batch a buffer holds / a stream generated last (-1: none yet)
        bool buf_used[GEN_DEPTH];
        for (uint32_t b = 0; b < GEN_DEPTH; b++) {
            buf_unit[b] = -1;
            buf_used[b] = false;
        }
        for (uint32_t k = 0; k < GEN_STREAMS; k++) stream_unit[k] = -1;
        const auto generated = [&](int u) { // has batch u left its generator stream?
            if (u < 0 || done[(uint32_t)u]) return true;
            const hipError_t qe = hipEventQuery(ready[buf_of[(uint32_t)u]]);
            (void)hipGetLastError();
            if (qe != hipSuccess && qe != hipErrorNotReady && e == hipSuccess) e = qe; // (a device error: the polling loops below end on it)
            return qe == hipSuccess;
        };
        const auto top_up = [&]() {
            while (e == hipSuccess && next_gen < units) {
                uint32_t b = GEN_DEPTH;
                for (uint32_t c = 0; c < GEN_DEPTH && b == GEN_DEPTH; c++)
                    if (buf_unit[c] < 0 || done[(uint32_t)buf_unit[c]]) b = c; // never used, or its batch has been scored
                if (b == GEN_DEPTH) break;
                uint32_t k = GEN_STREAMS;
                for (uint32_t c = 0; c < GEN_STREAMS && k == GEN_STREAMS; c++)
                    if (generated(stream_unit[(turn + c) % GEN_STREAMS])) k = (turn + c) % GEN_STREAMS; // an idle stream, in turn
                if (k == GEN_STREAMS) break; // all are busy - one of them perhaps for long: decided when one comes free
                turn = k + 1;
                e = generate_unit_on(next_gen, b, k, buf_used[b]);
                buf_of[next_gen] = b;
                buf_unit[b] = (int)next_gen;
                buf_used[b] = true;
                stream_unit[k] = (int)next_gen;
                next_gen++;
            }
        };
        // The call's stragglers (LateStage): once the last batch is on its generator stream, the list is complete when both
        // generator streams have drained; the wave-per-root kernel goes onto stream 0 behind them, under the scoring of the
        // last batches.
        bool late_enqueued = false;
        const auto enqueue_late = [&]() {
            if (!late || late_enqueued || next_gen < units || e != hipSuccess) return;
            late_enqueued = true;
            for (uint32_t k = 1; k < GEN_STREAMS && e == hipSuccess; k++) {
                if (!rq.ready[6]) e = hipEventCreateWithFlags(&rq.ready[6], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(rq.ready[6], g[k]);
                if (e == hipSuccess) e = hipStreamWaitEvent(g[0], rq.ready[6], 0);
            }
            if (e != hipSuccess) return;
            late->run(g[0]);
            launch_ransac_live(late->F, late->cap, d_late_live, d_late_live + late->cap, d_late_live + late->cap + 1, g[0]);
            if (!rq.ready[7]) e = hipEventCreateWithFlags(&rq.ready[7], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(rq.ready[7], g[0]);
        };
        // A report_matches listener (fundamentalmatrix.rs:126-131: the running maximum of the inlier counts, shown in the
        // progress bar's message, reconstruction.rs:859-863) gets its count per scored BATCH: the best count goes to
        // page-locked memory behind the batch's pick, an event marks it, and the host reports what has arrived whenever it
        // is here anyway - no synchronisation per round, the batched path survives the listener the reference always passes.
        uint32_t *h_counts = nullptr;
        std::vector<hipEvent_t> count_ev;
        uint32_t counts_reported = 0;
        if (g_listener.wants_counts()) {
            h_counts = static_cast<uint32_t *>(pinned_scratch(dev->d, (size_t)units * sizeof(uint32_t)));
            if (!h_counts) e = hipErrorOutOfMemory;
            count_ev.resize(units, nullptr);
            for (uint32_t u = 0; u < units && e == hipSuccess; u++) e = hipEventCreateWithFlags(&count_ev[u], hipEventDisableTiming);
        }
        uint32_t scored_order = 0; // batches in the order they were scored: their counts are reported in that order
        const auto report_counts = [&](bool wait) {
            while (h_counts && counts_reported < scored_order && e == hipSuccess) {
                const hipError_t qe = wait ? hipEventSynchronize(count_ev[counts_reported]) : hipEventQuery(count_ev[counts_reported]);
                if (qe == hipErrorNotReady) {
                    (void)hipGetLastError();
                    break;
                }
                if (qe != hipSuccess) {
                    e = qe;
                    break;
                }
                g_listener.count_seen(h_counts[counts_reported]);
                counts_reported++;
            }
        };
        top_up();
        enqueue_late();
        while (e == hipSuccess && n_done < units) {
            // the oldest batch that is ready; with several to choose from and none ready yet, the host polls (the
            // scoring chain has nothing to run then anyway); a single candidate is simply enqueued behind its event.
            // The polling is bounded: after POLL_LIMIT_MS without a ready batch the oldest pending one is enqueued behind
            // its event - in-order waiting on the stream, no spinning - so a generator that is stuck (not failed: a device
            // error ends the loop through `e`) cannot keep a host core busy without bound.
            constexpr double POLL_LIMIT_MS = 50.0;
            uint32_t pick = 0xFFFFFFFFu, pending = 0, oldest = 0xFFFFFFFFu;
            for (uint32_t u = 0; u < next_gen; u++)
                if (!done[u]) {
                    pending++;
                    if (oldest == 0xFFFFFFFFu) oldest = u;
                }
            const auto poll_start = std::chrono::steady_clock::now();
            while (pick == 0xFFFFFFFFu && e == hipSuccess) {
                for (uint32_t u = 0; u < next_gen && pick == 0xFFFFFFFFu; u++)
                    if (!done[u] && generated((int)u)) pick = u;
                if (pick == 0xFFFFFFFFu && pending == 1 && next_gen == units) pick = oldest; // the last one: nothing to decide
                if (pick == 0xFFFFFFFFu && (dev->d.ransac_in_order || std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - poll_start).count() > POLL_LIMIT_MS))
                    pick = oldest; // (cvhip_ransac_set_in_order: the test hook that forces this branch)
                if (pick == 0xFFFFFFFFu) {
                    std::this_thread::yield();
                    report_counts(false);
                    const uint32_t before = next_gen;
                    top_up(); // (a generator stream may have come free)
                    enqueue_late();
                    pending += next_gen - before;
                    if (oldest == 0xFFFFFFFFu && next_gen > before) oldest = before;
                }
            }
            if (e != hipSuccess) break;
            const uint32_t b = buf_of[pick], r0 = pick * GEN_BATCH, nr = std::min(GEN_BATCH, rounds - r0), HS = nr * H;
            double *F_unit = d_F + (size_t)b * GEN_BATCH * H * 9;
            uint32_t *lv = d_live + (size_t)b * GEN_BATCH * live_words;
            e = hipStreamWaitEvent(s, ready[b], 0);
            // (the first batch scored has no best hypothesis to be abandoned against: its first live hypotheses go
            // first, as a round of their own - see the round-by-round loop below)
            constexpr uint32_t HEAD = 2048;
            const uint32_t parts = n_done == 0 && HS > 4 * HEAD ? 2u : 1u;
            for (uint32_t part = 0; part < parts && e == hipSuccess; part++) {
                const uint32_t first = part == 0 ? 0u : HEAD, end = parts == 2 && part == 0 ? HEAD : 0xFFFFFFFFu;
                launch_ransac_score_round(F_unit, HS, d_m, d_mo, d_mf, N, t, lv, lv + HS, d_tied, d_coord_max, true, true, min_count, d_best, d_cnt,
                                          d_err, s, d_cand, first, end, ws);
                hipLaunchKernelGGL(ransac_round_tied_list_kernel, dim3(1), dim3(1024), 0, s, d_cand, d_tied);
                hipLaunchKernelGGL(ransac_tied_approx_kernel, dim3(16), dim3(1024), 0, s, F_unit, m4, N, t, (const uint32_t *)d_tied, d_best, d_err);
                hipLaunchKernelGGL(ransac_pick_best_approx_kernel, dim3(1), dim3(1024), 0, s, F_unit, m4, N, t, (const uint32_t *)nullptr, d_err,
                                   min_count, (const uint32_t *)d_tied, d_best, reinterpret_cast<uint4 *>(d_mo), reinterpret_cast<float *>(d_mf),
                                   r0 * H);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipEventRecord(scored[b], s);
            if (e == hipSuccess && h_counts) { // (matches_count of an invalid best is 0)
                e = hipMemcpyAsync(&h_counts[scored_order], &d_best->matches_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipEventRecord(count_ev[scored_order], s);
            }
            scored_order++;
            done[pick] = 1;
            n_done++;
            for (uint32_t q = 0; q < nr; q++)
                if (++reported < rounds) g_listener.round_done(reported, rounds, false, 0); // position only: enqueued, not finished
            report_counts(false);
            top_up();
            enqueue_late();
        }
        enqueue_late();
        if (e == hipSuccess && late && late_enqueued) { // the stragglers' hypotheses: one more round, met last
            e = hipStreamWaitEvent(s, rq.ready[7], 0);
            launch_ransac_score_round(late->F, late->cap, d_m, d_mo, d_mf, N, t, d_late_live, d_late_live + late->cap, d_tied, d_coord_max, true, true,
                                      min_count, d_best, d_cnt, d_err, s, d_cand, 0u, 0xFFFFFFFFu, ws);
            hipLaunchKernelGGL(ransac_round_tied_list_kernel, dim3(1), dim3(1024), 0, s, d_cand, d_tied);
            hipLaunchKernelGGL(ransac_tied_approx_kernel, dim3(16), dim3(1024), 0, s, late->F, m4, N, t, (const uint32_t *)d_tied, d_best, d_err);
            hipLaunchKernelGGL(ransac_pick_best_approx_kernel, dim3(1), dim3(1024), 0, s, late->F, m4, N, t, (const uint32_t *)nullptr, d_err, min_count,
                               (const uint32_t *)d_tied, d_best, reinterpret_cast<uint4 *>(d_mo), reinterpret_cast<float *>(d_mf), rounds * H);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&late_count, late->list, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        report_counts(true);
        for (hipEvent_t ev : count_ev)
            if (ev) (void)hipEventDestroy(ev);
        if (e == hipSuccess) g_listener.round_done(rounds, rounds, true, h_best.valid ? h_best.matches_count : 0);
    }
    for (uint32_t u = 0; !score_batches && u + 1 < GEN_DEPTH && u < units && e == hipSuccess; u++) e = generate_unit(u);
    for (uint32_t round = 0; !score_batches && e == hipSuccess && round < rounds; round++) {
        const uint32_t u = round / GEN_BATCH, q = round % GEN_BATCH, b = u % GEN_DEPTH;
        double *F_round = d_F + ((size_t)b * GEN_BATCH + q) * H * 9;
        if (q == 0) {
            if (u + GEN_DEPTH - 1 < units) e = generate_unit(u + GEN_DEPTH - 1);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, ready[b], 0);
        }
        uint32_t *lv = d_live + ((size_t)b * GEN_BATCH + q) * live_words;
        // (a batch as one round: its first round's turn scores all of it, the others only report their position)
        const uint32_t nr = std::min(GEN_BATCH, rounds - u * GEN_BATCH), HS = score_batches ? nr * H : H;
        if (score_batches && q != 0) {
            if (round + 1 < rounds) g_listener.round_done(round + 1, rounds, false, 0);
            if (round + 1 < rounds) continue;
        }
        // The first round has no best hypothesis to be abandoned against and the list in the matcher's order: every
        // hypothesis would fold (nearly) the whole list - 0.48 ms against 0.2 for the later rounds.  Its first
        // ROUND0_HEAD live hypotheses therefore go first, as a round of their own: what they leave behind - a best count,
        // the re-sorted list - is what the other ~23 000 are abandoned by.  (Ord's result is the same: a later
        // hypothesis replaces the carried best only if it is strictly better, the smaller slot stays among equals.)
        constexpr uint32_t ROUND0_HEAD = 2048;
        const uint32_t parts = round == 0 && H > 4 * ROUND0_HEAD ? 2u : 1u;
        for (uint32_t part = 0; part < parts && !(score_batches && q != 0); part++) {
            const uint32_t first = part == 0 ? 0u : ROUND0_HEAD, end = parts == 2 && part == 0 ? ROUND0_HEAD : 0xFFFFFFFFu;
            launch_ransac_score_round(F_round, HS, d_m, d_mo, d_mf, N, t, lv, lv + HS, d_tied, d_coord_max, true, true, min_count, d_best, d_cnt, d_err,
                                      s, d_cand, first, end, ws);
            if (score_batches) {
                hipLaunchKernelGGL(ransac_round_tied_list_kernel, dim3(1), dim3(1024), 0, s, d_cand, d_tied);
                hipLaunchKernelGGL(ransac_tied_approx_kernel, dim3(16), dim3(1024), 0, s, F_round, m4, N, t, (const uint32_t *)d_tied, d_best, d_err);
                hipLaunchKernelGGL(ransac_pick_best_approx_kernel, dim3(1), dim3(1024), 0, s, F_round, m4, N, t, (const uint32_t *)nullptr, d_err,
                                   min_count, (const uint32_t *)d_tied, d_best, reinterpret_cast<uint4 *>(d_mo), reinterpret_cast<float *>(d_mf),
                                   round * H);
            } else {
                hipLaunchKernelGGL(ransac_round_finish_kernel, dim3(1), dim3(1024), 0, s, F_round, m4, N, t, d_err, min_count, d_cand, d_tied, d_best,
                                   reinterpret_cast<uint4 *>(d_mo), reinterpret_cast<float *>(d_mf), round * H);
            }
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && (score_batches ? q == 0 : (q + 1 == nr || round + 1 == rounds))) e = hipEventRecord(scored[b], s);
        if (score_batches && round + 1 < rounds) {
            g_listener.round_done(round + 1, rounds, false, 0); // position only: the round is enqueued, not finished
            continue;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) g_listener.round_done(round + 1, rounds, true, h_best.valid ? h_best.matches_count : 0);
        if (h_best.valid && h_best.matches_count > early_exit) break; // :135-141
    }
    for (uint32_t k = 0; k < GEN_STREAMS; k++)
        if (g[k]) (void)hipStreamSynchronize(g[k]); // rounds generated ahead of an early exit, or of an error
    int rc = CVHIP_OK;
    if (e == hipSuccess && late && late_count > late->cap)
        rc = fail(CVHIP_ERR_DEVICE, std::string(what) + ": more Levenberg-Marquardt stragglers than the call's list holds");
    if (e == hipSuccess && rc == CVHIP_OK && !h_best.valid) rc = fail(CVHIP_ERR_NO_MODEL, "No reliable matches found"); // :145
    if (e == hipSuccess && rc == CVHIP_OK) {
        std::memcpy(out_F, h_best.f, sizeof(h_best.f));
        hipLaunchKernelGGL(ransac_inlier_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_best, m4, N, t, d_mask);
        e = hipGetLastError();
        double *d_Fo = nullptr;
        if (refit_tail) { // optimize_result (:246-254) right behind it, without a round trip through the host
            if (e == hipSuccess) e = mem.alloc(&d_Fo, 9);
            if (e == hipSuccess) e = launch_refit_tail(mem, m4, N, t, d_mask, reinterpret_cast<const double *>(d_best), d_Fo, s);
            if (e == hipSuccess) e = hipMemcpyAsync(out_F, d_Fo, 9 * sizeof(double), hipMemcpyDeviceToHost, s);
        }
        std::vector<uint8_t> h_mask(N);
        if (e == hipSuccess) e = hipMemcpyAsync(h_mask.data(), d_mask, N, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < N; i++) cnt += h_mask[i];
            if (out_inlier_count) *out_inlier_count = cnt;
            if (out_inlier_mask) std::memcpy(out_inlier_mask, h_mask.data(), N);
        }
    }
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return rc;
}
} // namespace

namespace {
// refit_tail: optimize_result's refit and second inlier filter (:246-254) follow on the device (cvhip_find_ransac)
int ransac_perspective(cvhip_device *dev, const uint32_t *matches, uint32_t N, double max_dimension, uint64_t seed, uint32_t rounds,
                       double *out_F, uint32_t *out_inlier_count, uint8_t *out_inlier_mask, bool refit_tail)
{
    // constants of the perspective model, fundamentalmatrix.rs:16-30
    constexpr uint32_t RANSAC_K = 1000000, CHECK_INTERVAL = 50000, RANSAC_N = 7, RANSAC_D = 200, EARLY_EXIT = 50000,
                       TOP_INLIERS = 5000;
    if (!dev || !matches || !out_F) return fail(CVHIP_ERR_INVALID, "null argument");
    if (!(max_dimension > 0.0)) return fail(CVHIP_ERR_INVALID, "max_dimension must be positive");
    if (N < RANSAC_D + RANSAC_N) return fail(CVHIP_ERR_NO_MODEL, "Not enough matches"); // :107-109
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    const double t = 10.0 / 1000.0 * max_dimension; // :23, :85
    const uint32_t limit = std::min(N, TOP_INLIERS);
    if (rounds == 0 || rounds > RANSAC_K / CHECK_INTERVAL) rounds = RANSAC_K / CHECK_INTERVAL;
    // per generated round (GEN_DEPTH buffers, see ransac_rounds): the pencils, then the LM queue
    const int pencil = dev->d.ransac_pencil;
    const size_t batch_roots = 3 * GEN_BATCH * (size_t)CHECK_INTERVAL;
    const size_t queue_bytes = ((1 + batch_roots) * sizeof(uint32_t) + 255) / 256 * 256; // every root can be queued
    const size_t gen_bytes = (GEN_BATCH * (size_t)CHECK_INTERVAL * sizeof(PerspPencil) + 255) / 256 * 256 + LM_QUEUES * queue_bytes;
    DevAllocs mem(dev->d);
    char *d_gen = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_gen, GEN_DEPTH * gen_bytes));
    // the call's stragglers (LateStage): thin-SVD pencil with the two-pass LM only
    constexpr uint32_t LATE_CAP = 65536; // (config 5 produces ~150 per call; beyond the cap the call fails, it never drops one)
    LateStage late;
    const bool use_late = pencil == CVHIP_PENCIL_THIN_SVD && dev->d.ransac_lm_pipeline != 0;
    if (use_late) {
        char *d_late = nullptr;
        CVHIP_TRY_HIP(mem.alloc(&d_late, 256 + (size_t)LATE_CAP * sizeof(LateRoot)));
        CVHIP_TRY_HIP(mem.alloc(&late.F, (size_t)LATE_CAP * 9));
        late.list = reinterpret_cast<uint32_t *>(d_late);
        late.cap = LATE_CAP;
        late.run = [&late, t](hipStream_t s) {
            hipLaunchKernelGGL(ransac_perspective_lm_kernel, dim3(4096), dim3(64), 0, s, (const PerspPencil *)nullptr, t, late.F, (const uint32_t *)nullptr,
                               (const uint32_t *)late.list, late.cap);
        };
    }
    const int rc = ransac_rounds(
        dev, mem, matches, N, rounds, CHECK_INTERVAL, 3, t, RANSAC_D + RANSAC_N, EARLY_EXIT, out_F, out_inlier_count, out_inlier_mask,
        "ransac_perspective", refit_tail,
        [&](const uint4 *m4, uint32_t round0, uint32_t n_rounds, int buffer, double *d_F, hipStream_t s, bool defer_late) {
            PerspPencil *pencils = (PerspPencil *)(d_gen + (size_t)buffer * gen_bytes);
            char *queues = (char *)pencils + (GEN_BATCH * (size_t)CHECK_INTERVAL * sizeof(PerspPencil) + 255) / 256 * 256;
            launch_generate_perspective(pencil, dev->d.ransac_lm_pipeline, m4, limit, t, (unsigned long long)seed, round0, CHECK_INTERVAL,
                                        n_rounds * CHECK_INTERVAL, nullptr, pencils, (uint32_t *)queues, queue_bytes / sizeof(uint32_t),
                                        defer_late ? late.list : nullptr, late.cap, d_F, s);
        },
        use_late ? &late : nullptr);
    return rc;
}
} // namespace

extern "C" int cvhip_ransac_set_pencil(cvhip_device *dev, int pencil)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_set_pencil: null device");
    if (pencil != CVHIP_PENCIL_THIN_SVD && pencil != CVHIP_PENCIL_NULL_SPACE) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_set_pencil: unknown mode");
    dev->d.ransac_pencil = pencil;
    return CVHIP_OK;
}

extern "C" int cvhip_ransac_set_count_mfma(cvhip_device *dev, int enable)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_set_count_mfma: null device");
    dev->d.ransac_count_mfma = enable ? 1 : 0;
    return CVHIP_OK;
}

extern "C" int cvhip_ransac_set_in_order(cvhip_device *dev, int enable)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_set_in_order: null device");
    dev->d.ransac_in_order = enable ? 1 : 0;
    return CVHIP_OK;
}

extern "C" int cvhip_ransac_set_lm_pipeline(cvhip_device *dev, int enable)
{
    if (!dev) return fail(CVHIP_ERR_INVALID, "cvhip_ransac_set_lm_pipeline: null device");
    dev->d.ransac_lm_pipeline = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return CVHIP_OK;
}

extern "C" int cvhip_ransac_perspective(cvhip_device *dev, const uint32_t *matches, uint32_t N, double max_dimension,
                                        uint64_t seed, uint32_t rounds, double *out_F, uint32_t *out_inlier_count,
                                        uint8_t *out_inlier_mask)
{
    return ransac_perspective(dev, matches, N, max_dimension, seed, rounds, out_F, out_inlier_count, out_inlier_mask, false);
}

// Test hooks of the two generators: the models of B caller-chosen samples (`per` match indices each).
namespace {
template <typename Launch>
int models_of_samples(cvhip_device *dev, const uint32_t *matches, uint32_t N, const uint32_t *sample_idx, uint32_t B,
                      uint32_t per, uint32_t slots, double *out_F, const char *what, Launch launch)
{
    if (!dev || !matches || !sample_idx || !out_F) return fail(CVHIP_ERR_INVALID, "null argument");
    if (B == 0) return CVHIP_OK;
    for (size_t i = 0; i < (size_t)B * per; i++)
        if (sample_idx[i] >= N) return fail(CVHIP_ERR_INVALID, "sample index out of range");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    uint32_t *d_m = nullptr, *d_idx = nullptr;
    double *d_F = nullptr;
    const size_t f_bytes = (size_t)B * slots * 9 * sizeof(double);
    hipError_t e = hipMalloc(&d_m, (size_t)N * 16);
    if (e == hipSuccess) e = hipMalloc(&d_idx, (size_t)B * per * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_F, f_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, matches, (size_t)N * 16, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, sample_idx, (size_t)B * per * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        launch(reinterpret_cast<const uint4 *>(d_m), (const uint32_t *)d_idx, d_F, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_F, d_F, f_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_m);
    (void)hipFree(d_idx);
    (void)hipFree(d_F);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return CVHIP_OK;
}
} // namespace

extern "C" int cvhip_ransac_perspective_models(cvhip_device *dev, const uint32_t *matches, uint32_t N,
                                               const uint32_t *sample_idx, uint32_t B, double t, double *out_F)
{
    if (!dev || B == 0) return models_of_samples(dev, matches, N, sample_idx, B, 7, 3, out_F, "ransac_perspective_models",
                                                 [](const uint4 *, const uint32_t *, double *, hipStream_t) {});
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    PerspPencil *d_pencils = nullptr;
    const size_t stride = (1 + 3 * (size_t)B + 63) / 64 * 64; // words per queue
    const size_t pencil_bytes = ((size_t)B * sizeof(PerspPencil) + 255) / 256 * 256;
    CVHIP_TRY_HIP(hipMalloc(&d_pencils, pencil_bytes + LM_QUEUES * stride * sizeof(uint32_t)));
    uint32_t *d_queue = (uint32_t *)((char *)d_pencils + pencil_bytes);
    const int pencil = dev->d.ransac_pencil;
    const int rc = models_of_samples(dev, matches, N, sample_idx, B, 7, 3, out_F, "ransac_perspective_models",
                                     [&](const uint4 *m4, const uint32_t *idx, double *d_F, hipStream_t s) {
                                         launch_generate_perspective(pencil, dev->d.ransac_lm_pipeline, m4, N, t, 0ull, 0u, B, B, idx, d_pencils, d_queue, stride, nullptr, 0u, d_F, s);
                                         if (getenv("CVHIP_LM_CENSUS")) { // (diagnostic: how many roots each stage of the LM funnel received)
                                             uint32_t n[LM_QUEUES] = {};
                                             for (int k = 0; k < LM_QUEUES; k++) (void)hipMemcpyAsync(&n[k], d_queue + k * stride, 4, hipMemcpyDeviceToHost, s);
                                             (void)hipStreamSynchronize(s);
                                             fprintf(stderr, "[cvhip] LM census (%u samples): queued", B);
                                             for (int k = 0; k < LM_QUEUES; k++) fprintf(stderr, " %u", n[k]);
                                             fprintf(stderr, "\n");
                                         }
                                     });
    (void)hipFree(d_pencils); // (models_of_samples has synchronised the stream)
    return rc;
}

extern "C" int cvhip_ransac_affine_models(cvhip_device *dev, const uint32_t *matches, uint32_t N,
                                          const uint32_t *sample_idx, uint32_t B, double t, double *out_F)
{
    return models_of_samples(dev, matches, N, sample_idx, B, 4, 1, out_F, "ransac_affine_models",
                             [&](const uint4 *m4, const uint32_t *idx, double *d_F, hipStream_t s) {
                                 hipLaunchKernelGGL(ransac_generate_affine_kernel, dim3((B + 63) / 64), dim3(64), 0, s, m4, N, t,
                                                    0ull, 0u, B, idx, d_F);
                             });
}

extern "C" int cvhip_optimize_perspective_f(const double *F, const uint32_t *matches, uint32_t n, double *out_F,
                                            int *out_refined)
{
    if (!F || !out_F || !out_refined || (n && !matches)) return cvhip::fail(CVHIP_ERR_INVALID, "cvhip_optimize_perspective_f: null argument");
    try {
        std::vector<lm::Obs> obs;
        obs.reserve(n);
        for (uint32_t i = 0; i < n; i++) obs.push_back(lm::make_obs(matches[4 * (size_t)i], matches[4 * (size_t)i + 1], matches[4 * (size_t)i + 2], matches[4 * (size_t)i + 3]));
        std::vector<double> r(n), r_new(n), J((size_t)n * 7);
        double Fin[9], M[9];
        for (int k = 0; k < 9; k++) Fin[k] = F[k];
        const bool ok = lm::optimize_perspective_f(Fin, obs.data(), n, r.data(), r_new.data(), J.data(), M);
        *out_refined = ok ? 1 : 0;
        for (int k = 0; k < 9; k++) out_F[k] = ok ? M[k] : F[k]; // optimize_result: .unwrap_or(res.f), :246
        return CVHIP_OK;
    } catch (const std::bad_alloc &) {
        return cvhip::fail(CVHIP_ERR_NOMEM, "cvhip_optimize_perspective_f: out of host memory");
    }
}

// ---------------------------------------------------------------------------------------------------------
// optimize_result's refit (:246) on the device.  lm::levenberg_marquardt above is one thread's loop; on ~19 000
// inliers it costs the host 12 ms per pair (a third of config 5's RANSAC stage).  Here ONE workgroup runs the same
// statements with the work inside each statement spread over its 1024 threads:
//  * residuals and Jacobian rows are per observation (lm::residual_of / lm::gradient_of, unchanged);
//  * every long_dot is evaluated by eight threads, one per partial sum of blas `dot` (indices k, k + 8, ... in
//    order), then combined and given its tail by one thread exactly as long_dot does - the additions and their order
//    are the serial function's, so the values are too;
//  * the 7x7 system, norms, rho and the control flow are thread 0's (lm::solve7, same code), broadcast through LDS.
// The Jacobian is kept column by column (a chain's operands are then consecutive in memory: eight threads of a dot
// product share cache lines instead of touching eight).  J'J is recomputed only when J changed (a rejected step
// leaves J, hence J'J before the damping term, as it was), and
// the closing test's r.r is the `after` (accepted step) or `before` (rejected) that was just computed from the same
// vector.  tests/test_orb_ransac_gpu.py compares the result with the host function bit for bit.
// ---------------------------------------------------------------------------------------------------------
namespace refit {
constexpr int THREADS = 512; // (28 dots x 8 chains = 224 of them carry the long phases; 512 leave 256 registers a thread)
constexpr int MAX_DOTS = 49;
struct DotJob {
    const double *a, *b;
    uint32_t sa, sb;
};
struct Shared {
    double q[7], g[7], A0[49], step[7], trial[7];
    double parts[MAX_DOTS * 8], dotv[MAX_DOTS];
    DotJob jobs[MAX_DOTS];
    double mu, nu, rho;
    int action;
};
enum { CONTINUE = 0, DONE_TRUE = 1, DONE_FALSE = 2, ACCEPT = 3, ACCEPT_CONVERGED = 4, REJECT = 5 };

// sh.dotv[d] = lm::long_dot(jobs[d].a, sa, jobs[d].b, sb, n) for d < nd <= 7 (g = J'r, and the two r.r of every iteration).
// long_dot keeps eight partial sums (indices k, k + 8, ... in order): eight chains of dependent additions per dot, so only
// 16 .. 56 lanes carry this phase; one lane per chain fetching its own strided operands block after block waited 57 us
// per call for two dots of 19 000.  Here all threads load a tile of the factors with coalesced accesses - the NEXT tile's loads are
// in flight while this one is added - and leave the products in LDS; the chain lanes add them from there in long_dot's
// order.  prod: STAGE_DOUBLES doubles of LDS.
constexpr uint32_t STAGE_DOUBLES = 4096 + 64;
__device__ void dots_staged(Shared &sh, double *prod, int nd, uint32_t n)
{
    __syncthreads(); // jobs written, operands complete
    const uint32_t n8 = n & ~7u, tid = threadIdx.x;
    const uint32_t rows = (4096u / (uint32_t)nd) & ~63u, pitch = rows + 8u; // rows of a tile; +8: the dots' rows start in different banks
    const uint32_t per_tile = rows * (uint32_t)nd;
    constexpr uint32_t PER_THREAD = 4096 / THREADS;
    double fa[PER_THREAD], fb[PER_THREAD];
    uint32_t at[PER_THREAD]; // where the element's product goes (~0u: the thread has no element there)
    const auto load = [&](uint32_t base) {
#pragma unroll
        for (uint32_t u = 0; u < PER_THREAD; u++) {
            const uint32_t e = tid + THREADS * u, d = e / rows, k = e - d * rows, idx = base + k;
            at[u] = ~0u;
            if (e < per_tile && idx < n8) {
                const DotJob jb = sh.jobs[d];
                fa[u] = jb.a[(size_t)idx * jb.sa];
                fb[u] = jb.b[(size_t)idx * jb.sb];
                at[u] = d * pitch + k;
            }
        }
    };
    double part = 0.0;
    if (n8) load(0);
    for (uint32_t base = 0; base < n8; base += rows) {
#pragma unroll
        for (uint32_t u = 0; u < PER_THREAD; u++)
            if (at[u] != ~0u) prod[at[u]] = fa[u] * fb[u];
        __syncthreads();
        if (base + rows < n8) load(base + rows);
        if (tid < (uint32_t)nd * 8u) {
            const double *mine = prod + (tid >> 3) * pitch + (tid & 7u);
            const uint32_t steps = min(rows, n8 - base) >> 3;
            // (sixteen products at a time, the next sixteen on their way from LDS while these are added: one dependent
            // addition after the other is all the lane waits for)
            constexpr uint32_t BATCH = 16;
            double ba[BATCH], bb[BATCH]; // two batches, used in turn (no copies between them)
            const auto fetch = [&](double (&into)[BATCH], uint32_t from) {
#pragma unroll
                for (uint32_t k = 0; k < BATCH; k++) into[k] = mine[8u * (from + k)];
            };
            uint32_t u = 0;
            if (steps >= BATCH) fetch(ba, 0);
            while (u + BATCH <= steps) {
                if (u + 2u * BATCH <= steps) fetch(bb, u + BATCH);
#pragma unroll
                for (uint32_t k = 0; k < BATCH; k++) part += ba[k];
                u += BATCH;
                if (u + BATCH > steps) break;
                if (u + 2u * BATCH <= steps) fetch(ba, u + BATCH);
#pragma unroll
                for (uint32_t k = 0; k < BATCH; k++) part += bb[k];
                u += BATCH;
            }
            for (; u < steps; u++) part += mine[8u * u];
        }
        __syncthreads();
    }
    if (tid < (uint32_t)nd * 8u) sh.parts[tid] = part;
    __syncthreads();
    for (int d = (int)tid; d < nd; d += THREADS) {
        const DotJob jb = sh.jobs[d];
        double total = 0.0;
        for (uint32_t k = 0; k < 4; k++) total += sh.parts[d * 8 + k] + sh.parts[d * 8 + k + 4];
        for (uint32_t i = n8; i < n; i++) total += jb.a[(size_t)i * jb.sa] * jb.b[(size_t)i * jb.sb];
        sh.dotv[d] = total;
    }
    __syncthreads();
}
__device__ void evaluate(const double *at_lds, const uint4 *__restrict__ inl, uint32_t n, double *into)
{
    double at[7], M[9];
    for (int i = 0; i < 7; i++) at[i] = at_lds[i];
    lm::matrix_of(at, M);
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint4 m = inl[i];
        into[i] = lm::residual_of(M, lm::make_obs(m.x, m.y, m.z, m.w));
    }
}
// Jacobian at sh.q and g = J'res -> sh.g
__device__ void linearise(Shared &sh, double *stage, const uint4 *__restrict__ inl, uint32_t n, const double *res, double *J)
{
    double at[7], M[9];
    for (int i = 0; i < 7; i++) at[i] = sh.q[i];
    lm::matrix_of(at, M);
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint4 m = inl[i];
        double row[7];
        lm::gradient_of(M, lm::make_obs(m.x, m.y, m.z, m.w), row);
        for (int j = 0; j < 7; j++) J[(size_t)j * n + i] = row[j];
    }
    if (threadIdx.x < 7) sh.jobs[threadIdx.x] = DotJob{J + (size_t)threadIdx.x * n, res, 1u, 1u};
    dots_staged(sh, stage, 7, n);
    if (threadIdx.x < 7) sh.g[threadIdx.x] = sh.dotv[threadIdx.x];
    __syncthreads();
}
// J'J.  a[i] * b[i] == b[i] * a[i]: entry (j, i) is entry (i, j) bit for bit, so 28 of the 49 dots are evaluated - by 224
// chain lanes that would pull each of J's seven columns through one CU's L1 eight times over (75 us per call).
// Instead a tile of the seven columns is loaded ONCE into LDS by all threads (the next tile's loads in flight meanwhile)
// and the chain lanes form their products from there: long_dot's products, long_dot's additions, in its order.
__device__ void normal_matrix(Shared &sh, double *stage, uint32_t n, const double *J)
{
    constexpr uint32_t ROWS = THREADS, PITCH = ROWS + 8u; // one row of the tile per thread; 7 * PITCH <= STAGE_DOUBLES
    static_assert(7u * PITCH <= STAGE_DOUBLES, "the stage holds a tile of seven columns");
    const uint32_t n8 = n & ~7u, tid = threadIdx.x;
    int a = 0, b = 0;
    if (tid < 28u * 8u) { // pair number -> (a, b), a <= b
        int rem = (int)(tid >> 3);
        while (rem >= 7 - a) {
            rem -= 7 - a;
            a++;
        }
        b = a + rem;
    }
    __syncthreads(); // J complete; the stage is free
    double col[7], part = 0.0;
    const auto load = [&](uint32_t base) {
        if (base + tid < n8) {
#pragma unroll
            for (int c = 0; c < 7; c++) col[c] = J[(size_t)c * n + base + tid];
        }
    };
    if (n8) load(0);
    for (uint32_t base = 0; base < n8; base += ROWS) {
        if (base + tid < n8) {
#pragma unroll
            for (int c = 0; c < 7; c++) stage[c * PITCH + tid] = col[c];
        }
        __syncthreads();
        if (base + ROWS < n8) load(base + ROWS);
        if (tid < 28u * 8u) {
            const double *ca = stage + a * PITCH + (tid & 7u), *cb = stage + b * PITCH + (tid & 7u);
            const uint32_t steps = min(ROWS, n8 - base) >> 3;
            for (uint32_t u = 0; u < steps; u++) part += ca[8u * u] * cb[8u * u];
        }
        __syncthreads();
    }
    if (tid < 28u * 8u) sh.parts[tid] = part;
    __syncthreads();
    if (tid < 28u) {
        int pa = 0, rem = (int)tid;
        while (rem >= 7 - pa) {
            rem -= 7 - pa;
            pa++;
        }
        const int pb = pa + rem;
        double total = 0.0;
        for (uint32_t k = 0; k < 4; k++) total += sh.parts[tid * 8 + k] + sh.parts[tid * 8 + k + 4];
        for (uint32_t i = n8; i < n; i++) total += J[(size_t)pa * n + i] * J[(size_t)pb * n + i];
        sh.A0[pa * 7 + pb] = total;
        sh.A0[pb * 7 + pa] = total;
    }
    __syncthreads();
}
__device__ double largest7(const double *v)
{
    double m = v[0];
    for (int j = 1; j < 7; j++)
        if (m < v[j]) m = v[j];
    return m;
}
} // namespace refit

// inl: the n inliers; r, r_new [n], J [7 x n, column by column]: workspace; F_in: the winner (F[8] = 1); F_out / refined as
// cvhip_optimize_perspective_f
// n_dev (optional): the number of inliers where only the device knows it (the device loops' tail); J is then laid out
// for that n as well
__global__ __launch_bounds__(refit::THREADS) void ransac_refit_kernel(const uint4 *__restrict__ inl, uint32_t n, const uint32_t *__restrict__ n_dev,
                                                                       double *r, double *r_new, double *J, const double *F_in,
                                                                       double *F_out, int *refined)
{
    using namespace refit;
    if (n_dev) n = *n_dev;
    __shared__ Shared sh;
    __shared__ double stage[STAGE_DOUBLES];
    const bool lead = threadIdx.x == 0;
    if (threadIdx.x < 7) sh.q[threadIdx.x] = F_in[threadIdx.x]; // params_from_perspective_f, :429-440
    __syncthreads();
    evaluate(sh.q, inl, n, r);
    __syncthreads();
    linearise(sh, stage, inl, n, r, J);
    bool found = false, failed = false;
    if (fabs(largest7(sh.g)) <= 1e-12) found = true; // (every thread reads the same values)
    if (!found) {
        normal_matrix(sh, stage, n, J);
        if (lead) {
            double mu = 0.0;
            for (int j = 0; j < 7; j++) {
                const double djj = sh.A0[j * 7 + j];
                if (j == 0 || djj >= mu) mu = djj;
            }
            sh.mu = mu * 1e-3;
            sh.nu = 2.0;
        }
        __syncthreads();
        for (int iteration = 0; iteration < 1000 && !found && !failed; iteration++) {
            if (lead) {
                double A[49], step[7], q[7];
                for (int i = 0; i < 49; i++) A[i] = sh.A0[i];
                for (int i = 0; i < 7; i++) A[i * 7 + i] += sh.mu;
                for (int j = 0; j < 7; j++) {
                    step[j] = sh.g[j];
                    q[j] = sh.q[j];
                }
                int action = CONTINUE;
                if (!lm::solve7(A, step)) {
                    action = DONE_FALSE;
                } else if (sqrt(lm::long_dot(step, 1, step, 1, 7)) <= 1e-12 * (sqrt(lm::long_dot(q, 1, q, 1, 7)) + 1e-12)) {
                    action = DONE_TRUE;
                } else {
                    for (int j = 0; j < 7; j++) {
                        sh.step[j] = step[j];
                        sh.trial[j] = q[j] + step[j];
                    }
                }
                sh.action = action;
            }
            __syncthreads();
            if (sh.action == DONE_FALSE) {
                failed = true;
                break;
            }
            if (sh.action == DONE_TRUE) {
                found = true;
                break;
            }
            evaluate(sh.trial, inl, n, r_new);
            if (lead) {
                sh.jobs[0] = DotJob{r, r, 1u, 1u};
                sh.jobs[1] = DotJob{r_new, r_new, 1u, 1u};
            }
            dots_staged(sh, stage, 2, n);
            if (lead) {
                const double before = sh.dotv[0], after = sh.dotv[1];
                double step[7], damped[7];
                for (int j = 0; j < 7; j++) {
                    step[j] = sh.step[j];
                    damped[j] = step[j] * sh.mu + sh.g[j];
                }
                const double rho = (before - after) / lm::long_dot(step, 1, damped, 1, 7);
                sh.rho = rho;
                if (rho > 0.0) {
                    const bool converged = sqrt(before) - sqrt(after) < 0.0 * sqrt(before);
                    sh.action = converged ? ACCEPT_CONVERGED : ACCEPT;
                    for (int j = 0; j < 7; j++) sh.q[j] = sh.trial[j];
                } else {
                    sh.mu *= sh.nu;
                    sh.nu *= 2.0;
                    sh.action = REJECT;
                }
            }
            __syncthreads();
            const int action = sh.action;
            const double rr = action == REJECT ? sh.dotv[0] : sh.dotv[1]; // r.r of the vector r is NOW
            if (action != REJECT) {
                for (uint32_t i = threadIdx.x; i < n; i += THREADS) r[i] = r_new[i];
                __syncthreads();
                linearise(sh, stage, inl, n, r, J);
                if (action == ACCEPT_CONVERGED || fabs(largest7(sh.g)) <= 1e-12) {
                    found = true;
                    break;
                }
                normal_matrix(sh, stage, n, J); // J changed: J'J for the next iteration
                if (lead) {
                    const double w = 2.0 * sh.rho - 1.0, shrink = 1.0 - w * w * w;
                    sh.mu *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
                    sh.nu = 2.0;
                }
                __syncthreads();
            }
            if (sqrt(rr) <= 1e-12) found = true;
        }
    }
    if (lead) {
        bool ok = found && !failed;
        double M[9];
        if (ok) {
            double q[7];
            for (int i = 0; i < 7; i++) q[i] = sh.q[i];
            lm::matrix_of(q, M);
            const double Mt[9] = {M[0], M[3], M[6], M[1], M[4], M[7], M[2], M[5], M[8]};
            double sv[3];
            lm::singular3(Mt, sv);
            ok = !(fabs(sv[1]) < 1e-3 || fabs(sv[2]) > 1e-3); // :418-423
        }
        *refined = ok ? 1 : 0;
        for (int k = 0; k < 9; k++) F_out[k] = ok ? M[k] : F_in[k]; // optimize_result: .unwrap_or(res.f), :246
    }
}

extern "C" int cvhip_optimize_perspective_f_device(cvhip_device *dev, const double *F, const uint32_t *matches, uint32_t n,
                                                   double *out_F, int *out_refined)
{
    if (!dev || !F || !out_F || !out_refined || (n && !matches)) return fail(CVHIP_ERR_INVALID, "cvhip_optimize_perspective_f_device: null argument");
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    DevAllocs mem(dev->d);
    uint32_t *d_inl = nullptr;
    double *d_r = nullptr, *d_rn = nullptr, *d_J = nullptr, *d_F = nullptr;
    int *d_ref = nullptr;
    CVHIP_TRY_HIP(mem.alloc(&d_inl, (size_t)n * 4));
    CVHIP_TRY_HIP(mem.alloc(&d_r, n));
    CVHIP_TRY_HIP(mem.alloc(&d_rn, n));
    CVHIP_TRY_HIP(mem.alloc(&d_J, (size_t)n * 7));
    CVHIP_TRY_HIP(mem.alloc(&d_F, 18));
    CVHIP_TRY_HIP(mem.alloc(&d_ref, 1));
    if (n) CVHIP_TRY_HIP(hipMemcpyAsync(d_inl, matches, (size_t)n * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    CVHIP_TRY_HIP(hipMemcpyAsync(d_F, F, 9 * sizeof(double), dev_ptr(F) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ransac_refit_kernel, dim3(1), dim3(refit::THREADS), 0, s, reinterpret_cast<const uint4 *>(d_inl), n,
                       (const uint32_t *)nullptr, d_r, d_rn, d_J, (const double *)d_F, d_F + 9, d_ref);
    CVHIP_TRY_HIP(hipGetLastError());
    double h_F[9];
    int h_ref = 0;
    CVHIP_TRY_HIP(hipMemcpyAsync(h_F, d_F + 9, sizeof(h_F), hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipMemcpyAsync(&h_ref, d_ref, sizeof(int), hipMemcpyDeviceToHost, s));
    CVHIP_TRY_HIP(hipStreamSynchronize(s));
    std::memcpy(out_F, h_F, sizeof(h_F));
    *out_refined = h_ref;
    return CVHIP_OK;
}

namespace {
hipError_t launch_refit_tail(DevAllocs &mem, const uint4 *m4, uint32_t N, double t, uint8_t *d_mask, const double *d_F_in,
                             double *d_F_out, hipStream_t s)
{
    uint32_t *d_inl = nullptr, *d_n = nullptr;
    double *d_r = nullptr, *d_rn = nullptr, *d_J = nullptr;
    int *d_ref = nullptr;
    hipError_t e = mem.alloc(&d_inl, (size_t)N * 4);
    if (e == hipSuccess) e = mem.alloc(&d_n, 1);
    if (e == hipSuccess) e = mem.alloc(&d_r, N);
    if (e == hipSuccess) e = mem.alloc(&d_rn, N);
    if (e == hipSuccess) e = mem.alloc(&d_J, (size_t)N * 7);
    if (e == hipSuccess) e = mem.alloc(&d_ref, 1);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ransac_compact_inliers_kernel, dim3(1), dim3(1024), 0, s, (const uint8_t *)d_mask, m4, N, reinterpret_cast<uint4 *>(d_inl), d_n);
    hipLaunchKernelGGL(ransac_refit_kernel, dim3(1), dim3(refit::THREADS), 0, s, reinterpret_cast<const uint4 *>(d_inl), 0u,
                       (const uint32_t *)d_n, d_r, d_rn, d_J, d_F_in, d_F_out, d_ref);
    hipLaunchKernelGGL(ransac_refit_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const double *)d_F_out, m4, N, t, d_mask);
    return hipGetLastError();
}
} // namespace

// fits_model (:452-458) of one F for every match: the inlier filter of optimize_result (:233-236, 248-254).
extern "C" int cvhip_fits_model(cvhip_device *dev, const double *F, const uint32_t *matches, uint32_t N, double t,
                                uint8_t *out_mask)
{
    if (!dev || !F || (!matches && N) || (!out_mask && N)) return fail(CVHIP_ERR_INVALID, "null argument");
    if (N == 0) return CVHIP_OK;
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    hipStream_t s = dev->d.stream;
    uint32_t *d_m = nullptr;
    RansacBest *d_best = nullptr;
    uint8_t *d_mask = nullptr;
    RansacBest h_best;
    std::memset(&h_best, 0, sizeof(h_best));
    for (int i = 0; i < 9; i++) h_best.f[i] = F[i];
    hipError_t e = hipMalloc(&d_m, (size_t)N * 16);
    if (e == hipSuccess) e = hipMalloc(&d_best, sizeof(RansacBest));
    if (e == hipSuccess) e = hipMalloc(&d_mask, N);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, matches, (size_t)N * 16, dev_ptr(matches) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_best, &h_best, sizeof(RansacBest), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ransac_inlier_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_best,
                           reinterpret_cast<const uint4 *>(d_m), N, t, d_mask);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_mask, d_mask, N, dev_ptr(out_mask) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_m);
    (void)hipFree(d_best);
    (void)hipFree(d_mask);
    if (e != hipSuccess) return fail(CVHIP_ERR_DEVICE, std::string("fits_model: ") + hipGetErrorString(e));
    return CVHIP_OK;
}

// FundamentalMatrix::new(projection, max_dimension).find_ransac(matches) as one call (fundamentalmatrix.rs:72-147
// + optimize_result :231-257): the device RANSAC of the model, then for the perspective model the reference's LM
// refit of the winner on its inliers and the re-selection of the inliers with the refitted matrix.
extern "C" int cvhip_find_ransac(cvhip_device *dev, int projection, const uint32_t *matches, uint32_t N, double max_dimension,
                                 uint64_t seed, double *out_F, uint32_t *out_inlier_count, uint8_t *out_inlier_mask,
                                 cvhip_progress_fn progress, cvhip_matches_fn report_matches, void *user)
{
    if (projection != 0 && projection != 1) return fail(CVHIP_ERR_INVALID, "projection must be 0 or 1");
    struct ListenerScope { // the listener is this call's: installed for its duration on this thread
        ListenerScope(cvhip_progress_fn p, cvhip_matches_fn m, void *u)
        {
            g_listener = RansacListener{};
            g_listener.progress = p;
            g_listener.matches = m;
            g_listener.user = u;
        }
        ~ListenerScope() { g_listener = RansacListener{}; }
    } scope(progress, report_matches, user);
    if (projection == 0) return cvhip_ransac_affine(dev, matches, N, seed, out_F, out_inlier_count, out_inlier_mask);
    if (!dev || !matches || !out_F) return fail(CVHIP_ERR_INVALID, "null argument");
    try {
        // the RANSAC rounds, the winner's inliers, optimize_result's refit on them (:246) and the inliers of the refitted
        // matrix (:248-254; of the winner itself where the refit returns None): one enqueue, one read-back
        return ransac_perspective(dev, matches, N, max_dimension, seed, 0, out_F, out_inlier_count, out_inlier_mask, true);
    } catch (const std::bad_alloc &) {
        return cvhip::fail(CVHIP_ERR_NOMEM, "cvhip_find_ransac: out of host memory");
    }
}
