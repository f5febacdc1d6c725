/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this.  The product path
 * (dbgsom_amd/) never links or calls it.
 *
 * CPU restatement, in plain C, of the arithmetic of the reference's BMU search
 *   dbgsom/BaseSom.py:446-464  _get_winning_neurons  ->  sklearn NearestNeighbors.kneighbors
 * whose brute engine (scikit-learn 1.7.2, a third-party dependency, requirements.txt:4
 * "scikit-learn>=1.2", unpinned) evaluates, in float64 even for float32 input,
 *
 *     r_ij = (|x_i|^2 + (-2 * <x_i, w_j>)) + |w_j|^2 ;  r = max(r, 0) ;
 *     winner_i = argmin_j r_ij  (ties -> lowest j) ;  dist_i = sqrt(r)
 *
 *   (sklearn/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:471-510 for f64/f64 and
 *    f32/f32 inputs; sklearn/metrics/pairwise.py:424-431 + neighbors/_base.py:749-758 for the
 *    f32-X / f64-W case the reference hits from epoch 1 on).
 *
 * BLAS leaves the summation order of <x_i, w_j> unspecified; this restatement FIXES it to
 * the sequential chain  acc = fma(x_k, w_k, acc), k = 0..d-1, acc0 = +0  (likewise for the two
 * squared norms).  That is the order the gfx950 f64 MFMA path produces, so winners AND
 * distances can be compared bit for bit.  tests/test_oracle_golden.py pins this file against
 * the golden vectors captured from the reference itself (tests/golden/).
 *
 * Second half: the reference's two numba kernels restated serially
 *   numba_voronoi_set_centers  BaseSom.py:1028-1055   (weighted sums, id-indexed; the
 *                                                       compaction quirk Q1 is applied by the caller)
 *   numba_quantization_error   BaseSom.py:1058-1073   (serial semantics)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JB 32 /* prototypes per register block (vectorised over j) */

static inline double sq_chain_f64(const double *a, int64_t d) {
    double acc = 0.0;
    for (int64_t k = 0; k < d; ++k) acc = __builtin_fma(a[k], a[k], acc);
    return acc;
}

/* squared row norms, sequential fma chain; dtype: 0 = f32 rows (up-cast exactly), 1 = f64 */
void oracle_row_sqnorms(const void *A, int dtype, int64_t rows, int64_t d, int64_t ld,
                        double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i) {
        double acc = 0.0;
        if (dtype == 0) {
            const float *a = (const float *)A + i * ld;
            for (int64_t k = 0; k < d; ++k) {
                double v = (double)a[k];
                acc = __builtin_fma(v, v, acc);
            }
        } else {
            acc = sq_chain_f64((const double *)A + i * ld, d);
        }
        out[i] = acc;
    }
}

/*
 * BMU search, k in {1,2}.  X: N x d (ld = row stride in elements), dtype 0=f32 1=f64.
 * W: M x d float64, contiguous.  idx: N x k int64, dist: N x k float64 (sqrt applied),
 * sorted by (r, j) lexicographically.  Returns 0, or -1 on bad arguments.
 */
int oracle_bmu_chain(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                     const double *W, int64_t M, int k, int64_t *idx, double *dist) {
    if (k < 1 || k > 2 || M < k || d < 1 || N < 0) return -1;
    const int64_t Mp = (M + JB - 1) / JB * JB;
    /* W transposed and padded: Wt[kk][j] so the j loop is contiguous */
    double *Wt = (double *)calloc((size_t)d * (size_t)Mp, sizeof(double));
    double *yy = (double *)malloc((size_t)M * sizeof(double));
    if (!Wt || !yy) { free(Wt); free(yy); return -1; }
    for (int64_t j = 0; j < M; ++j) {
        yy[j] = sq_chain_f64(W + j * d, d);
        for (int64_t kk = 0; kk < d; ++kk) Wt[kk * Mp + j] = W[j * d + kk];
    }
#pragma omp parallel
    {
        double *xrow = (double *)malloc((size_t)d * sizeof(double));
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < N; ++i) {
            if (x_dtype == 0) {
                const float *x = (const float *)X + i * ldx;
                for (int64_t kk = 0; kk < d; ++kk) xrow[kk] = (double)x[kk];
            } else {
                memcpy(xrow, (const double *)X + i * ldx, (size_t)d * sizeof(double));
            }
            const double xx = sq_chain_f64(xrow, d);
            double b0 = INFINITY, b1 = INFINITY;
            int64_t j0 = -1, j1 = -1;
            for (int64_t jb = 0; jb < Mp; jb += JB) {
                double acc[JB];
                for (int t = 0; t < JB; ++t) acc[t] = 0.0;
                for (int64_t kk = 0; kk < d; ++kk) {
                    const double xv = xrow[kk];
                    const double *w = Wt + kk * Mp + jb;
                    for (int t = 0; t < JB; ++t) acc[t] = __builtin_fma(xv, w[t], acc[t]);
                }
                for (int t = 0; t < JB; ++t) {
                    const int64_t j = jb + t;
                    if (j >= M) break;
                    double r = (xx + (-2.0 * acc[t])) + yy[j];
                    if (!(r > 0.0)) r = (r != r) ? r : 0.0; /* max(r, 0); NaN propagates */
                    /* strict '<' keeps the lowest j on ties (j ascends) */
                    if (r < b0) { b1 = b0; j1 = j0; b0 = r; j0 = j; }
                    else if (r < b1) { b1 = r; j1 = j; }
                }
            }
            idx[i * k] = j0;
            dist[i * k] = sqrt(b0);
            if (k == 2) { idx[i * k + 1] = j1; dist[i * k + 1] = sqrt(b1); }
        }
        free(xrow);
    }
    free(Wt);
    free(yy);
    return 0;
}

/*
 * Per-neuron statistics of one epoch (serial, sample order):
 *   S[j,:] = sum_{i: win_i = j} kw_i * x_i     (numerator of the weighted Voronoi centre)
 *   K[j]   = sum kw_i ;  a[j] = |{i}| ;  E[j] = sum dist_i
 * S is M x d, zero-initialised here.
 */
int oracle_accumulate(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                      const int64_t *win, const double *kw, const double *dist, int64_t M,
                      double *S, double *K, double *a, double *E) {
    memset(S, 0, (size_t)M * (size_t)d * sizeof(double));
    memset(K, 0, (size_t)M * sizeof(double));
    memset(a, 0, (size_t)M * sizeof(double));
    memset(E, 0, (size_t)M * sizeof(double));
    for (int64_t i = 0; i < N; ++i) {
        const int64_t j = win[i];
        if (j < 0 || j >= M) return -1;
        double *s = S + j * d;
        const double w = kw[i];
        if (x_dtype == 0) {
            const float *x = (const float *)X + i * ldx;
            for (int64_t kk = 0; kk < d; ++kk) s[kk] += w * (double)x[kk];
        } else {
            const double *x = (const double *)X + i * ldx;
            for (int64_t kk = 0; kk < d; ++kk) s[kk] += w * x[kk];
        }
        K[j] += w;
        a[j] += 1.0;
        E[j] += dist[i];
    }
    return 0;
}
