/*
 * rr_oracle.c -- CPU restatement of river-route's Muskingum hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under river_route_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle.py
 * against golden vectors produced in the build container by running the
 * reference's own Python (tests/golden/make_golden.py) and against the
 * known-answer cases the reference's tests hold (tests/test_uhkernels.py:52-99,
 * tests/test_tools.py:48-60).
 *
 * Each function follows the reference statement order (column-oriented CSC scatter
 * + forward substitution), single thread, fp64, int32 indices -- not the gather
 * form the GPU path uses -- so the two are independent derivations of the same math.
 *
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_ALLOC -1
#define ORC_E_COEFF_SUM -2
#define ORC_E_UNKNOWN_DOWNSTREAM -3
#define ORC_E_NOT_TOPOLOGICAL -4

/* One routing sub-step shared by the three routers: on entry rhs holds the
 * diagonal/lateral part; adds c2*(A q) by CSC scatter, then forward-substitutes
 * (I - diag(c1) A) q+ = rhs in column order.
 * river_route/routers/_numba_kernels.py:29-39 (muskingum), 70-78 (rapid), 152-162 (unit). */
static void substep_scatter_solve(int64_t n, const int32_t *indptr, const int32_t *indices,
                                  const double *lhs_off, const double *c2,
                                  const double *q_src, double *q_dst, double *rhs)
{
    for (int64_t col = 0; col < n; ++col) {
        const double qv = q_src[col];
        for (int32_t j = indptr[col]; j < indptr[col + 1]; ++j) {
            const int32_t row = indices[j];
            rhs[row] += c2[row] * qv;
        }
    }
    for (int64_t col = 0; col < n; ++col) {
        q_dst[col] = rhs[col];
        for (int32_t j = indptr[col]; j < indptr[col + 1]; ++j)
            rhs[indices[j]] -= lhs_off[j] * q_dst[col];
    }
}

/* river_route/routers/_numba_kernels.py:8-46 */
int orc_muskingum_route(int64_t n, const int32_t *indptr, const int32_t *indices,
                        const double *lhs_off, const double *c2, const double *c3,
                        double *q_t, double *discharge,
                        int64_t num_output_steps, int64_t num_routing_per_output)
{
    double *rhs = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *isum = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!rhs || !isum) { free(rhs); free(isum); return ORC_E_ALLOC; }
    const double inv = 1.0 / (double)num_routing_per_output;
    for (int64_t o = 0; o < num_output_steps; ++o) {
        for (int64_t i = 0; i < n; ++i) isum[i] = 0.0;
        for (int64_t s = 0; s < num_routing_per_output; ++s) {
            for (int64_t i = 0; i < n; ++i) rhs[i] = c3[i] * q_t[i];
            substep_scatter_solve(n, indptr, indices, lhs_off, c2, q_t, q_t, rhs);
            for (int64_t i = 0; i < n; ++i) isum[i] += q_t[i];
        }
        double *row = discharge + o * n;
        for (int64_t i = 0; i < n; ++i) {
            const double v = isum[i] * inv;
            row[i] = v > 0.0 ? v : 0.0;
        }
    }
    free(rhs); free(isum);
    return ORC_OK;
}

/* river_route/routers/_numba_kernels.py:49-84.  qlateral is (T, n) C-order. */
int orc_rapid_route(int64_t n, const int32_t *indptr, const int32_t *indices,
                    const double *lhs_off, const double *c2, const double *c3,
                    const double *c4_dt, double *q_t, const double *qlateral,
                    double *discharge, int64_t num_runoff_steps, int64_t num_substeps)
{
    double *rhs = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *isum = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!rhs || !isum) { free(rhs); free(isum); return ORC_E_ALLOC; }
    const double inv = 1.0 / (double)num_substeps;
    for (int64_t t = 0; t < num_runoff_steps; ++t) {
        const double *ql = qlateral + t * n;
        for (int64_t i = 0; i < n; ++i) isum[i] = 0.0;
        for (int64_t s = 0; s < num_substeps; ++s) {
            for (int64_t i = 0; i < n; ++i) rhs[i] = c3[i] * q_t[i] + c4_dt[i] * ql[i];
            substep_scatter_solve(n, indptr, indices, lhs_off, c2, q_t, q_t, rhs);
            for (int64_t i = 0; i < n; ++i) isum[i] += q_t[i];
        }
        double *row = discharge + t * n;
        for (int64_t i = 0; i < n; ++i) {
            const double v = isum[i] * inv;
            row[i] = v > 0.0 ? v : 0.0;
        }
    }
    free(rhs); free(isum);
    return ORC_OK;
}

/* y = M x for a CSC matrix with explicit data (river_route/routers/_numba_kernels.py:126-139) */
static void csc_spmv(int64_t nrows, int64_t ncols, const int32_t *indptr, const int32_t *indices,
                     const double *data, const double *x, double *y)
{
    for (int64_t i = 0; i < nrows; ++i) y[i] = 0.0;
    for (int64_t col = 0; col < ncols; ++col) {
        const double v = x[col];
        for (int32_t j = indptr[col]; j < indptr[col + 1]; ++j)
            y[indices[j]] += data[j] * v;
    }
}

/* river_route/routers/_numba_kernels.py:88-171.  convolved_lateral and discharge are
 * (T, n_total) C-order; hw_idx / inner_idx are int64 positions into the full index space. */
int orc_unit_route(int64_t n_inner, int64_t n_hw, int64_t n_total,
                   const int32_t *lhs_indptr, const int32_t *lhs_indices, const double *lhs_off,
                   const int32_t *a_in_indptr, const int32_t *a_in_indices, const double *a_in_data,
                   const int32_t *a_hw_indptr, const int32_t *a_hw_indices, const double *a_hw_data,
                   const double *c1_in, const double *c2_in, const double *c3_in,
                   const int64_t *hw_idx, const int64_t *inner_idx,
                   double *q_ch, double *q_full,
                   const double *convolved, double *discharge,
                   int64_t num_runoff_steps, int64_t num_substeps)
{
    const size_t ni = (size_t)(n_inner ? n_inner : 1), nh = (size_t)(n_hw ? n_hw : 1);
    double *buf = (double *)malloc(sizeof(double) * (6 * ni + nh));
    if (!buf) return ORC_E_ALLOC;
    double *rhs = buf, *isum = buf + ni, *ql_in = buf + 2 * ni, *a_in_res = buf + 3 * ni,
           *a_hw_res = buf + 4 * ni, *c1_a_ql = buf + 5 * ni, *ql_hw = buf + 6 * ni;
    const double inv = 1.0 / (double)num_substeps;
    for (int64_t t = 0; t < num_runoff_steps; ++t) {
        const double *lat = convolved + t * n_total;
        double *row = discharge + t * n_total;
        for (int64_t i = 0; i < n_hw; ++i) ql_hw[i] = lat[hw_idx[i]];          /* 116-117 */
        for (int64_t i = 0; i < n_inner; ++i) ql_in[i] = lat[inner_idx[i]];    /* 118-119 */
        for (int64_t i = 0; i < n_hw; ++i) row[hw_idx[i]] = ql_hw[i];          /* 122-123: no clamp, no mean */
        csc_spmv(n_inner, n_inner, a_in_indptr, a_in_indices, a_in_data, ql_in, a_in_res);
        csc_spmv(n_inner, n_hw, a_hw_indptr, a_hw_indices, a_hw_data, ql_hw, a_hw_res);
        for (int64_t i = 0; i < n_inner; ++i) c1_a_ql[i] = c1_in[i] * (a_in_res[i] + a_hw_res[i]);
        for (int64_t i = 0; i < n_inner; ++i) isum[i] = 0.0;
        for (int64_t s = 0; s < num_substeps; ++s) {
            for (int64_t i = 0; i < n_inner; ++i)
                rhs[i] = c1_a_ql[i] + c2_in[i] * a_hw_res[i] + c3_in[i] * q_ch[i];   /* 150-151 */
            substep_scatter_solve(n_inner, lhs_indptr, lhs_indices, lhs_off, c2_in, q_full, q_ch, rhs);
            for (int64_t i = 0; i < n_inner; ++i) {                                   /* 165-167 */
                q_full[i] = q_ch[i] + ql_in[i];
                isum[i] += q_full[i];
            }
        }
        for (int64_t i = 0; i < n_inner; ++i) {                                       /* 169-171 */
            const double v = isum[i] * inv;
            row[inner_idx[i]] = v > 0.0 ? v : 0.0;
        }
    }
    free(buf);
    return ORC_OK;
}

/* river_route/routers/Muskingum.py:172-185.  Returns ORC_E_COEFF_SUM where the reference
 * raises ValueError (numpy.allclose(c1+c2+c3, 1): |s-1| <= 1e-8 + 1e-5*1, NaN fails). */
int orc_muskingum_coefficients(int64_t n, const double *k, const double *x, double dt_routing,
                               double *c1, double *c2, double *c3)
{
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double r = dt_routing / k[i];
        const double den = r + (2.0 * (1.0 - x[i]));
        const double twox = 2.0 * x[i];
        c1[i] = (r - twox) / den;
        c2[i] = (r + twox) / den;
        c3[i] = ((2.0 * (1.0 - x[i])) - r) / den;
        const double s = c1[i] + c2[i] + c3[i];
        if (!(fabs(s - 1.0) <= 1e-8 + 1e-5)) bad = 1;
    }
    return bad ? ORC_E_COEFF_SUM : ORC_OK;
}

static int cmp_i64_pair(const void *a, const void *b)
{
    const int64_t x = ((const int64_t *)a)[0], y = ((const int64_t *)b)[0];
    return (x > y) - (x < y);
}

/* river_route/tools.py:75-109.  Fills CSC (indptr[n+1], indices[<=n]) of A[down, up] = 1; every
 * column holds at most one entry.  Same rejections, in the same scan order, as the reference. */
int orc_adjacency_matrix(int64_t n, const int64_t *river_ids, const int64_t *downstream_ids,
                         int32_t *indptr, int32_t *indices)
{
    int64_t *tab = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(n ? n : 1));
    if (!tab) return ORC_E_ALLOC;
    for (int64_t i = 0; i < n; ++i) { tab[2 * i] = river_ids[i]; tab[2 * i + 1] = i; }
    qsort(tab, (size_t)n, 2 * sizeof(int64_t), cmp_i64_pair);
    int32_t nnz = 0;
    indptr[0] = 0;
    for (int64_t up = 0; up < n; ++up) {
        const int64_t d = downstream_ids[up];
        if (d >= 0) {
            int64_t lo = 0, hi = n - 1, pos = -1;
            while (lo <= hi) {
                const int64_t mid = (lo + hi) / 2;
                if (tab[2 * mid] == d) { pos = mid; break; }
                if (tab[2 * mid] < d) lo = mid + 1; else hi = mid - 1;
            }
            if (pos < 0) { free(tab); return ORC_E_UNKNOWN_DOWNSTREAM; }
            /* duplicates: the reference's dict keeps the LAST index of a repeated id */
            while (pos + 1 < n && tab[2 * (pos + 1)] == d) ++pos;
            const int64_t down = tab[2 * pos + 1];
            if (down <= up) { free(tab); return ORC_E_NOT_TOPOLOGICAL; }
            indices[nnz++] = (int32_t)down;
        }
        indptr[up + 1] = nnz;
    }
    free(tab);
    return ORC_OK;
}

/* river_route/uhkernels/UnitHydrograph.py:64-75 -- the definitional direct form, one step. */
void orc_uh_convolve_incrementally(int64_t n_ks, int64_t n, const double *kernel, double *state,
                                   const double *runoff, double *out)
{
    for (int64_t s = 0; s < n_ks; ++s)
        for (int64_t i = 0; i < n; ++i) state[s * n + i] += kernel[s * n + i] * runoff[i];
    for (int64_t i = 0; i < n; ++i) out[i] = state[i];
    for (int64_t s = 0; s + 1 < n_ks; ++s)
        memcpy(state + s * n, state + (s + 1) * n, sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) state[(n_ks - 1) * n + i] = 0.0;
}

/* river_route/uhkernels/UnitHydrograph.py:77-107.  scipy.signal.fftconvolve(lateral, kernel,
 * axes=0, mode='full') is, by its published definition, the linear convolution
 * buf[m, i] = sum_s kernel[s, i] * lateral[m - s, i]; restated here in direct form
 * (agrees with the FFT evaluation to rounding, ~1e-13 relative).  state is (n_ks, n) in/out. */
int orc_uh_convolve(int64_t T, int64_t n_ks, int64_t n, const double *kernel, double *state,
                    const double *lateral, double *out)
{
    const int64_t M = T + n_ks - 1;
    double *buf = (double *)calloc((size_t)(M > 0 ? M : 1) * (size_t)(n ? n : 1), sizeof(double));
    if (!buf) return ORC_E_ALLOC;
    for (int64_t m = 0; m < M; ++m) {
        const int64_t s_lo = m - (T - 1) > 0 ? m - (T - 1) : 0;
        const int64_t s_hi = m < n_ks - 1 ? m : n_ks - 1;
        double *b = buf + m * n;
        for (int64_t s = s_lo; s <= s_hi; ++s) {
            const double *kr = kernel + s * n, *lr = lateral + (m - s) * n;
            for (int64_t i = 0; i < n; ++i) b[i] += kr[i] * lr[i];
        }
    }
    /* buf[:n_ks] += state  (line 100); buf has M = T+n_ks-1 >= n_ks rows whenever T >= 1 */
    for (int64_t s = 0; s < n_ks && s < M; ++s)
        for (int64_t i = 0; i < n; ++i) buf[s * n + i] += state[s * n + i];
    /* state[:] = 0; state[:n_ks-1] = buf[T:]  (lines 103-105) */
    memset(state, 0, sizeof(double) * (size_t)n_ks * (size_t)n);
    for (int64_t s = 0; s + 1 < n_ks; ++s)
        memcpy(state + s * n, buf + (T + s) * n, sizeof(double) * (size_t)n);
    memcpy(out, buf, sizeof(double) * (size_t)T * (size_t)n);
    free(buf);
    return ORC_OK;
}
