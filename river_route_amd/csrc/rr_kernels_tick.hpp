// rr_kernels_tick.hpp -- the streaming kernels: one routing tick per launch (k_tick, k_tick_unit), the two-phase row permutation that feeds them, state scatter / gather.
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

struct TickArgs {
    const int32_t *child_ptr;  // [n+1]
    const int32_t *lag;        // [n]
    const double *w;           // [n] c1 of the downstream reach, stored at the UPSTREAM position
    const double *c1row;       // [n] the same as ONE weight per reach, or NULL when the weights into a reach differ
    const double *c2, *c3, *c4;
    const double *xa;          // values written one tick ago
    const double *xb;          // values written two ticks ago
    double *xc;                // this tick's values
    double *isum;              // running sum over the sub-steps of one output row (nsub > 1 only)
    const int32_t *bidx;       // [n] ghost / export slot, read by flagged lanes only
    const double *ghost;       // [total_substeps, n_ghost] prescribed series
    double *exports;           // [total_substeps, n_export] recorded series
    int32_t n_ghost, n_export;
    const double *in;          // lateral rows, engine order (NULL for channel-only)
    double *out;               // discharge rows, engine order
    int64_t in_ld, out_ld;
    Div32 in_rows, out_rows;
    int32_t p_lo, p_hi;        // active engine positions
    int64_t tau;               // tick
    int64_t total_substeps;    // T * nsub
    Div32 nsub;
    double inv_nsub;
};

// One routing tick for Muskingum / RapidMuskingum.  One reach per lane; positions are lag-ordered so a
// wave reads contiguous spans of every array, including the upstream values (rr_plan.hpp).
template <bool HAS_LATERAL, bool SINGLE_SUBSTEP>
__global__ __launch_bounds__(kBlock) void k_tick(const TickArgs a)
{
    const int32_t p = a.p_lo + (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= a.p_hi) return;
    const int32_t lag_bits = a.lag[p];
    const int32_t ts = (int32_t)a.tau - (lag_bits & kLagMask);
    if (ts < 0 || ts >= (int32_t)a.total_substeps) return;
    if (lag_bits & kGhostBit) {   // boundary inflow: the value another GPU computed for this sub-step
        a.xc[p] = a.ghost[(int64_t)ts * a.n_ghost + a.bidx[p]];
        return;
    }
    uint32_t t, s;
    if (SINGLE_SUBSTEP) { t = (uint32_t)ts; s = 0; }
    else t = a.nsub.div((uint32_t)ts, s);

    const int32_t u0 = a.child_ptr[p], u1 = a.child_ptr[p + 1];
    double r;
    if (a.c1row) {
        // one upstream weight per reach (what the reference's callers produce): the arithmetic of k_tile, operation for
        // operation, so a call routed here and one routed there agree bit for bit (split run == joint run)
        double s_new = 0.0, s_old = 0.0;
        for (int32_t u = u0; u < u1; ++u) { s_new += a.xa[u]; s_old += a.xb[u]; }
        const double lat = HAS_LATERAL ? a.c4[p] * a.in[(int64_t)a.in_rows.mod(t) * a.in_ld + p] : 0.0;
        r = __builtin_fma(a.c1row[p], s_new, __builtin_fma(a.c2[p], s_old, __builtin_fma(a.c3[p], a.xa[p], lat)));
    } else {
        r = a.c3[p] * a.xa[p];
        if (HAS_LATERAL) r += a.c4[p] * a.in[(int64_t)a.in_rows.mod(t) * a.in_ld + p];
        const double c2 = a.c2[p];
        for (int32_t u = u0; u < u1; ++u) r += c2 * a.xb[u];
        for (int32_t u = u0; u < u1; ++u) r += a.w[u] * a.xa[u];
    }
    if (lag_bits & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[p]] = r;
    a.xc[p] = r;

    if (SINGLE_SUBSTEP) {
        a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = r > 0.0 ? r : 0.0;
    } else {
        const double acc = (s == 0 ? 0.0 : a.isum[p]) + r;
        if (s + 1 == a.nsub.d) {
            const double v = acc * a.inv_nsub;
            a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = v > 0.0 ? v : 0.0;
        } else {
            a.isum[p] = acc;
        }
    }
}

struct UnitTickArgs {
    TickArgs t;
    const uint16_t *hw_children;  // [n] count of headwater tributaries (stored first among the upstream range)
    double *qch;                  // [n] channel-only discharge of inner reaches, updated in place
    // general edge data (rr_plan_set_unit_weights; never produced by the reference's callers, whose A is all ones and whose
    // lhs_off_data is -c1[row]): a2[u] = a_inner_data / a_hw_data of the edge leaving u, c1own[p] = c1 of the reach itself
    const double *a2, *c1own;
    double *zc;                   // general edge data: this tick's q_ch+ of every inner reach ...
    const double *za;             // ... and last tick's (the work rows hold a reach's discharge where its lateral inflow was: in place)
};

// One routing tick for UnitMuskingum (river_route/routers/_numba_kernels.py:113-171 in gather form).
// A headwater publishes its convolved lateral l_t as both its "old" and "new" discharge; an inner reach
// routes q_ch and publishes q_full = q_ch + l_t.
template <bool SINGLE_SUBSTEP>
__global__ __launch_bounds__(kBlock) void k_tick_unit(const UnitTickArgs ua)
{
    const TickArgs &a = ua.t;
    const int32_t p = a.p_lo + (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= a.p_hi) return;
    const int32_t lag_bits = a.lag[p];
    const int32_t ts = (int32_t)a.tau - (lag_bits & kLagMask);
    if (ts < 0 || ts >= (int32_t)a.total_substeps) return;
    if (lag_bits & kGhostBit) {   // boundary inflow: the discharge another GPU published for this sub-step
        a.xc[p] = a.ghost[(int64_t)ts * a.n_ghost + a.bidx[p]];
        return;
    }
    uint32_t t, s;
    if (SINGLE_SUBSTEP) { t = (uint32_t)ts; s = 0; }
    else t = a.nsub.div((uint32_t)ts, s);

    const double lat = a.in[(int64_t)a.in_rows.mod(t) * a.in_ld + p];
    const int32_t u0 = a.child_ptr[p], u1 = a.child_ptr[p + 1];
    if (u0 == u1) {  // headwater: discharge is the lateral inflow, unclamped and un-averaged (lines 122-123)
        a.xc[p] = lat;
        if (lag_bits & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[p]] = lat;
        if (s == 0) a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = lat;
        return;
    }
    const int32_t uh = u0 + (int32_t)ua.hw_children[p];
    double r;
    if (a.c1row) {      // the arithmetic of k_tile, operation for operation (see k_tick)
        double s_hw = 0.0, s_new = 0.0, s_old = 0.0;
        for (int32_t u = u0; u < uh; ++u) s_hw += a.xa[u];   // headwater tributaries: "old" value is l_t too
        for (int32_t u = uh; u < u1; ++u) { s_new += a.xa[u]; s_old += a.xb[u]; }
        r = __builtin_fma(a.c1row[p], s_hw + s_new, __builtin_fma(a.c2[p], s_hw + s_old, a.c3[p] * ua.qch[p]));
    } else if (ua.a2) {
        // _numba_kernels.py:126-162 term by term: c1 (A_in l_in + A_hw l_hw) + c2 A_hw l_hw + c3 q_ch + c2 (structure of lhs) q_full
        // - lhs_off q_ch+, with the upstream reach's own lateral (same row, its position) separating q_ch+ from what it published
        r = a.c3[p] * ua.qch[p];
        const double c1 = ua.c1own[p], c2 = a.c2[p];
        for (int32_t u = u0; u < uh; ++u) r += (c1 + c2) * (ua.a2[u] * a.xa[u]);
        for (int32_t u = uh; u < u1; ++u) {
            const double qu = ua.za[u];      // q_ch+ of the upstream reach; what it published is q_ch+ + its lateral inflow
            r += c2 * a.xb[u] + c1 * (ua.a2[u] * (a.xa[u] - qu)) + a.w[u] * qu;
        }
        ua.zc[p] = r;
    } else {
        r = a.c3[p] * ua.qch[p];
        const double c2 = a.c2[p];
        for (int32_t u = u0; u < uh; ++u) r += c2 * a.xa[u];
        for (int32_t u = uh; u < u1; ++u) r += c2 * a.xb[u];
        for (int32_t u = u0; u < u1; ++u) r += a.w[u] * a.xa[u];
    }
    ua.qch[p] = r;
    const double qfull = r + lat;
    a.xc[p] = qfull;
    if (lag_bits & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[p]] = qfull;

    if (SINGLE_SUBSTEP) {
        a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = qfull > 0.0 ? qfull : 0.0;
    } else {
        const double acc = (s == 0 ? 0.0 : a.isum[p]) + qfull;
        if (s + 1 == a.nsub.d) {
            const double v = acc * a.inv_nsub;
            a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = v > 0.0 ? v : 0.0;
        } else {
            a.isum[p] = acc;
        }
    }
}

// ---- two-phase tiled permutation of (time, reach) rows, rr_plan.hpp / DESIGN.md section 4 ----
constexpr int kPermThreads = 1024;
constexpr int kPermE = 8;   // elements per thread: 8192-element (64 KiB) tiles


// Phase A: source tile -> LDS (sorted by destination tile) -> runs of the intermediate rows M[r, :].
template <int E>
__global__ __launch_bounds__(kPermThreads) void k_perm_a(const RowView src, double *__restrict__ m_rows, int64_t n,
                                                         const uint16_t *__restrict__ slot_a,
                                                         const int32_t *__restrict__ m_index, int64_t t0,
                                                         int32_t nrows, int32_t rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t base = (int64_t)blockIdx.x * (E * kPermThreads);
    const int tid = threadIdx.x;
    int32_t slot[E], mi[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int64_t i = base + k * kPermThreads + tid;
        slot[k] = i < n ? (int32_t)slot_a[i] : -1;
        mi[k] = i < n ? m_index[i] : -1;
    }
    const int32_t r0 = (int32_t)blockIdx.y * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    for (int32_t r = r0; r < r1; ++r) {
        const double *srow = src.row(t0 + r) + base;
        double v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = slot[k] >= 0 ? srow[k * kPermThreads + tid] : 0.0;
#pragma unroll
        for (int k = 0; k < E; ++k) if (slot[k] >= 0) lds[slot[k]] = v[k];
        __syncthreads();
        double *mrow = m_rows + (int64_t)r * n;
#pragma unroll
        for (int k = 0; k < E; ++k) if (mi[k] >= 0) mrow[mi[k]] = lds[k * kPermThreads + tid];
        __syncthreads();
    }
}

// Phase B: one destination tile's bucket of M (contiguous) -> LDS at destination offsets -> coalesced rows.
template <int E>
__global__ __launch_bounds__(kPermThreads) void k_perm_b(const RowView dst, const double *__restrict__ m_rows,
                                                         int64_t n, const uint16_t *__restrict__ slot_b, int64_t t0,
                                                         int32_t nrows, int32_t rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t base = (int64_t)blockIdx.x * (E * kPermThreads);
    const int tid = threadIdx.x;
    int32_t slot[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int64_t g = base + k * kPermThreads + tid;
        slot[k] = g < n ? (int32_t)slot_b[g] : -1;
    }
    const int32_t r0 = (int32_t)blockIdx.y * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    for (int32_t r = r0; r < r1; ++r) {
        const double *mrow = m_rows + (int64_t)r * n + base;
        double v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = slot[k] >= 0 ? mrow[k * kPermThreads + tid] : 0.0;
#pragma unroll
        for (int k = 0; k < E; ++k) if (slot[k] >= 0) lds[slot[k]] = v[k];
        __syncthreads();
        double *drow = dst.row(t0 + r) + base;
#pragma unroll
        for (int k = 0; k < E; ++k) if (slot[k] >= 0) drow[k * kPermThreads + tid] = lds[k * kPermThreads + tid];
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_state_in(double *x0, double *x1, double *x2, const double *q_t,
                                                     const int32_t *perm, int32_t n)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= n) return;
    const double v = q_t[perm[p]];
    x0[p] = v; x1[p] = v; x2[p] = v;
}

// q_t[i] = value written at the reach's last tick, lag + total_substeps - 1
__global__ __launch_bounds__(kBlock) void k_state_out(double *q_t, const double *x, int64_t n64,
                                                      const int32_t *lag, const int32_t *inv, int32_t n,
                                                      int64_t total_substeps)
{
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n) return;
    const int32_t p = inv[i];
    const int64_t last = (int64_t)(lag[p] & kLagMask) + total_substeps - 1;
    q_t[i] = x[(last % 3) * n64 + p];
}

__global__ __launch_bounds__(kBlock) void k_unit_state_in(double *x0, double *x1, double *x2, double *qch,
                                                          const double *q_ch, const double *q_full,
                                                          const int32_t *inner_pos, int32_t n_inner)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    const int32_t p = inner_pos[k];
    const double v = q_full[k];
    x0[p] = v; x1[p] = v; x2[p] = v;
    qch[p] = q_ch[k];
}

__global__ __launch_bounds__(kBlock) void k_unit_state_out(double *q_ch, double *q_full, const double *x,
                                                           int64_t n64, const double *qch, const int32_t *lag,
                                                           const int32_t *inner_pos, int32_t n_inner,
                                                           int64_t total_substeps)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    const int32_t p = inner_pos[k];
    const int64_t last = (int64_t)(lag[p] & kLagMask) + total_substeps - 1;
    q_full[k] = x[(last % 3) * n64 + p];
    q_ch[k] = qch[p];
}

}  // namespace
