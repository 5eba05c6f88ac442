// rr_engine.hip -- gfx950 kernels, the streaming executor and the C ABI of librr_hip.so.
//
// The math (river_route/routers/_numba_kernels.py:63-84 in gather form):
//   q+[i] = c3[i] q[i] + c4dt[i] ql[t,i] + c2[i] * sum_{u in up(i)} q[u] + sum_{u in up(i)} c1[i] q+[u]
// The second sum is a sparse triangular solve down the river tree.  Instead of sweeping the tree level by
// level inside one time step, the engine PIPELINES time down the tree: at routing tick tau the reach at
// engine position p advances its own sub-step  ts = tau - lag[p]  (lag = levels between p and the farthest
// headwater of the whole network).  Because lag(down) = lag(up) + 1 on every edge, the upstream values a
// reach needs -- q+[u] at ts and q[u] at ts-1 -- are exactly what its upstream reaches wrote one and two
// ticks ago.  Every tick is therefore one dependency-free, fully coalesced streaming kernel over all
// reaches; there are T*nsub + depth - 1 ticks in a call.  State lives in three rotating buffers X[tau % 3].
//
// Layout: rr_plan.hpp.  No CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rr_hip.h"
#include "rr_plan.hpp"

#define RR_VERSION_NUM 100

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(RR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                 \
    } while (0)

constexpr int kBlock = 256;
constexpr int kSampleGroup = 16;   // routing-tick launches per HIP-event bracket
// lag[] carries two flag bits for the boundary reaches of a partitioned network
constexpr int32_t kGhostBit = 1 << 30;    // value prescribed from the ghost series (an upstream reach owned by another GPU)
constexpr int32_t kExportBit = 1 << 29;   // value also copied to the export series (feeds another GPU)
constexpr int32_t kLagMask = kExportBit - 1;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// Index arithmetic: every tick, row and chunk index of a call is below 2^31 (checked in session_begin), and the
// hardware has no integer divide -- a 64-bit `%` is a ~100-instruction emulation.  Division by a run-time constant
// goes through a host-computed double reciprocal instead: floor(x * (1/d)) is exact or one too small (only when d
// divides x), which the single fix-up repairs.
struct Div32 {
    uint32_t d;
    double inv;
    Div32() = default;
    __host__ __device__ explicit Div32(uint32_t d_) : d(d_ ? d_ : 1u), inv(1.0 / (double)(d_ ? d_ : 1u)) {}
    __device__ __forceinline__ uint32_t div(uint32_t x, uint32_t &rem) const
    {
        uint32_t q = (uint32_t)((double)x * inv);
        uint32_t r = x - q * d;
        if (r >= d) { r -= d; ++q; }
        rem = r;
        return q;
    }
    __device__ __forceinline__ uint32_t mod(uint32_t x) const { uint32_t r; div(x, r); return r; }
};

struct TickArgs {
    const int32_t *child_ptr;  // [n+1]
    const int32_t *lag;        // [n]
    const double *w;           // [n] c1 of the downstream reach, stored at the UPSTREAM position
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
    double r = a.c3[p] * a.xa[p];
    if (HAS_LATERAL) r += a.c4[p] * a.in[(int64_t)a.in_rows.mod(t) * a.in_ld + p];
    const double c2 = a.c2[p];
    for (int32_t u = u0; u < u1; ++u) r += c2 * a.xb[u];
    for (int32_t u = u0; u < u1; ++u) r += a.w[u] * a.xa[u];
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
    const int32_t ts = (int32_t)a.tau - (a.lag[p] & kLagMask);
    if (ts < 0 || ts >= (int32_t)a.total_substeps) return;
    uint32_t t, s;
    if (SINGLE_SUBSTEP) { t = (uint32_t)ts; s = 0; }
    else t = a.nsub.div((uint32_t)ts, s);

    const double lat = a.in[(int64_t)a.in_rows.mod(t) * a.in_ld + p];
    const int32_t u0 = a.child_ptr[p], u1 = a.child_ptr[p + 1];
    if (u0 == u1) {  // headwater: discharge is the lateral inflow, unclamped and un-averaged (lines 122-123)
        a.xc[p] = lat;
        if (s == 0) a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = lat;
        return;
    }
    const int32_t uh = u0 + (int32_t)ua.hw_children[p];
    double r = a.c3[p] * ua.qch[p];
    const double c2 = a.c2[p];
    for (int32_t u = u0; u < uh; ++u) r += c2 * a.xa[u];   // headwater tributaries: "old" value is l_t too
    for (int32_t u = uh; u < u1; ++u) r += c2 * a.xb[u];
    for (int32_t u = u0; u < u1; ++u) r += a.w[u] * a.xa[u];
    ua.qch[p] = r;
    const double qfull = r + lat;
    a.xc[p] = qfull;

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

// Row addressing of a (rows, ld) array read or written cyclically: row of step t is (t - t0) % rows.
struct RowView {
    double *base;
    int64_t ld;
    int64_t t0;
    Div32 rows;
    RowView() = default;
    RowView(double *base_, int64_t ld_, int64_t t0_, uint32_t rows_) : base(base_), ld(ld_), t0(t0_), rows(rows_) {}
    __device__ __forceinline__ double *row(int64_t t) const { return base + (int64_t)rows.mod((uint32_t)(t - t0)) * ld; }
};

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

// ---- time-tiled routing: blocks of positions advance K ticks per launch ----
//
// k_tick streams ~88 B per reach-step because nothing survives from one tick to the next.  k_wave cuts the
// engine order into blocks of BS = PPT * 1024 consecutive positions and lets one workgroup advance its block
// by K ticks: coefficients and state sit in registers, the block's own discharges in LDS (double-buffered, one
// barrier per tick), so HBM sees only the lateral read and the discharge write of every reach-step, plus
//   * the block's state and coefficients once per K ticks, and
//   * the "halo": upstream reaches of a block's first level live in the block(s) to its LEFT (upstream reaches
//     always have smaller positions), so every block writes the values of its last level to a small history
//     ring hist[tick % hist_rows][p] and its right neighbour reads them one tick later.
// Task (block b, tick-chunk c) needs (b-1, c) (halo) and (b, c-1) (own state): all tasks on one anti-diagonal
// b + c are independent, so there is ONE LAUNCH PER DIAGONAL and still no inter-workgroup synchronisation.

struct WaveArgs {
    const int32_t *child_ptr, *lag;
    const double *c1row, *c2, *c3, *c4;   // c1row: the (uniform) weight of a reach's upstream terms
    double *sq, *ss, *si;                 // carried state: discharge, sum of upstream discharges one tick back, interval sum
    double *sqch;                         // UnitMuskingum: channel-only discharge of inner reaches
    const uint16_t *hw_children;          // UnitMuskingum: headwater tributaries come first in a reach's upstream range
    double *hist;                         // [hist_rows, n]
    const int32_t *bidx;                  // ghost / export slots (flag bits live in lag[])
    const double *ghost;
    double *exports;
    int32_t n_ghost, n_export;
    const double *in;
    double *out;
    int64_t in_ld, out_ld;
    Div32 in_rows, out_rows;
    double *rec;                          // record ring [rec_chunks][n][16] (record mode)
    Div32 rec_chunks;
#ifdef RR_WAVE_TRACE
    long long *trace; int64_t trace_diag;   // development build: per-block timestamps of one diagonal (profiles/microbench/wave_dbg.py)
#endif
    int32_t n, hist_rows, K, b_first, lh;   // lh: LDS positions per tick buffer (halo capacity + block)
    int64_t diag, total;
    Div32 nsub;
    double inv_nsub;
};

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its global loads/stores, so the
// lateral/halo prefetches stay in flight across ticks (__syncthreads() would drain vmcnt every tick).
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Predicated global stores without a branch: a raw buffer store whose byte offset is pushed past the end of the
// buffer is dropped by the bounds check.  Unlike `if (cond) *ptr = v` the instruction is always issued, so hipcc can
// count it in vmcnt and the in-order wait for an older prefetch does not have to assume the worst (which drained the
// younger prefetches too and tied every tick to a full store round trip).
constexpr uint32_t kBufferFlags = 0x00020000;      // gfx9 raw buffer, 32-bit data format
constexpr uint32_t kDropStore = 0xFFFFFFF0u;       // offset outside any buffer this file creates (< 4 GiB - 16)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, (int)kBufferFlags);
}
__device__ __forceinline__ void store_f64(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, double v)
{
    u32x2 bits;
    __builtin_memcpy(&bits, &v, sizeof bits);
    __builtin_amdgcn_raw_buffer_store_b64(bits, r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void store_f64x2(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, double2 v)
{
    u32x4 bits;
    __builtin_memcpy(&bits, &v, sizeof bits);
    __builtin_amdgcn_raw_buffer_store_b128(bits, r, (int)byte_off, 0, 0);
}

// PPT positions per thread; HPT halo values per thread (halo <= HPT * 1024 positions); two ticks of HBM
// prefetch in flight (stages A/B, the tick loop is unrolled by two so the stage registers are static).
template <int TH, int PPT, int HPT, bool SINGLE_SUBSTEP, bool UNIT>
__global__ __launch_bounds__(TH) void k_wave(const WaveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [2][lh]: positions [h0, b1) of one tick
    constexpr int BS = PPT * TH;
    const int tid = threadIdx.x;
    const int32_t b = a.b_first + (int32_t)blockIdx.x;
    const int64_t chunk = a.diag - b;
    const int32_t b0 = b * BS, b1 = min(a.n, b0 + BS);
    const int32_t h0 = min(a.child_ptr[b0], b0);          // first halo position (upstream reaches left of the block)
    const int32_t nh = b0 - h0;
    const int32_t halo_lo = max(b0, a.child_ptr[b1]);     // own positions read by blocks to the right
    auto position = [&](int k) { return b0 + k * TH + tid; };
    // lg keeps the ghost / export flag bits of lag[]; slot is the column of a flagged reach in its boundary series
    // UNIT (UnitMuskingum, _numba_kernels.py:113-171): q is what a reach publishes (q_full, or the lateral itself for a
    // headwater), qch the channel-only discharge; uh splits the upstream range into headwater and inner tributaries.
    int32_t lg[PPT], u0[PPT], u1[PPT], slot[PPT], uh[UNIT ? PPT : 1];
    double c1[PPT], c2[PPT], c3[PPT], c4[PPT], q[PPT], s_prev[PPT], isum[SINGLE_SUBSTEP ? 1 : PPT], qch[UNIT ? PPT : 1];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int32_t p = position(k);
        slot[k] = 0;
        if (UNIT) { uh[k] = 0; qch[k] = 0.0; }
        if (p < b1) {
            lg[k] = a.lag[p]; u0[k] = a.child_ptr[p] - h0; u1[k] = a.child_ptr[p + 1] - h0;
            if (UNIT) { uh[k] = u0[k] + (int32_t)a.hw_children[p]; qch[k] = a.sqch[p]; }
            if (lg[k] & (kGhostBit | kExportBit)) slot[k] = a.bidx[p];
            c1[k] = a.c1row[p]; c2[k] = a.c2[p]; c3[k] = a.c3[p]; c4[k] = a.in ? a.c4[p] : 0.0;
            q[k] = a.sq[p]; s_prev[k] = a.ss[p];
            if (!SINGLE_SUBSTEP) isum[k] = a.si[p];
        } else {
            lg[k] = -1; u0[k] = u1[k] = 0; c1[k] = c2[k] = c3[k] = c4[k] = q[k] = s_prev[k] = 0.0;   // lg < 0: not a reach
            if (!SINGLE_SUBSTEP) isum[k] = 0.0;
        }
    }
    const int32_t tau0 = (int32_t)chunk * a.K, tau_end = tau0 + a.K, total = (int32_t)a.total;

    // Prefetches are branch-free (addresses are clamped to something valid, the value is ignored where it does
    // not apply) and nothing touches the loaded registers until the tick that consumes them, so hipcc leaves the
    // loads in flight across the barriers instead of waiting right behind each one.
    const double *lat_base = a.in ? a.in : a.sq;     // channel-only routing: any readable array, c4 is zero
    const Div32 lat_rows = a.in ? a.in_rows : Div32(1u);
    const int64_t lat_ld = a.in ? a.in_ld : 0;
    auto fetch_lat = [&](int32_t tau, double (&lat)[PPT]) {     // lateral of the row each reach is at
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int32_t ts = tau - (lg[k] < 0 ? 0 : (lg[k] & kLagMask));
            ts = ts < 0 ? 0 : (ts >= total ? total - 1 : ts);
            uint32_t sub;
            const uint32_t t = SINGLE_SUBSTEP ? (uint32_t)ts : a.nsub.div((uint32_t)ts, sub);
            const double *src = lat_base + (int64_t)lat_rows.mod(t) * lat_ld + min(position(k), b1 - 1);
            if (lg[k] >= 0 && (lg[k] & kGhostBit)) src = a.ghost + (int64_t)ts * a.n_ghost + slot[k];   // prescribed boundary inflow
            lat[k] = *src;
        }
    };
    // the history ring is walked one row per tick: rows of the next halo fetch and of the next tick's own writes
    const uint32_t hrows = (uint32_t)a.hist_rows;
    auto next_row = [&](uint32_t r) { return r + 1 == hrows ? 0u : r + 1; };
    uint32_t hr_fetch = Div32(hrows).mod((uint32_t)tau0 + hrows - 1), hr_tick = next_row(hr_fetch);
    auto fetch_halo = [&](double (&h)[HPT]) {      // the left neighbours' values of the next tick in sequence
        const double *hrow = a.hist + (int64_t)hr_fetch * a.n + h0;
        hr_fetch = next_row(hr_fetch);
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int32_t i = j * TH + tid;
            h[j] = hrow[i < nh ? i : 0];
        }
    };
    auto put_halo = [&](double *buf, const double (&h)[HPT]) {
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int32_t i = j * TH + tid;
            if (i < nh) buf[i] = h[j];
        }
    };
    auto tick = [&](int32_t tau, const double (&lat)[PPT], const double (&h)[HPT]) {
        const double *rd = lds + (size_t)((tau + 1) & 1) * a.lh;   // values of tick tau-1
        double *wr = lds + (size_t)(tau & 1) * a.lh;
        const __amdgpu_buffer_rsrc_t hist_row = make_rsrc(a.hist + (int64_t)hr_tick * a.n, (uint32_t)a.n * 8u);
        hr_tick = next_row(hr_tick);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            // a slot past the end of the network has lag -1: no upstream range, never active, publishes 0.0 to nobody
            const int32_t p = position(k);
            double s_cur = 0.0, s_hw = 0.0;
            if (UNIT) {
                for (int32_t u = u0[k]; u < uh[k]; ++u) s_hw += rd[u];       // headwater tributaries: "old" value is l_t too
                for (int32_t u = uh[k]; u < u1[k]; ++u) s_cur += rd[u];
            } else {
                for (int32_t u = u0[k]; u < u1[k]; ++u) s_cur += rd[u];
            }
            const int32_t ts = tau - (lg[k] & kLagMask);
            const bool active = ts >= 0 && ts < total;
            if (UNIT && active) {
                uint32_t sub = 0u;
                const uint32_t t = SINGLE_SUBSTEP ? (uint32_t)ts : a.nsub.div((uint32_t)ts, sub);
                double *orow = a.out + (int64_t)a.out_rows.mod(t) * a.out_ld;
                if (u0[k] == u1[k]) {   // headwater: discharge is the lateral inflow, unclamped and un-averaged
                    q[k] = lat[k];
                    if (sub == 0) orow[p] = lat[k];
                } else {
                    const double r = __builtin_fma(c1[k], s_hw + s_cur, __builtin_fma(c2[k], s_hw + s_prev[k], c3[k] * qch[k]));
                    qch[k] = r;
                    const double qfull = r + lat[k];
                    q[k] = qfull;
                    if (SINGLE_SUBSTEP) {
                        orow[p] = qfull > 0.0 ? qfull : 0.0;
                    } else {
                        const double acc = (sub == 0 ? 0.0 : isum[k]) + qfull;
                        if (sub + 1 == a.nsub.d) { const double v = acc * a.inv_nsub; orow[p] = v > 0.0 ? v : 0.0; }
                        isum[k] = acc;
                    }
                }
            } else if (active && (lg[k] & kGhostBit)) {
                q[k] = lat[k];      // a ghost only republishes what its owner computed
            } else if (active) {
                // explicit fma: the unrolled copies of this tick must round identically (split run == joint run)
                const double r = __builtin_fma(c1[k], s_cur, __builtin_fma(c2[k], s_prev[k],
                                 __builtin_fma(c4[k], lat[k], c3[k] * q[k])));
                q[k] = r;
                if (lg[k] & kExportBit) a.exports[(int64_t)ts * a.n_export + slot[k]] = r;
                if (!SINGLE_SUBSTEP) {
                    uint32_t sub;
                    const uint32_t t = a.nsub.div((uint32_t)ts, sub);
                    const double acc = (sub == 0 ? 0.0 : isum[k]) + r;
                    if (sub + 1 == a.nsub.d) {
                        const double v = acc * a.inv_nsub;
                        a.out[(int64_t)a.out_rows.mod(t) * a.out_ld + p] = v > 0.0 ? v : 0.0;
                    }
                    isum[k] = acc;
                }
            }
            if (SINGLE_SUBSTEP && !UNIT) {
                // the discharge row store is issued by every lane, every tick: lanes with nothing to write (pipeline
                // fill and drain, ghosts, slots past the network) aim at the interval-sum array, unused with one
                // sub-step.  An always-issued store is one hipcc can count in vmcnt (see store_f64).
                const bool writes = active && !(lg[k] & kGhostBit);
                double *dst = writes ? a.out + (int64_t)a.out_rows.mod((uint32_t)ts) * a.out_ld + p : a.si + min(p, a.n - 1);
                *dst = q[k] > 0.0 ? q[k] : 0.0;
            }
            s_prev[k] = s_cur;
            wr[nh + k * TH + tid] = q[k];
            store_f64(hist_row, p >= halo_lo ? (uint32_t)p * 8u : kDropStore, q[k]);     // positions >= n fall off the row
        }
        put_halo(wr, h);
    };

    // Three register stages, each fetched two ticks before it is consumed and BEFORE the tick in between issues
    // its stores: vmcnt retires in order, so a wait for stage s only has to cover ops older than the two younger
    // fetches and never the stores of the tick just finished.
    double lat0[PPT], lat1[PPT], lat2[PPT], h0v[HPT], h1v[HPT], h2v[HPT];
    {   // tick tau0 reads the buffer of tick tau0 - 1: own discharges + the halo row of that tick
        double *buf = lds + (size_t)((tau0 + 1) & 1) * a.lh;
        fetch_halo(h0v);                                      // tick tau0 - 1
#pragma unroll
        for (int k = 0; k < PPT; ++k) buf[nh + k * TH + tid] = q[k];
        put_halo(buf, h0v);
    }
    fetch_lat(tau0, lat0); fetch_halo(h0v);
    fetch_lat(tau0 + 1, lat1); fetch_halo(h1v);               // rows past the chunk are clamped, never used
    barrier_lds();
    for (int32_t tau = tau0; tau < tau_end; tau += 3) {
        fetch_lat(tau + 2, lat2); fetch_halo(h2v);
        tick(tau, lat0, h0v);
        barrier_lds();
        if (tau + 1 < tau_end) {
            fetch_lat(tau + 3, lat0); fetch_halo(h0v);
            tick(tau + 1, lat1, h1v);
        }
        barrier_lds();
        if (tau + 2 < tau_end) {
            fetch_lat(tau + 4, lat1); fetch_halo(h1v);
            tick(tau + 2, lat2, h2v);
        }
        barrier_lds();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (lg[k] < 0) continue;
        const int32_t p = position(k);
        a.sq[p] = q[k]; a.ss[p] = s_prev[k];
        if (!SINGLE_SUBSTEP) a.si[p] = isum[k];
        if (UNIT) a.sqch[p] = qch[k];
    }
}

__device__ __forceinline__ void load_f64x2(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, double &x, double &y)
{
    const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);   // one 16-byte request per lane
    double2 v;
    __builtin_memcpy(&v, &bits, sizeof v);
    x = v.x; y = v.y;
}

// The tick loops are fully unrolled (the record slots are registers), so anything derived from a per-position constant
// is "loop invariant" and hipcc keeps every such derivative in a VGPR for the whole task.  fresh() hands the constant
// back as an opaque value: the two-instruction unpacking is redone each tick and the registers stay free.
__device__ __forceinline__ int32_t fresh(int32_t v) { asm volatile("" : "+v"(v)); return v; }

// ---- record mode ----
// With one sub-step per row, a task (block, chunk of kRec = 16 ticks) consumes for every position exactly the 16
// consecutive rows t = tick - lag.  The work ring is therefore kept as RECORDS indexed by tick:
//     rec[(tick / 16) % chunks][position][tick % 16]        (128 bytes per position and chunk)
// A task reads its block's records as one contiguous 128 B * BS stream into registers, routes 16 ticks with the
// lateral values (already scaled by c4dt) taken from and the discharges written back to those registers, and
// stores the records in place.  The permutation to and from params order becomes ONE pass each way that moves
// whole 128-byte records (k_rec_in / k_rec_out) instead of two tiled passes over rows.
constexpr int kRec = 16;
// Record stores go through a per-wave LDS transpose: a lane owns a position (its record lives in registers), but a
// store instruction in which every lane writes 16 bytes of a different record costs L2 one request per lane.  After
// the transpose four neighbouring lanes write the 64 contiguous bytes of one half record: a quarter of the requests.
constexpr int kStageStride = 10;   // doubles per position in the staging area: 64 bytes + 16 of padding (bank spread, skip flag)
// Lanes of one wave exchange data through its staging area without a workgroup barrier: a wave's LDS instructions
// execute in order.  The compiler still has to be told that other lanes wrote (it would reuse earlier reads).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// positions transposed at a time: half a wave (2.5 KiB of staging per wave).  (A 512-thread x 2-position shape with
// quarter-wave staging fits two blocks per CU; measured at 500k reaches it is 20 % slower than one 1,024 x 2 block:
// twice the halo traffic and barriers, and the load and tick phases of co-resident blocks do not overlap usefully.)
constexpr int stage_lanes(int) { return 32; }
// LDS in doubles: X[2][lh] | c1[BS] | c2[BS] | c3[BS] | stage[waves][stage_lanes * kStageStride].  The three
// coefficients and the own discharge are read from LDS once per tick: the registers go to the records.
constexpr size_t wave_rec_lds_bytes(int64_t lh, int threads, int ppt)
{
    return (size_t)(2 * lh + 3 * (int64_t)ppt * threads + (threads / 64) * stage_lanes(threads) * kStageStride) * sizeof(double);
}

template <int TH, int PPT, int HPT, bool UNIT>
__global__ __launch_bounds__(TH) void k_wave_rec(const WaveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int BS = PPT * TH;
    constexpr int NS = 3;                  // history rows in flight (register stages)
    constexpr int kStageLanes = stage_lanes(TH);
    const int tid = threadIdx.x;
    const int32_t b = a.b_first + (int32_t)blockIdx.x;
    const int64_t chunk = a.diag - b;
    const int32_t b0 = b * BS, b1 = min(a.n, b0 + BS);
    const int32_t h0 = min(a.child_ptr[b0], b0);
    const int32_t nh = b0 - h0;
    const int32_t halo_lo = max(b0, a.child_ptr[b1]);
    auto position = [&](int k) { return b0 + k * TH + tid; };
    const int32_t tau0 = (int32_t)chunk * kRec, total = (int32_t)a.total;
    const int lane = tid & 63;
    double *cc1 = lds + 2 * (size_t)a.lh, *cc2 = cc1 + BS, *cc3 = cc2 + BS;
    double *stage = cc3 + BS + (size_t)(tid >> 6) * (kStageLanes * kStageStride);   // this wave's transpose area
#ifdef RR_WAVE_TRACE
    const bool trace = a.trace && a.diag == a.trace_diag && tid == 0;
    long long *tq = a.trace + (int64_t)b * 8;
#define RR_TRACE(i) do { if (trace) tq[i] = wall_clock64(); } while (0)
#else
#define RR_TRACE(i) do { } while (0)
#endif
    RR_TRACE(0);

    double *rbase = a.rec + (int64_t)a.rec_chunks.mod((uint32_t)chunk) * a.n * kRec;
    const __amdgpu_buffer_rsrc_t rec_chunk = make_rsrc(rbase, (uint32_t)a.n * 128u);   // n < 2^25 in record mode (session_begin)
    double *first_buf = lds + (size_t)((tau0 + 1) & 1) * a.lh;     // tick tau0 reads the buffer of tick tau0 - 1

    // A slot past the end of the network keeps lag -1: no upstream range, never active, publishes 0.0 to nobody.
    // up[k]: LDS slot of the first upstream value (low 16 bits) and the number of upstream reaches (high 16 bits)
    int32_t lg[PPT], up[PPT], uh[UNIT ? PPT : 1];
    double s_prev[PPT], qch[UNIT ? PPT : 1];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int32_t p = position(k);
        if (UNIT) { uh[k] = 0; qch[k] = 0.0; }
        if (p < b1) {
            const int32_t first_up = a.child_ptr[p];
            lg[k] = a.lag[p]; up[k] = (first_up - h0) | ((a.child_ptr[p + 1] - first_up) << 16);
            if (UNIT) { uh[k] = (first_up - h0) + (int32_t)a.hw_children[p]; qch[k] = a.sqch[p]; }
            s_prev[k] = a.ss[p];
            first_buf[nh + k * TH + tid] = a.sq[p];
            cc1[k * TH + tid] = a.c1row[p]; cc2[k * TH + tid] = a.c2[p]; cc3[k * TH + tid] = a.c3[p];
        } else {
            lg[k] = -1; up[k] = 0; s_prev[k] = 0.0;
            first_buf[nh + k * TH + tid] = 0.0;
            cc1[k * TH + tid] = cc2[k * TH + tid] = cc3[k * TH + tid] = 0.0;
        }
    }
    // Records: like the stores, the loads are issued four lanes per 64-byte sector (a quarter of the L2 requests of
    // one record per lane), all of them back to back, and then handed to the owning lanes through the staging area.
    double rec[PPT][kRec];
    {
        constexpr int HW = 64 / kStageLanes, MS = kStageLanes / 16, ROUNDS = PPT * 2 * HW;   // round = (k, half, h)
        double raw_x[ROUNDS * MS], raw_y[ROUNDS * MS];      // (an array of double2 is not promoted to registers)
#pragma unroll
        for (int i = 0; i < ROUNDS * MS; ++i) {
            const int k = i / (2 * HW * MS), half = i / (HW * MS) % 2, h = i / MS % HW, m = i % MS;
            const int32_t pos = min(b0 + k * TH + (tid - lane) + h * kStageLanes + 16 * m + (lane >> 2), b1 - 1);
            load_f64x2(rec_chunk, (uint32_t)pos * 128u + (uint32_t)(half * 64 + (lane & 3) * 16), raw_x[i], raw_y[i]);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int k = r / (2 * HW), half = r / HW % 2, h = r % HW;
#pragma unroll
            for (int m = 0; m < MS; ++m)
                reinterpret_cast<double2 *>(stage + (16 * m + (lane >> 2)) * kStageStride)[lane & 3] = make_double2(raw_x[r * MS + m], raw_y[r * MS + m]);
            wave_lds_fence();
            if (lane / kStageLanes == h) {
                const double2 *mine = reinterpret_cast<const double2 *>(stage + (lane % kStageLanes) * kStageStride);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double2 v = mine[j]; rec[k][8 * half + 2 * j] = v.x; rec[k][8 * half + 2 * j + 1] = v.y; }
            }
            wave_lds_fence();
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (lg[k] < 0) {
#pragma unroll
            for (int j = 0; j < kRec; ++j) rec[k][j] = 0.0;
        } else if (lg[k] & kGhostBit) {   // boundary inflow: this chunk of the ghost series instead of the ring
            const int32_t g = a.bidx[position(k)];
#pragma unroll
            for (int j = 0; j < kRec; ++j) {
                int32_t ts = tau0 + j - (lg[k] & kLagMask);
                ts = ts < 0 ? 0 : (ts >= total ? total - 1 : ts);
                rec[k][j] = a.ghost[(int64_t)ts * a.n_ghost + g];
            }
        }
    }
    const bool has_lat = a.in != nullptr;   // channel-only routing: the records only carry discharge

    // the history ring is walked one row per tick: rows of the next halo fetch and of the next tick's own writes
    const uint32_t hrows = (uint32_t)a.hist_rows;
    auto next_row = [&](uint32_t r) { return r + 1 == hrows ? 0u : r + 1; };
    uint32_t hr_fetch = Div32(hrows).mod((uint32_t)tau0 + hrows - 1), hr_tick = next_row(hr_fetch);
    auto fetch_halo = [&](double (&h)[HPT]) {      // one history row per call, starting at tick tau0 - 1
        const double *hrow = a.hist + (int64_t)hr_fetch * a.n + h0;
        hr_fetch = next_row(hr_fetch);
        const int32_t t = fresh(tid);
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int32_t i = j * TH + t;
            h[j] = hrow[i < nh ? i : 0];
        }
    };
    auto put_halo = [&](double *buf, const double (&h)[HPT]) {
        const int32_t t = fresh(tid);
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int32_t i = j * TH + t;
            if (i < nh) buf[i] = h[j];
        }
    };
    // Eight slots of every record are final: write that 64-byte sector.  Half a wave at a time parks its sectors in
    // the wave's staging area, then all 64 lanes store them, four lanes per sector.
    auto store_half_records = [&](int k, int half) {
#pragma unroll
        for (int h = 0; h < 64 / kStageLanes; ++h) {
            if (lane / kStageLanes == h) {
                double2 *mine = reinterpret_cast<double2 *>(stage + (lane % kStageLanes) * kStageStride);
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[j] = make_double2(rec[k][8 * half + 2 * j], rec[k][8 * half + 2 * j + 1]);
                reinterpret_cast<int32_t *>(mine + 4)[0] = (lg[k] < 0 || (lg[k] & kGhostBit)) ? 1 : 0;   // not this block's to write
            }
            wave_lds_fence();
            const uint32_t first = (uint32_t)(b0 + k * TH + (tid - lane) + h * kStageLanes) * 128u + (uint32_t)half * 64u;
#pragma unroll
            for (int m = 0; m < kStageLanes / 16; ++m) {       // lanes 4i .. 4i+3: the four 16-byte pieces of position 16 m + i
                const int pm = 16 * m + (lane >> 2), piece = lane & 3;
                const double2 *theirs = reinterpret_cast<const double2 *>(stage + pm * kStageStride);
                const double2 v = theirs[piece];
                const bool skip = reinterpret_cast<const int32_t *>(theirs + 4)[0] != 0;
                store_f64x2(rec_chunk, skip ? kDropStore : first + (uint32_t)pm * 128u + (uint32_t)piece * 16u, v);
            }
            wave_lds_fence();
        }
    };

    double hs[NS][HPT];
    fetch_halo(hs[NS - 1]);
    put_halo(first_buf, hs[NS - 1]);
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) fetch_halo(hs[i]);
    barrier_lds();
    RR_TRACE(1);
#pragma unroll
    for (int s = 0; s < kRec; ++s) {
        const int32_t tau = tau0 + s;
        if (s == 1) RR_TRACE(2);
        if (s == 8) RR_TRACE(3);
        fetch_halo(hs[(s + NS - 1) % NS]);
        const double *rd = lds + (size_t)((tau + 1) & 1) * a.lh;
        double *wr = lds + (size_t)(tau & 1) * a.lh;
        const __amdgpu_buffer_rsrc_t hist_row = make_rsrc(a.hist + (int64_t)hr_tick * a.n, (uint32_t)a.n * 8u);
        hr_tick = next_row(hr_tick);
        const int32_t t = fresh(tid);       // slot and offset arithmetic is redone per tick, not held in registers
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int32_t p = b0 + k * TH + t, lgk = fresh(lg[k]), upk = fresh(up[k]);
            const int32_t u0 = upk & 0xFFFF, u1 = u0 + (upk >> 16);
            double qk = rd[nh + k * TH + t];       // own discharge one tick back
            const double c1 = cc1[k * TH + t], c2 = cc2[k * TH + t], c3 = cc3[k * TH + t];
            double s_cur = 0.0, s_hw = 0.0;
            if (UNIT) {   // headwater tributaries come first in the upstream range
                for (int32_t u = u0; u < uh[k]; ++u) s_hw += rd[u];
                for (int32_t u = uh[k]; u < u1; ++u) s_cur += rd[u];
            } else {
                for (int32_t u = u0; u < u1; ++u) s_cur += rd[u];
            }
            const int32_t ts = tau - (lgk & kLagMask);
            if (ts >= 0 && ts < total) {
                const double lat = has_lat ? rec[k][s] : 0.0;
                if (UNIT) {
                    if (u0 == u1) {
                        qk = lat;        // headwater: discharge = lateral, the record slot already holds it
                    } else {
                        const double r = __builtin_fma(c1, s_hw + s_cur, __builtin_fma(c2, s_hw + s_prev[k], c3 * qch[k]));
                        qch[k] = r;
                        qk = r + lat;
                        rec[k][s] = qk > 0.0 ? qk : 0.0;
                    }
                } else if (lgk & kGhostBit) {
                    qk = rec[k][s];
                } else {
                    // explicit fma: every copy of this tick must round identically (split run == joint run)
                    qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev[k], __builtin_fma(c3, qk, lat)));
                    if (lgk & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[p]] = qk;
                    rec[k][s] = qk > 0.0 ? qk : 0.0;
                }
            }
            s_prev[k] = s_cur;
            wr[nh + k * TH + t] = qk;
            store_f64(hist_row, p >= halo_lo ? (uint32_t)p * 8u : kDropStore, qk);   // positions >= n fall off the row
        }
        put_halo(wr, hs[s % NS]);
        if ((s & 7) == 7) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) store_half_records(k, s >> 3);
        }
        barrier_lds();
    }
    const double *last = lds + (size_t)((tau0 + kRec - 1) & 1) * a.lh;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (lg[k] < 0) continue;
        const int32_t p = position(k);
        a.sq[p] = last[nh + k * TH + tid]; a.ss[p] = s_prev[k];
        if (UNIT) a.sqch[p] = qch[k];
    }
    RR_TRACE(4);
#undef RR_TRACE
}

// sq = q0 in engine order, ss = sum of the upstream reaches' q0, every history row = q0
__global__ __launch_bounds__(kBlock) void k_wave_state_in(double *sq, double *ss, double *hist, int32_t hist_rows,
                                                          const double *q_t, const int32_t *perm,
                                                          const int32_t *child_ptr, int32_t n)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= n) return;
    const double v = q_t[perm[p]];
    double s = 0.0;
    for (int32_t u = child_ptr[p]; u < child_ptr[p + 1]; ++u) s += q_t[perm[u]];
    sq[p] = v; ss[p] = s;
    for (int32_t r = 0; r < hist_rows; ++r) hist[(int64_t)r * n + p] = v;
}

// UnitMuskingum state for the time-tiled kernel: published discharge = q_full on inner reaches (0 on headwaters until
// their first tick), ss = sum over the INNER tributaries only, qch = channel discharge; history rows = published values.
__global__ __launch_bounds__(kBlock) void k_wave_unit_state_in(double *sq, double *ss, double *qch, double *hist,
                                                               int32_t hist_rows, const double *x0,
                                                               const int32_t *child_ptr, int32_t n)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= n) return;
    const double v = x0[p];     // x0: q_full scattered to engine positions, zeros on headwaters (k_unit_state_in)
    double s = 0.0;
    for (int32_t u = child_ptr[p]; u < child_ptr[p + 1]; ++u)
        if (child_ptr[u + 1] > child_ptr[u]) s += x0[u];
    sq[p] = v; ss[p] = s;
    for (int32_t r = 0; r < hist_rows; ++r) hist[(int64_t)r * n + p] = v;
    (void)qch;
}

__global__ __launch_bounds__(kBlock) void k_wave_unit_state_out(double *q_ch, double *q_full, const double *sq,
                                                                const double *qch, const int32_t *inner_pos,
                                                                int32_t n_inner)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    const int32_t p = inner_pos[k];
    q_full[k] = sq[p];
    q_ch[k] = qch[p];
}

__global__ __launch_bounds__(kBlock) void k_wave_state_out(double *q_t, const double *sq, const int32_t *inv, int32_t n)
{
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) q_t[i] = sq[inv[i]];
}

// Unit-hydrograph convolution, direct form (UnitHydrograph.py:93-107):
//   out[t, i] = [t < n_ks] state[t, i] + sum_{s=0}^{min(t, n_ks-1)} kernel[s, i] * lateral[t - s, i]
// One reach per lane, TB consecutive outputs per thread held in registers; per tap one kernel value and one
// new lateral value are loaded and the TB-wide window slides in registers.
template <int TB>
__global__ __launch_bounds__(kBlock) void k_uh_convolve(const double *__restrict__ kernel,
                                                        const double *__restrict__ state,
                                                        const double *__restrict__ lateral,
                                                        double *__restrict__ out, int64_t T, int32_t n_ks, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t t0 = (int64_t)blockIdx.y * TB;
    double acc[TB], win[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int64_t t = t0 + j;
        acc[j] = (t < n_ks && t < T) ? state[t * n + i] : 0.0;
        win[j] = (t < T) ? lateral[t * n + i] : 0.0;   // lateral[t0 + j - s] for s = 0
    }
    for (int32_t s = 0; s < n_ks; ++s) {
        const double kv = kernel[(int64_t)s * n + i];
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[j] += kv * win[j];
        // slide: win[j] <- lateral[t0 + j - (s+1)]
#pragma unroll
        for (int j = TB - 1; j > 0; --j) win[j] = win[j - 1];
        const int64_t tn = t0 - (s + 1);
        win[0] = (tn >= 0) ? lateral[tn * n + i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < TB; ++j)
        if (t0 + j < T) out[(t0 + j) * n + i] = acc[j];
}

// Long-series form of the same convolution: one reach per lane walks its whole time segment once.  The kernel
// column sits in registers (static indices: the tap loop is unrolled over the padded length NK), the last NK
// lateral values in an LDS ring [slot][lane] (conflict-free), so HBM sees each lateral row and each output row
// exactly once and the kernel taps once per segment -- k_uh_convolve re-reads the taps for every 8 rows.
#ifndef RR_UH_THREADS
#define RR_UH_THREADS 128
#endif
#ifndef RR_UH48
#define RR_UH48 64, 48, 8, 2
#endif
constexpr int kUhThreads = RR_UH_THREADS;

// NK window slots (power of two), R outputs per pass (every window value read from LDS feeds R accumulators),
// D passes of lateral rows in flight.  The window costs NK * 8 B of LDS per thread, which caps the kernel at about
// one wave per SIMD: latency is hidden by depth instead (R * D rows per lane in flight; registers are free at
// that occupancy).
template <int NK, int NT, int R, int D>    // NT taps held in registers (n_ks <= NT <= NK - (R - 1))
__global__ __launch_bounds__(kUhThreads) void k_uh_convolve_ring(const double *__restrict__ kernel,
                                                                const double *__restrict__ state,
                                                                const double *__restrict__ lateral,
                                                                double *__restrict__ out, int64_t T, int32_t n_ks,
                                                                int64_t n, int64_t seg_rows)
{
    extern __shared__ __attribute__((aligned(16))) double win[];   // [NK][kUhThreads]
    static_assert(NT + R - 1 <= NK, "window ring too small");
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kUhThreads + tid;
    const int64_t t0 = (int64_t)blockIdx.y * seg_rows, t1 = min(T, t0 + seg_rows);
    const bool live = i < n;
    const int64_t col = live ? i : 0;
    double kv[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) kv[s] = (live && s < n_ks) ? kernel[(int64_t)s * n + col] : 0.0;
    // slot of lateral[t] is t & (NK - 1); preload the rows before the segment
#pragma unroll
    for (int s = 1; s < NT; ++s) {
        const int64_t t = t0 - s;
        win[(size_t)((uint64_t)t & (NK - 1)) * kUhThreads + tid] = (t >= 0 && s < n_ks) ? lateral[t * n + col] : 0.0;
    }
    double nxt[D][R];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int j = 0; j < R; ++j) nxt[dd][j] = lateral[min(t0 + dd * R + j, T - 1) * n + col];
    // window value m = t + R - 1 - d is tap (j + d - (R - 1)) of output t + j.  With one wave per SIMD nothing else
    // hides the LDS latency: the window is read CH values at a time, one chunk ahead of the FMAs.
    constexpr int CH = 8, ND = NT + R - 1, NCH = (ND + CH - 1) / CH;
    constexpr int PASSES = NK / R;        // passes per group of NK rows
    constexpr bool STATIC_GROUPS = PASSES % D == 0;
    // Rows are handled in groups of NK (segments start at multiples of NK, rr_uh_convolve_dev).  A group that needs
    // no carried-in state, no clamped prefetch and no partial store runs with every window slot a compile-time
    // constant (the LDS offsets become immediates); the slot and row arithmetic of the general pass was two thirds
    // of its instructions, and with one wave per SIMD every instruction is on the critical path.
    for (int64_t tb = t0; tb < t1; tb += NK) {
        const bool fast = STATIC_GROUPS && tb >= n_ks && tb + NK <= t1 && tb + NK + R * D <= T;
        if (fast) {
            const double *lat_g = lateral + tb * n + col;     // row tb of this column
            double *out_g = out + tb * n + col;
#pragma unroll
            for (int pp = 0; pp < PASSES; ++pp) {
                constexpr int mask = NK - 1;
                const int dd = pp % D;
                double acc[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    win[(size_t)((pp * R + j) & mask) * kUhThreads + tid] = nxt[dd][j];
                    acc[j] = 0.0;
                }
#pragma unroll
                for (int j = 0; j < R; ++j) nxt[dd][j] = lat_g[(int64_t)(pp * R + R * D + j) * n];
                double wv[2][CH];
                auto read_chunk = [&](int c, double (&v)[CH]) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        const int d = c * CH + e;
                        if (d < ND) v[e] = win[(size_t)((pp * R + (R - 1) - d) & mask) * kUhThreads + tid];
                    }
                };
                read_chunk(0, wv[0]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 1 < NCH) read_chunk(c + 1, wv[(c + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        const int d = c * CH + e;
#pragma unroll
                        for (int j = 0; j < R; ++j) {
                            const int sidx = j + d - (R - 1);
                            if (d < ND && sidx >= 0 && sidx < NT) acc[j] = __builtin_fma(kv[sidx], wv[c & 1][e], acc[j]);
                        }
                    }
                }
                if (live) {
#pragma unroll
                    for (int j = 0; j < R; ++j) out_g[(int64_t)(pp * R + j) * n] = acc[j];
                }
            }
            continue;
        }
        const int64_t tg_end = min(t1, tb + NK);
        for (int64_t tg = tb; tg < tg_end; tg += R * D) {
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const int64_t t = tg + dd * R;
                if (t >= tg_end) break;
                double acc[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    win[(size_t)((uint64_t)(t + j) & (NK - 1)) * kUhThreads + tid] = nxt[dd][j];
                    acc[j] = (t + j < n_ks && t + j < T) ? state[(t + j) * n + col] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < R; ++j) nxt[dd][j] = lateral[min(t + R * D + j, T - 1) * n + col];     // D passes ahead
                double wv[2][CH];
                auto read_chunk = [&](int c, double (&v)[CH]) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        const int d = c * CH + e;
                        if (d < ND) v[e] = win[(size_t)((uint64_t)(t + (R - 1) - d) & (NK - 1)) * kUhThreads + tid];
                    }
                };
                read_chunk(0, wv[0]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 1 < NCH) read_chunk(c + 1, wv[(c + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        const int d = c * CH + e;
#pragma unroll
                        for (int j = 0; j < R; ++j) {
                            const int sidx = j + d - (R - 1);
                            if (d < ND && sidx >= 0 && sidx < NT) acc[j] = __builtin_fma(kv[sidx], wv[c & 1][e], acc[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < R; ++j) if (live && t + j < t1) out[(t + j) * n + i] = acc[j];
            }
        }
    }
}

// Carry-over tail: new_state[s, i] = buf[T + s, i] for s < n_ks - 1, 0 for s = n_ks - 1 (lines 103-105), where
// buf[m] = sum_{s'} kernel[s'] lateral[m - s'] (+ state[m] when m < n_ks).
__global__ __launch_bounds__(kBlock) void k_uh_tail(const double *__restrict__ kernel,
                                                    const double *__restrict__ state,
                                                    const double *__restrict__ lateral,
                                                    double *__restrict__ new_state, int64_t T, int32_t n_ks, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t s = (int32_t)blockIdx.y;
    if (s == n_ks - 1) { new_state[(int64_t)s * n + i] = 0.0; return; }
    const int64_t m = T + s;
    double acc = m < n_ks ? state[m * n + i] : 0.0;
    for (int32_t k = s + 1; k < n_ks; ++k) {
        const int64_t tt = m - k;
        if (tt < 0) break;
        acc += kernel[(int64_t)k * n + i] * lateral[tt * n + i];
    }
    new_state[(int64_t)s * n + i] = acc;
}

// ---- record-mode permutation (one pass each way), see k_wave_rec ----
// Column i of the params-order rows is engine position p = inv[i] with lag L = 16 * sh + o.  Row t of that column
// is slot (t + L) % 16 of record (t + L) / 16, so the 64 rows [64 j - o, 64 j + 64 - o) are exactly the four
// records 4 j + sh .. 4 j + sh + 3.  k_rec_in reads the 79 rows [64 j - 15, 64 j + 64) of a 64-column tile
// coalesced into LDS and writes four whole 128-byte records per column (8 lanes x 16 B per record); k_rec_out
// reads five records per column the same way and writes the 64 rows [64 j, 64 j + 64) of the tile coalesced.
#ifndef RR_REC_BATCH
#define RR_REC_BATCH 8
#define RR_REC_COLS 32
#endif
#ifndef RR_REC_THREADS
#define RR_REC_THREADS 256
#endif
constexpr int kRecCols = RR_REC_COLS, kRecBatch = RR_REC_BATCH, kRecThreads = RR_REC_THREADS;
constexpr int kRecRows = 16 * kRecBatch;    // rows of one batch

struct RecPermArgs {
    double *rec;
    Div32 rec_chunks;
    int64_t n, T, batch;
    const int2 *colmeta;      // per params column: {engine position, lag}
    const double *scale;      // c4dt in PARAMS order (RapidMuskingum: the ring holds c4dt * lateral) or NULL
    RowView rows;             // params-order rows (source of k_rec_in, destination of k_rec_out)
};

__global__ __launch_bounds__(kRecThreads) void k_rec_in(const RecPermArgs a)
{
    constexpr int R = 16 * kRecBatch + 15;
    __shared__ double tile[R][kRecCols + 1];
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * kRecCols;
    const int64_t row_first = kRecRows * a.batch - 15;
    {   // all row loads in flight first (branch-free: out-of-range rows/columns are clamped and zeroed afterwards)
        constexpr int RPT = (R + kRecThreads / kRecCols - 1) / (kRecThreads / kRecCols);
        const int c = tid % kRecCols, r0 = tid / kRecCols;
        const int64_t i = min(col0 + c, a.n - 1);
        double v[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int64_t t = row_first + r0 + q * (kRecThreads / kRecCols);
            v[q] = a.rows.row(t < 0 ? 0 : (t >= a.T ? a.T - 1 : t))[i];
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = r0 + q * (kRecThreads / kRecCols);
            const int64_t t = row_first + r;
            if (r < R) tile[r][c] = (t >= 0 && t < a.T && col0 + c < a.n) ? v[q] : 0.0;
        }
    }
    __syncthreads();
    constexpr int IT = kRecCols * kRecBatch * 8 / kRecThreads;
    int2 meta[IT];
    double f[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {     // all metadata loads first: they are independent
        const int64_t i = col0 + (it * kRecThreads + tid) / (8 * kRecBatch);
        meta[it] = i < a.n ? a.colmeta[i] : make_int2(-1, 0);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t i = col0 + (it * kRecThreads + tid) / (8 * kRecBatch);
        f[it] = (a.scale && i < a.n) ? a.scale[i] : 1.0;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int piece = it * kRecThreads + tid;       // (column, record, 16-byte part): 8 consecutive lanes = one record
        const int c = piece / (8 * kRecBatch), k = (piece >> 3) % kRecBatch, part = piece & 7;
        const int32_t p = meta[it].x;
        if (p < 0) continue;
        const int32_t lag = meta[it].y;
        const int o = lag & 15;
        const uint32_t chunk = (uint32_t)kRecBatch * (uint32_t)a.batch + (uint32_t)(lag >> 4) + k;
        const int r = 15 - o + 16 * k + 2 * part;
        const double v0 = tile[r][c] * f[it], v1 = tile[r + 1][c] * f[it];
        double2 *dst = reinterpret_cast<double2 *>(a.rec + ((int64_t)a.rec_chunks.mod(chunk) * a.n + p) * kRec) + part;
        *dst = make_double2(v0, v1);
    }
}

__global__ __launch_bounds__(kRecThreads) void k_rec_out(const RecPermArgs a)
{
    constexpr int S = 16 * (kRecBatch + 1);
    __shared__ double recs[kRecCols][S + 1];
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * kRecCols;
    constexpr int IT = kRecCols * (kRecBatch + 1) * 8 / kRecThreads;
    int2 meta[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t i = col0 + (it * kRecThreads + tid) / ((kRecBatch + 1) * 8);
        meta[it] = i < a.n ? a.colmeta[i] : make_int2(-1, 0);
    }
    double2 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {     // all record reads in flight before the first LDS write
        const int piece = it * kRecThreads + tid;
        const int k = (piece >> 3) % (kRecBatch + 1), part = piece & 7;
        const int32_t p = meta[it].x < 0 ? 0 : meta[it].x;
        const uint32_t chunk = (uint32_t)kRecBatch * (uint32_t)a.batch + (uint32_t)(meta[it].y >> 4) + k;
        v[it] = *(reinterpret_cast<const double2 *>(a.rec + ((int64_t)a.rec_chunks.mod(chunk) * a.n + p) * kRec) + part);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int piece = it * kRecThreads + tid;
        const int c = piece / ((kRecBatch + 1) * 8), k = (piece >> 3) % (kRecBatch + 1), part = piece & 7;
        recs[c][16 * k + 2 * part] = v[it].x;
        recs[c][16 * k + 2 * part + 1] = v[it].y;
    }
    __syncthreads();
    const int c = tid % kRecCols;
    const int64_t i = col0 + c;
    if (i >= a.n) return;
    const int o = a.colmeta[i].y & 15;
    for (int r = tid / kRecCols; r < 16 * kRecBatch; r += kRecThreads / kRecCols) {
        const int64_t t = kRecRows * a.batch + r;
        if (t < a.T) a.rows.row(t)[i] = recs[c][o + r];
    }
}

// Router post-processing on the device (TransformMuskingum.py:128-142): mean over `factor` consecutive rows
// (sequential sum then one division, as numpy's reduction over a strided axis does) and the float32 cast.
__global__ __launch_bounds__(kBlock) void k_resample_cast(const double *__restrict__ src, float *__restrict__ dst,
                                                          int64_t n, int64_t out_rows, int32_t factor)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t o = blockIdx.y;
    if (i >= n || o >= out_rows) return;
    const double *p = src + o * factor * n + i;
    double acc = p[0];
    for (int32_t j = 1; j < factor; ++j) acc += p[(int64_t)j * n];
    dst[o * n + i] = (float)(factor > 1 ? acc / (double)factor : acc);
}

// ---- gridded runoff -> catchment lateral inflow (river_route/runoff.py:288-330) ----
// qlateral[t, r] = sum_k weights[k] * runoff[t, point[k]] over the CSR row of river r (scipy's csr @ dense: the terms
// in stored order, multiply and add rounded separately), then cumulative -> incremental (row t minus row t - 1, row 0
// kept), clip at zero, NaN -> 0, times the catchment area.  One lane per river and a chunk of kRunoffRows time steps:
// with the runoff stored point-major (stride_t = 1) every gathered point is one contiguous run of the chunk's rows.
constexpr int kRunoffRows = 16;

// VEC: the block is point-major with rows padded to a multiple of kRunoffRows elements (stride_t = 1,
// stride_p % kRunoffRows == 0, 16-byte aligned base), so a chunk of one grid point is read as whole 16-byte vectors.
template <typename RT, bool VEC>
__global__ __launch_bounds__(kBlock) void k_runoff_to_qlateral(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                               const double *__restrict__ weights, const RT *__restrict__ runoff,
                                                               int64_t stride_t, int64_t stride_p, const double *__restrict__ area,
                                                               int flags, double *__restrict__ out, int64_t n_rivers, int64_t T)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.y * kRunoffRows;
    if (r >= n_rivers) return;
    const int nt = (int)min((int64_t)kRunoffRows, T - t0);
    const bool cumulative = flags & RR_RUNOFF_CUMULATIVE, force_positive = flags & RR_RUNOFF_FORCE_POSITIVE,
               keep_nan = flags & RR_RUNOFF_KEEP_NAN;
    double acc[kRunoffRows + 1];        // slot 0: row t0 - 1 (cumulative input only)
#pragma unroll
    for (int j = 0; j <= kRunoffRows; ++j) acc[j] = 0.0;
    const bool need_prev = cumulative && t0 > 0;
    for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) {
        const double w = weights[k];
        const RT *src = runoff + (int64_t)indices[k] * stride_p + (t0 - 1) * stride_t;
        if (VEC) {
            constexpr int VL = 16 / (int)sizeof(RT);      // elements per 16-byte load
            struct alignas(16) Vec { RT v[VL]; };
            const Vec *vsrc = reinterpret_cast<const Vec *>(src + 1);
            if (need_prev) acc[0] = __dadd_rn(acc[0], __dmul_rn(w, (double)src[0]));
#pragma unroll
            for (int q = 0; q < kRunoffRows / VL; ++q) {
                const Vec x = vsrc[q];                    // rows past T lie in the row padding: read, never used
#pragma unroll
                for (int e = 0; e < VL; ++e) acc[1 + q * VL + e] = __dadd_rn(acc[1 + q * VL + e], __dmul_rn(w, (double)x.v[e]));
            }
        } else {
#pragma unroll
            for (int j = 0; j <= kRunoffRows; ++j) {
                if (j == 0 ? need_prev : j <= nt) acc[j] = __dadd_rn(acc[j], __dmul_rn(w, (double)src[(int64_t)j * stride_t]));
            }
        }
    }
    const double a = area ? area[r] : 1.0;
#pragma unroll
    for (int j = 1; j <= kRunoffRows; ++j) {
        if (j > nt) break;
        double v = (cumulative && t0 + j - 1 > 0) ? acc[j] - acc[j - 1] : acc[j];
        if (force_positive) v = v < 0.0 ? 0.0 : v;      // np.clip leaves NaN alone, as does this comparison
        if (v != v && !keep_nan) v = 0.0;
        out[(t0 + j - 1) * n_rivers + r] = area ? v * a : v;
    }
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// plan object
// ------------------------------------------------------------------------------------------------

enum class Mode { Rapid, Muskingum, Unit };

// Where the (time, reach) rows in params order come from / go to.
struct Rows {
    const double *dev_in = nullptr;   // device array, rows_in rows
    const double *host_in = nullptr;  // host array, T rows
    int64_t rows_in = 0;
    double *dev_out = nullptr;
    double *host_out = nullptr;
    int64_t rows_out = 0;
};

// One routing call in flight: rows enter (permutation in), ticks run, finished rows leave (permutation out).
// route_core() runs a session start to finish; the rr_stream_* entry points keep it open between calls so the
// lag pipeline is never drained while forcing or boundary series arrive in chunks (multi-GPU).
struct Session {
    bool open = false;
    Mode mode = Mode::Rapid;
    int64_t T = 0, nsub = 1, total = 0, total_ticks = 0;
    Rows io;
    hipStream_t stream = nullptr;
    bool direct = false, has_in = true;
    int64_t ring_rows = 0;
    int64_t rows_loaded = 0, rows_stored = 0, tau = 0;
    const double *ghost_series = nullptr;
    double *export_series = nullptr;
    TickArgs a{};
    bool wave = false;            // time-tiled k_wave instead of per-tick k_tick
    bool rec = false;             // record-mode ring + one-pass permutation (k_wave_rec, k_rec_in, k_rec_out)
    int64_t rec_chunks = 0, in_batches = 0, n_in_batches = 0, out_batches = 0, n_out_batches = 0;
    int64_t diag = 0, n_diags = 0, n_chunks = 0;
    WaveArgs wa{};
    bool bracket_open = false;
    int64_t bracket_reaches = 0;
    size_t max_samples = 0;
};

struct rr_plan {
    rr::HostPlan h;
    int device = RR_DEVICE_NONE;
    bool coeffs_set = false, has_c4 = false;
    int64_t chunk_rows = 16, sample_every = 0;

    int32_t *d_child_ptr = nullptr, *d_lag = nullptr, *d_perm = nullptr, *d_inv = nullptr, *d_inner_pos = nullptr;
    int32_t *d_bidx = nullptr;   // ghost / export slot of flagged positions
    uint16_t *d_hwc = nullptr;
    double *d_w = nullptr, *d_c2 = nullptr, *d_c3 = nullptr, *d_c4 = nullptr;
    double *d_x = nullptr, *d_isum = nullptr, *d_qch = nullptr;
    double *d_ring = nullptr;
    int64_t ring_cap = 0;  // doubles
    double *d_stage = nullptr;
    int64_t stage_cap = 0;
    double *d_mrows = nullptr;   // intermediate rows of the tiled permutation
    int64_t mrows_cap = 0;
    // tiled permutations: [0] params order -> engine order (pi = perm), [1] engine -> params (pi = inv)
    uint16_t *d_slot_a[2] = {nullptr, nullptr}, *d_slot_b[2] = {nullptr, nullptr};
    int32_t *d_m_index[2] = {nullptr, nullptr};
    int64_t perm_rows_per_block = 2;

    // time-tiled routing (k_wave)
    bool wave_enabled = true, wave_forced = false, wave_now = false, weights_uniform = false;
    int wave_threads = 512, wave_ppt = 4, wave_hpt = 4;
    int64_t wave_K = 16, wave_nb = 0, wave_jmax = 0, wave_lh = 0;
    double *d_c1row = nullptr, *d_sq = nullptr, *d_ss = nullptr, *d_si = nullptr, *d_hist = nullptr;
    int64_t hist_cap = 0;
    int2 *d_colmeta = nullptr;   // per params column {engine position, lag}
    double *d_c4_params = nullptr;   // c4dt in params order (scale of the record-mode permutation)
    bool rec_enabled = true;     // record mode where it applies (one sub-step, K = 16, permuted order, device rows)
    size_t dev_total_bytes = 0;

    // boundary reaches of a partitioned network
    int64_t n_ghost = 0, n_export = 0;
    int64_t ghost_min_lag = 0, export_max_lag = 0;
    int64_t wave_ghost_slack = 0, wave_export_skew = 0;   // the same bounds in the time-tiled schedule (block skew included)
    std::vector<int32_t> ghost_pos;   // engine positions of the ghosts, in the caller's ghost order

    Session ses;

    // profile of the last route call
    std::vector<hipEvent_t> ev;
    std::vector<int64_t> ev_reaches;
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    int64_t prof_launches = 0, prof_samples = 0, prof_brackets = 0, prof_reach_steps = 0;
    hipStream_t last_stream = nullptr;
};

namespace {

template <typename T>
int dev_alloc(T **p, int64_t count)
{
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess)
        return fail(RR_E_ALLOC, std::string("hipMalloc of ") + std::to_string((size_t)count * sizeof(T)) +
                                    " bytes failed: " + hipGetErrorString(e));
    return RR_OK;
}

template <typename T>
int dev_upload(T *dst, const std::vector<T> &src)
{
    if (src.empty()) return RR_OK;
    HIPCHK(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RR_OK;
}

int need_device(const rr_plan *plan)
{
    if (!plan) return fail(RR_E_INVALID, "null plan");
    if (plan->device < 0)
        return fail(RR_E_NO_DEVICE, "this plan is host-only (RR_DEVICE_NONE): the HIP engine has no CPU fallback");
    HIPCHK(hipSetDevice(plan->device));
    return RR_OK;
}

template <typename T>
int ensure_cap(T **buf, int64_t *cap, int64_t count)
{
    if (*cap >= count) return RR_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    int rc = dev_alloc(buf, count);
    if (rc) return rc;
    *cap = count;
    return RR_OK;
}

// ---- session -------------------------------------------------------------------------------------

// Which routing kernel a call of `total` sub-steps uses.  The time-tiled schedule pays (blocks * K) extra ticks of
// fill/drain per call but runs a full tick ~2.8x faster than the streaming kernel, and its partly filled launches are
// cheap; measured at 1M reaches (blocks * K = 7,824): 744 steps 35 vs 29 ms, 2,976 steps 73 vs 96 ms, break-even
// near 1,200 steps.  RR_WAVE=1 forces it, RR_WAVE=0 forbids it.
bool decide_wave(rr_plan *P, Mode mode, int64_t total)
{
    bool ok = P->wave_enabled && P->weights_uniform && P->h.n > 0 &&
              (mode != Mode::Unit || (P->n_ghost == 0 && P->n_export == 0));
    if (ok && !P->wave_forced) {
        ok = 6 * total >= P->wave_nb * P->wave_K;
    }
    P->wave_now = ok;
    return ok;
}

bool use_wave(const rr_plan *P, Mode) { return P->wave_now; }

int64_t wave_hist_rows(const rr_plan *P) { return (P->wave_jmax + 2) * P->wave_K; }


int session_begin(rr_plan *P, Mode mode, int64_t T, int64_t nsub, const Rows &io, hipStream_t stream,
                  const double *ghost_series, double *export_series)
{
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n;
    Session &S = P->ses;
    if (S.open) return fail(RR_E_STATE, "a routing call is already open on this plan (rr_stream_end it first)");
    S = Session();
    S.mode = mode; S.T = T; S.nsub = nsub; S.total = T * nsub; S.io = io; S.stream = stream;
    S.ghost_series = ghost_series; S.export_series = export_series;
    const int64_t dmax = H.depth - 1;
    S.total_ticks = S.total + dmax;
    // the kernels index ticks, rows and chunks in 32 bits (Div32)
    if (S.total_ticks >= (int64_t{1} << 31) - (int64_t{1} << 20) || io.rows_in >= (int64_t{1} << 31) || io.rows_out >= (int64_t{1} << 31))
        return fail(RR_E_UNSUPPORTED, "more than 2^31 routing ticks or rows in one call: split it into several calls");
    S.has_in = mode != Mode::Muskingum;
    const bool host_io = io.host_out != nullptr || io.host_in != nullptr;
    S.direct = H.identity && !host_io;   // engine order == params order: stream the caller's arrays
    const int64_t C = std::max<int64_t>(1, P->chunk_rows);

    P->prof_launches = P->prof_samples = P->prof_brackets = 0;
    P->prof_reach_steps = n * S.total;
    P->ev_reaches.clear();
    P->last_stream = stream;
    S.open = true;
    if (n == 0 || S.total == 0) return RR_OK;
    if (P->n_ghost > 0 && !ghost_series) { S.open = false; return fail(RR_E_INVALID, "plan has ghost reaches but no ghost series was given"); }
    if (P->n_export > 0 && !export_series) { S.open = false; return fail(RR_E_INVALID, "plan has export reaches but no export series was given"); }

    S.wave = use_wave(P, mode);
    if (S.wave) {
        S.n_chunks = (S.total_ticks + P->wave_K - 1) / P->wave_K;
        S.n_diags = S.n_chunks + P->wave_nb - 1;
    }
    // work ring in engine order: lateral rows come in, discharge rows overwrite them in place.  Rows stay until
    // the outlet-most reaches have passed them; the time-tiled schedule adds (blocks - 1) * K ticks of skew.
    const int64_t skew_ticks = dmax + (S.wave ? P->wave_nb * P->wave_K : 0);
    const int64_t lag_rows = (skew_ticks + nsub - 1) / nsub;
    S.rec = S.wave && P->rec_enabled && nsub == 1 && P->wave_K == kRec && !S.direct && !host_io && n < (int64_t{1} << 25);    // the 1024-thread shape with a 4,096-wide halo would spill
    if (S.rec) {
        // records are indexed by tick = row + lag: the live rows span (skew + depth) ticks
        // no wrap-around needed when every chunk the call can touch fits: batches * 4 + deepest lag + the look-ahead record
        const int64_t all_chunks = kRecBatch * ((T + 14) / (16 * kRecBatch) + 2) + (dmax >> 4) + 2;
        S.rec_chunks = std::min<int64_t>(all_chunks, (skew_ticks + dmax) / kRec + 32);
        const int64_t bytes = S.rec_chunks * kRec * n * (int64_t)sizeof(double);
        // up to five eighths of the card: a 1M-reach part of a partitioned 8M-reach network is deeper than a 1M-reach
        // network of its own (the trunk part holds the main stems) and needs 152 GB of the 288
        if (P->dev_total_bytes > 0 && bytes > (int64_t)(P->dev_total_bytes / 8 * 5)) S.rec = false;
    }
    S.ring_rows = S.direct ? 0 : (S.rec ? S.rec_chunks * kRec : std::min<int64_t>(T, lag_rows + 2 * C + 2));
    if (getenv("RR_VERBOSE"))
        fprintf(stderr, "rr: n=%lld T=%lld nsub=%lld wave=%d rec=%d direct=%d threads=%d ppt=%d hpt=%d K=%lld blocks=%lld lh=%lld lds=%zu ring_rows=%lld\n",
                (long long)n, (long long)T, (long long)nsub, (int)S.wave, (int)S.rec, (int)S.direct, P->wave_threads, P->wave_ppt,
                P->wave_hpt, (long long)P->wave_K, (long long)P->wave_nb, (long long)P->wave_lh,
                S.rec ? wave_rec_lds_bytes(P->wave_lh, P->wave_threads, P->wave_ppt) : (size_t)2 * P->wave_lh * sizeof(double),
                (long long)S.ring_rows);
    if (S.ring_rows > 0xFFFFFFFFLL || T > 0x7FFFFFFFLL) { S.open = false; return fail(RR_E_INVALID, "route: too many time rows"); }
    int rc = RR_OK;
    if (!S.direct) rc = ensure_cap(&P->d_ring, &P->ring_cap, S.ring_rows * n);
    if (rc == RR_E_ALLOC && S.rec) {      // no room for the record ring after all: the (smaller) row ring
        (void)hipGetLastError();
        S.rec = false;
        S.ring_rows = std::min<int64_t>(T, lag_rows + 2 * C + 2);
        rc = ensure_cap(&P->d_ring, &P->ring_cap, S.ring_rows * n);
    }
    if (!rc && !S.direct && !S.rec) rc = ensure_cap(&P->d_mrows, &P->mrows_cap, C * n);
    if (!rc && host_io) rc = ensure_cap(&P->d_stage, &P->stage_cap, C * n);
    if (rc) { S.open = false; return rc; }
    if (S.rec) {
        S.n_in_batches = S.has_in ? (T + 14) / (16 * kRecBatch) + 1 : 0;
        S.n_out_batches = (T + 16 * kRecBatch - 1) / (16 * kRecBatch);
    }

    TickArgs &a = S.a;
    a.child_ptr = P->d_child_ptr; a.lag = P->d_lag; a.w = P->d_w; a.c2 = P->d_c2; a.c3 = P->d_c3; a.c4 = P->d_c4;
    a.isum = P->d_isum; a.bidx = P->d_bidx;
    a.ghost = ghost_series; a.exports = export_series; a.n_ghost = (int32_t)P->n_ghost; a.n_export = (int32_t)P->n_export;
    a.total_substeps = S.total; a.nsub = Div32((uint32_t)nsub); a.inv_nsub = 1.0 / (double)nsub;
    if (S.direct) {
        a.in = io.dev_in; a.in_ld = n; a.in_rows = Div32((uint32_t)std::max<int64_t>(1, io.rows_in));
        a.out = io.dev_out; a.out_ld = n; a.out_rows = Div32((uint32_t)io.rows_out);
    } else {
        a.in = S.has_in ? P->d_ring : nullptr; a.in_ld = n; a.in_rows = Div32((uint32_t)S.ring_rows);
        a.out = P->d_ring; a.out_ld = n; a.out_rows = Div32((uint32_t)S.ring_rows);
    }

    if (S.wave) {
        WaveArgs &w = S.wa;
        w.child_ptr = P->d_child_ptr; w.lag = P->d_lag; w.c1row = P->d_c1row; w.c2 = P->d_c2; w.c3 = P->d_c3; w.c4 = P->d_c4;
        w.sq = P->d_sq; w.ss = P->d_ss; w.si = P->d_si; w.sqch = P->d_qch; w.hw_children = P->d_hwc; w.hist = P->d_hist; w.hist_rows = (int32_t)wave_hist_rows(P);
        w.bidx = P->d_bidx; w.ghost = ghost_series; w.exports = export_series;
        w.n_ghost = (int32_t)P->n_ghost; w.n_export = (int32_t)P->n_export;
        w.in = a.in; w.out = a.out; w.in_ld = a.in_ld; w.out_ld = a.out_ld; w.in_rows = a.in_rows; w.out_rows = a.out_rows;
        w.lh = (int32_t)P->wave_lh;
#ifdef RR_WAVE_TRACE
        w.trace = nullptr; w.trace_diag = -1;
        if (getenv("RR_WAVE_TRACE_DIAG")) {
            static long long *tbuf = nullptr;
            if (!tbuf) (void)hipMalloc(&tbuf, 8 * 8 * 4096);
            (void)hipMemset(tbuf, 0, 8 * 8 * 4096);
            w.trace = tbuf; w.trace_diag = atoll(getenv("RR_WAVE_TRACE_DIAG"));
        }
#endif
        w.rec = S.rec ? P->d_ring : nullptr; w.rec_chunks = Div32((uint32_t)S.rec_chunks);
        w.n = (int32_t)n; w.K = (int32_t)P->wave_K; w.total = S.total; w.nsub = Div32((uint32_t)nsub); w.inv_nsub = 1.0 / (double)nsub;
        if (!P->d_hist || P->hist_cap < (int64_t)w.hist_rows * n) { S.open = false; return fail(RR_E_STATE, "time-tiled routing: history ring not initialised"); }
    }
    S.max_samples = P->sample_every >= kSampleGroup ? (size_t)std::min<int64_t>(4096, S.total_ticks / P->sample_every + 1) : 0;
    if (S.wave && S.max_samples > 0) S.max_samples = (size_t)std::min<int64_t>(4096, S.n_diags / 8 + 1);     // every eighth launch
    while (P->ev.size() < 2 * S.max_samples) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        P->ev.push_back(e);
    }
    if (!P->ev_first) { HIPCHK(hipEventCreate(&P->ev_first)); HIPCHK(hipEventCreate(&P->ev_last)); }
    HIPCHK(hipEventRecord(P->ev_first, stream));
    return RR_OK;
}

// params order <-> engine order through the two-phase tiled permutation (k_perm_a / k_perm_b)
void permute_rows(rr_plan *P, int which, const RowView &src, const RowView &dst, int64_t t0, int nrows)
{
    const int64_t n = P->h.n;
    constexpr int E = kPermE;
    const int64_t tile = (int64_t)E * kPermThreads;
    const int rpb = (int)std::max<int64_t>(1, P->perm_rows_per_block);
    dim3 g((unsigned)((n + tile - 1) / tile), (unsigned)((nrows + rpb - 1) / rpb));
    const size_t lds_bytes = (size_t)tile * sizeof(double);
    hipStream_t stream = P->ses.stream;
    hipLaunchKernelGGL(k_perm_a<E>, g, dim3(kPermThreads), lds_bytes, stream, src, P->d_mrows, n,
                       (const uint16_t *)P->d_slot_a[which], (const int32_t *)P->d_m_index[which], t0, nrows, rpb);
    hipLaunchKernelGGL(k_perm_b<E>, g, dim3(kPermThreads), lds_bytes, stream, dst, (const double *)P->d_mrows, n,
                       (const uint16_t *)P->d_slot_b[which], t0, nrows, rpb);
}

int session_load_rows(rr_plan *P, int64_t r0, int64_t r1)   // params order -> ring
{
    Session &S = P->ses;
    if (S.direct || !S.has_in) return RR_OK;
    const int64_t n = P->h.n, C = std::max<int64_t>(1, P->chunk_rows);
    const int nrows = (int)(r1 - r0);
    const RowView ring_view{P->d_ring, n, 0, (uint32_t)std::max<int64_t>(1, S.ring_rows)};
    if (S.io.host_in) {
        HIPCHK(hipMemcpyAsync(P->d_stage, S.io.host_in + r0 * n, (size_t)nrows * n * sizeof(double),
                              hipMemcpyHostToDevice, S.stream));
        permute_rows(P, 0, RowView{P->d_stage, n, r0, (uint32_t)C}, ring_view, r0, nrows);
        HIPCHK(hipStreamSynchronize(S.stream));   // the stage is reused by the next chunk
    } else {
        permute_rows(P, 0, RowView{const_cast<double *>(S.io.dev_in), n, 0, (uint32_t)S.io.rows_in}, ring_view, r0, nrows);
    }
    return RR_OK;
}

int session_store_rows(rr_plan *P, int64_t r0, int64_t r1)   // ring -> params order
{
    Session &S = P->ses;
    if (S.direct) return RR_OK;
    const int64_t n = P->h.n, C = std::max<int64_t>(1, P->chunk_rows);
    const RowView ring_view{P->d_ring, n, 0, (uint32_t)std::max<int64_t>(1, S.ring_rows)};
    for (int64_t b0 = r0; b0 < r1; b0 += C) {
        const int nrows = (int)std::min<int64_t>(C, r1 - b0);
        if (S.io.host_out) {
            permute_rows(P, 1, ring_view, RowView{P->d_stage, n, b0, (uint32_t)C}, b0, nrows);
            HIPCHK(hipMemcpyAsync(S.io.host_out + b0 * n, P->d_stage, (size_t)nrows * n * sizeof(double),
                                  hipMemcpyDeviceToHost, S.stream));
            HIPCHK(hipStreamSynchronize(S.stream));
        } else {
            permute_rows(P, 1, ring_view, RowView{S.io.dev_out, n, 0, (uint32_t)S.io.rows_out}, b0, nrows);
        }
    }
    return RR_OK;
}

int session_launch_tick(rr_plan *P, int64_t tau)
{
    Session &S = P->ses;
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n, dmax = H.depth - 1;
    // active lags: tau - total < lag <= tau
    const int64_t lag_lo = std::max<int64_t>(0, tau - S.total + 1), lag_hi = std::min<int64_t>(tau, dmax);
    const int64_t p_lo = H.lag_start[lag_lo], p_hi = H.lag_start[lag_hi + 1];
    if (p_hi <= p_lo) return RR_OK;
    TickArgs &a = S.a;
    a.p_lo = (int32_t)p_lo; a.p_hi = (int32_t)p_hi; a.tau = tau;
    a.xc = P->d_x + (tau % 3) * n;
    a.xa = P->d_x + ((tau + 2) % 3) * n;
    a.xb = P->d_x + ((tau + 1) % 3) * n;
    // sampling: every sample_every-th launch opens a bracket of kSampleGroup consecutive launches, so the
    // event overhead (~5 us per pair) is amortised and the figure is comparable with rocprofv3's per-kernel time
    const int64_t phase = S.max_samples > 0 ? P->prof_launches % P->sample_every : -1;
    if (phase == 0 && !S.bracket_open && (size_t)P->prof_brackets < S.max_samples) {
        HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets], S.stream));
        S.bracket_open = true;
    }
    const dim3 g = grid1(p_hi - p_lo);
    const bool one = S.nsub == 1;
    if (S.mode == Mode::Unit) {
        UnitTickArgs ua{};
        ua.t = a; ua.hw_children = P->d_hwc; ua.qch = P->d_qch;
        if (one) hipLaunchKernelGGL(k_tick_unit<true>, g, dim3(kBlock), 0, S.stream, ua);
        else hipLaunchKernelGGL(k_tick_unit<false>, g, dim3(kBlock), 0, S.stream, ua);
    } else if (S.mode == Mode::Rapid) {
        if (one) hipLaunchKernelGGL((k_tick<true, true>), g, dim3(kBlock), 0, S.stream, a);
        else hipLaunchKernelGGL((k_tick<true, false>), g, dim3(kBlock), 0, S.stream, a);
    } else {
        if (one) hipLaunchKernelGGL((k_tick<false, true>), g, dim3(kBlock), 0, S.stream, a);
        else hipLaunchKernelGGL((k_tick<false, false>), g, dim3(kBlock), 0, S.stream, a);
    }
    if (S.bracket_open) {
        S.bracket_reaches += p_hi - p_lo;
        ++P->prof_samples;
        if (P->prof_samples % kSampleGroup == 0) {
            HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets + 1], S.stream));
            ++P->prof_brackets;
            P->ev_reaches.push_back(S.bracket_reaches);
            S.bracket_reaches = 0;
            S.bracket_open = false;
        }
    }
    ++P->prof_launches;
    return RR_OK;
}

typedef void (*wave_kernel_t)(const WaveArgs);

// Block = threads * ppt positions, halo capacity = threads * hpt.  Two shapes are built:
//   1024 threads x {1,2} positions (16 waves, latency hidden by occupancy) and
//    512 threads x {2,4} positions (8 waves, up to 256 VGPRs: latency hidden by the prefetch stages).
wave_kernel_t wave_kernel(int threads, int ppt, int hpt, bool one, bool unit)
{
#define RR_WAVE_PICK(T_, P_, H_)                                                                                    \
    (unit ? (one ? (wave_kernel_t)k_wave<T_, P_, H_, true, true> : (wave_kernel_t)k_wave<T_, P_, H_, false, true>)   \
          : (one ? (wave_kernel_t)k_wave<T_, P_, H_, true, false> : (wave_kernel_t)k_wave<T_, P_, H_, false, false>))
    if (threads == 1024) {
        if (hpt <= 2) return ppt == 1 ? RR_WAVE_PICK(1024, 1, 2) : RR_WAVE_PICK(1024, 2, 2);
        return ppt == 1 ? RR_WAVE_PICK(1024, 1, 4) : RR_WAVE_PICK(1024, 2, 4);
    }
    if (hpt <= 4) return ppt == 2 ? RR_WAVE_PICK(512, 2, 4) : RR_WAVE_PICK(512, 4, 4);
    return ppt == 2 ? RR_WAVE_PICK(512, 2, 8) : RR_WAVE_PICK(512, 4, 8);
#undef RR_WAVE_PICK
}

wave_kernel_t wave_rec_kernel(int threads, int ppt, int hpt, bool unit)
{
#define RR_REC_PICK(T_, P_, H_) (unit ? (wave_kernel_t)k_wave_rec<T_, P_, H_, true> : (wave_kernel_t)k_wave_rec<T_, P_, H_, false>)
    if (threads == 1024) {
        if (hpt <= 2) return ppt == 1 ? RR_REC_PICK(1024, 1, 2) : RR_REC_PICK(1024, 2, 2);
        return ppt == 1 ? RR_REC_PICK(1024, 1, 4) : RR_REC_PICK(1024, 2, 4);
    }
    if (hpt <= 4) return ppt == 2 ? RR_REC_PICK(512, 2, 4) : RR_REC_PICK(512, 4, 4);
    return ppt == 2 ? RR_REC_PICK(512, 2, 8) : RR_REC_PICK(512, 4, 8);
#undef RR_REC_PICK
}

// One anti-diagonal of the time-tiled schedule: tasks (block b, chunk diag - b) for every block whose chunk exists.
int session_launch_diag(rr_plan *P, int64_t d)
{
    Session &S = P->ses;
    const int64_t nb = P->wave_nb, K = P->wave_K, n = P->h.n;
    int64_t b_lo = std::max<int64_t>(0, d - (S.n_chunks - 1)), b_hi = std::min<int64_t>(nb - 1, d);
    // fill / drain: a block none of whose reaches is active during its chunk has nothing to do (lag is sorted, so the
    // idle blocks are a suffix while the pipeline fills and a prefix while it drains; history rows they leave
    // untouched are only ever read by reaches that are inactive themselves)
    const int64_t bs = (int64_t)P->wave_ppt * P->wave_threads;
    while (b_hi >= b_lo && (d - b_hi + 1) * K <= P->h.lag[b_hi * bs]) --b_hi;
    while (b_lo <= b_hi && (d - b_lo) * K >= (int64_t)P->h.lag[std::min(n, (b_lo + 1) * bs) - 1] + S.total) ++b_lo;
    if (b_hi < b_lo) { ++P->prof_launches; return RR_OK; }
    WaveArgs &w = S.wa;
    w.diag = d; w.b_first = (int32_t)b_lo;
    // every eighth launch is bracketed by HIP events, full or not (fill and drain launches run fewer blocks), so the
    // sampled average is the average rocprofv3 reports for the kernel; the reach-ticks of a sample are those of the
    // blocks it launched
    const bool sample = S.max_samples > 0 && (P->prof_launches % 8) == 0 && (size_t)P->prof_brackets < S.max_samples;
    if (sample) HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets], S.stream));
    const dim3 g((unsigned)(b_hi - b_lo + 1));
    const size_t lds_bytes = S.rec ? wave_rec_lds_bytes(P->wave_lh, P->wave_threads, P->wave_ppt) : (size_t)2 * P->wave_lh * sizeof(double);
    const dim3 t((unsigned)P->wave_threads);
    wave_kernel_t fn = S.rec ? wave_rec_kernel(P->wave_threads, P->wave_ppt, P->wave_hpt, S.mode == Mode::Unit)
                             : wave_kernel(P->wave_threads, P->wave_ppt, P->wave_hpt, S.nsub == 1, S.mode == Mode::Unit);
    hipLaunchKernelGGL(fn, g, t, lds_bytes, S.stream, w);
    if (sample) {
        HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets + 1], S.stream));
        P->ev_reaches.push_back((std::min<int64_t>(n, (b_hi + 1) * bs) - b_lo * bs) * K);
        P->prof_samples += K;
        ++P->prof_brackets;
    }
    ++P->prof_launches;
    return RR_OK;
}

void launch_rec_permute(rr_plan *P, bool in, int64_t batch)
{
    Session &S = P->ses;
    const int64_t n = P->h.n;
    RecPermArgs ra{};
    ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = n; ra.T = S.T; ra.batch = batch;
    ra.colmeta = P->d_colmeta;
    ra.scale = (in && S.mode == Mode::Rapid) ? P->d_c4_params : nullptr;
    ra.rows = in ? RowView{const_cast<double *>(S.io.dev_in), n, 0, (uint32_t)S.io.rows_in}
                 : RowView{S.io.dev_out, n, 0, (uint32_t)S.io.rows_out};
    const dim3 g((unsigned)((n + kRecCols - 1) / kRecCols));
    if (in) hipLaunchKernelGGL(k_rec_in, g, dim3(kRecThreads), 0, S.stream, ra);
    else hipLaunchKernelGGL(k_rec_out, g, dim3(kRecThreads), 0, S.stream, ra);
}

int session_advance_wave(rr_plan *P, int64_t rows_ready, int64_t ghost_ready, int64_t *export_ready)
{
    Session &S = P->ses;
    const int64_t dmax = P->h.depth - 1, C = std::max<int64_t>(1, P->chunk_rows);
    const int64_t nb = P->wave_nb, K = P->wave_K;
    const int64_t rows_per_batch = 16 * kRecBatch;
    rows_ready = std::min(rows_ready, S.T);
    for (;;) {
        bool progressed = false;
        int64_t have_rows;
        if (S.rec) {
            // one batch of 64 rows -> records; a record slot is recycled only after every row it can hold has left
            if (S.in_batches < S.n_in_batches) {
                const int64_t j = S.in_batches;
                const int64_t hi = kRecBatch * j + (dmax >> 4) + kRecBatch - 1;
                const bool rows_here = rows_ready >= std::min(rows_per_batch * (j + 1), S.T);
                const bool slot_free = hi < S.rec_chunks || S.rows_stored >= std::min(S.T, kRec * (hi - S.rec_chunks + 1));
                if (rows_here && slot_free) {
                    launch_rec_permute(P, true, j);
                    ++S.in_batches;
                    progressed = true;
                }
            }
            const int64_t loaded = S.in_batches >= S.n_in_batches ? S.T : std::max<int64_t>(0, rows_per_batch * S.in_batches - 15);
            S.rows_loaded = loaded;
            have_rows = S.has_in ? loaded : rows_ready;
        } else {
            if (S.has_in && S.rows_loaded < rows_ready) {
                const int64_t r1 = std::min(rows_ready, S.rows_loaded + C);
                int rc = session_load_rows(P, S.rows_loaded, r1);
                if (rc) return rc;
                S.rows_loaded = r1;
                progressed = true;
            }
            have_rows = S.has_in ? S.rows_loaded : rows_ready;
        }
        // diagonal d runs chunk d of block 0 (lag 0): ticks below (d+1)*K need rows below ceil((d+1)*K / nsub)
        int64_t launched = 0;
        const int64_t batch = std::max<int64_t>(1, (S.rec ? rows_per_batch : C) * S.nsub / K);
        while (S.diag < S.n_diags && launched < batch) {
            const int64_t need_ticks = std::min((S.diag + 1) * K, S.total);
            if (have_rows < S.T && have_rows * S.nsub < need_ticks) break;
            // a ghost in block b at lag L is read for sub-steps below (diag - b + 1) * K - L
            if (P->n_ghost > 0 && ghost_ready < S.total &&
                ghost_ready < std::min((S.diag + 1) * K - P->wave_ghost_slack, S.total)) break;
            int rc = session_launch_diag(P, S.diag);
            if (rc) return rc;
            ++S.diag; ++launched;
            progressed = true;
        }
        // the last block has finished chunk diag - nb; every other block is further along
        const int64_t c_done = S.diag - nb;   // chunks [0, c_done] complete everywhere
        int64_t done = 0;
        if (S.diag >= S.n_diags) done = S.T;
        else if (c_done >= 0) {
            const int64_t ticks = (c_done + 1) * K;
            done = ticks - dmax <= 0 ? 0 : (ticks - dmax) / S.nsub;
        }
        done = std::min(done, S.T);
        if (S.rec) {
            while (S.out_batches < S.n_out_batches && done >= std::min(rows_per_batch * (S.out_batches + 1), S.T)) {
                launch_rec_permute(P, false, S.out_batches);
                ++S.out_batches;
                S.rows_stored = std::min(S.T, rows_per_batch * S.out_batches);
                progressed = true;
            }
        } else if (done > S.rows_stored) {
            int rc = session_store_rows(P, S.rows_stored, done);
            if (rc) return rc;
            S.rows_stored = done;
            progressed = true;
        }
        if (!progressed) break;
    }
    if (S.diag >= S.n_diags) S.tau = S.total_ticks;
    if (export_ready) {   // an export reach in block b at lag L has produced sub-steps below (diag - b) * K - L
        const int64_t e = S.diag >= S.n_diags ? S.total : S.diag * K - P->wave_export_skew;
        *export_ready = std::max<int64_t>(0, std::min(e, S.total));
    }
    return RR_OK;
}

// Runs every tick whose inputs are present: lateral rows [0, rows_ready) and ghost sub-steps [0, ghost_ready).
// On return *export_ready = number of leading sub-steps of the export series that are final.
int session_advance(rr_plan *P, int64_t rows_ready, int64_t ghost_ready, int64_t *export_ready)
{
    Session &S = P->ses;
    if (!S.open) return fail(RR_E_STATE, "no routing call is open on this plan");
    const int64_t n = P->h.n, dmax = P->h.depth - 1, C = std::max<int64_t>(1, P->chunk_rows);
    if (export_ready) *export_ready = 0;
    if (n == 0 || S.total == 0) { if (export_ready) *export_ready = S.total; return RR_OK; }
    if (S.wave) return session_advance_wave(P, rows_ready, std::min(ghost_ready, S.total), export_ready);
    rows_ready = std::min(rows_ready, S.T);
    ghost_ready = std::min(ghost_ready, S.total);
    // a ghost at lag L is read at tick tau for sub-step tau - L: ticks below ghost_ready + min lag are safe
    const int64_t ghost_limit = (P->n_ghost == 0 || ghost_ready >= S.total) ? S.total_ticks
                                                                            : ghost_ready + P->ghost_min_lag;
    for (;;) {
        bool progressed = false;
        if (S.has_in && S.rows_loaded < rows_ready) {
            const int64_t r1 = std::min(rows_ready, S.rows_loaded + C);
            int rc = session_load_rows(P, S.rows_loaded, r1);
            if (rc) return rc;
            S.rows_loaded = r1;
            progressed = true;
        }
        const int64_t have_rows = S.has_in ? S.rows_loaded : rows_ready;
        const int64_t lat_limit = have_rows < S.T ? have_rows * S.nsub : S.total_ticks;
        // without lateral rows to pace the loop, run the ticks in chunk-sized batches so finished rows leave the ring
        const int64_t batch_limit = S.has_in ? S.total_ticks : S.tau + C * S.nsub;
        const int64_t tau_end = std::min(std::min(lat_limit, ghost_limit), std::min(batch_limit, S.total_ticks));
        for (; S.tau < tau_end; ++S.tau) {
            int rc = session_launch_tick(P, S.tau);
            if (rc) return rc;
            progressed = true;
        }
        // row t is final once the outlet-most reaches passed it: tick (t+1)*nsub - 1 + dmax
        int64_t done = S.tau >= S.total_ticks ? S.T : (S.tau - dmax < 0 ? 0 : (S.tau - dmax) / S.nsub);
        done = std::min(done, S.T);
        if (done > S.rows_stored) {
            int rc = session_store_rows(P, S.rows_stored, done);
            if (rc) return rc;
            S.rows_stored = done;
            progressed = true;
        }
        if (!progressed) break;
    }
    if (export_ready) {
        const int64_t e = S.tau >= S.total_ticks ? S.total : S.tau - P->export_max_lag;
        *export_ready = std::max<int64_t>(0, std::min(e, S.total));
    }
    return RR_OK;
}

int session_end(rr_plan *P)
{
    Session &S = P->ses;
    if (!S.open) return fail(RR_E_STATE, "no routing call is open on this plan");
    const bool complete = P->h.n == 0 || S.total == 0 || (S.tau >= S.total_ticks && S.rows_stored >= S.T);
    S.open = false;
    if (!complete) return fail(RR_E_STATE, "routing call closed before all of its time steps were routed");
    if (P->h.n == 0 || S.total == 0) return RR_OK;
    if (S.bracket_open) P->prof_samples -= P->prof_samples % kSampleGroup;   // incomplete bracket: not counted
    HIPCHK(hipEventRecord(P->ev_last, S.stream));
    HIPCHK(hipGetLastError());
#ifdef RR_WAVE_TRACE
    if (S.wave && S.wa.trace) {
        std::vector<long long> hbuf(8 * 4096);
        (void)hipStreamSynchronize(S.stream);
        (void)hipMemcpy(hbuf.data(), S.wa.trace, hbuf.size() * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("RR_WAVE_TRACE_FILE") ? getenv("RR_WAVE_TRACE_FILE") : "/tmp/wave_trace.txt", "w")) {
            for (int b = 0; b < 4096; ++b)
                if (hbuf[8 * b]) fprintf(f, "%d %lld %lld %lld %lld %lld\n", b, hbuf[8 * b], hbuf[8 * b + 1], hbuf[8 * b + 2], hbuf[8 * b + 3], hbuf[8 * b + 4]);
            fclose(f);
        }
    }
#endif
    return RR_OK;
}

// The whole call at once: what the reference's kernel boundary does.
int route_core(rr_plan *P, Mode mode, int64_t T, int64_t nsub, const Rows &io, hipStream_t stream)
{
    if (P->n_ghost > 0 || P->n_export > 0)
        return fail(RR_E_STATE, "plan has boundary reaches: use rr_stream_begin / rr_stream_advance / rr_stream_end");
    int rc = session_begin(P, mode, T, nsub, io, stream, nullptr, nullptr);
    if (rc) return rc;
    rc = session_advance(P, T, T * nsub, nullptr);
    if (rc) { P->ses.open = false; return rc; }
    return session_end(P);
}

int check_route_args(rr_plan *P, bool need_c4, int64_t T, int64_t nsub)
{
    int rc = need_device(P);
    if (rc) return rc;
    if (!P->coeffs_set) return fail(RR_E_STATE, "route called before rr_plan_set_coeffs");
    if (need_c4 && !P->has_c4) return fail(RR_E_STATE, "rr_rapid_route needs c4_dt (rr_plan_set_coeffs got NULL)");
    if (T < 0 || nsub < 1) return fail(RR_E_INVALID, "route: need num steps >= 0 and sub-steps >= 1");
    if (nsub > 0x7FFFFFFF) return fail(RR_E_INVALID, "route: too many sub-steps");
    return RR_OK;
}

int launch_state_in(rr_plan *P, Mode mode, const double *d_q, hipStream_t stream)
{
    const int64_t n = P->h.n;
    if (use_wave(P, mode)) {
        const int64_t hr = wave_hist_rows(P);
        int rc = ensure_cap(&P->d_hist, &P->hist_cap, hr * n);
        if (rc) return rc;
        hipLaunchKernelGGL(k_wave_state_in, grid1(n), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_hist, (int32_t)hr,
                           d_q, P->d_perm, P->d_child_ptr, (int32_t)n);
        return RR_OK;
    }
    hipLaunchKernelGGL(k_state_in, grid1(n), dim3(kBlock), 0, stream, P->d_x, P->d_x + n, P->d_x + 2 * n, d_q,
                       P->d_perm, (int32_t)n);
    return RR_OK;
}

void launch_state_out(rr_plan *P, Mode mode, double *d_q, int64_t total, hipStream_t stream)
{
    const int64_t n = P->h.n;
    if (use_wave(P, mode)) {
        hipLaunchKernelGGL(k_wave_state_out, grid1(n), dim3(kBlock), 0, stream, d_q, (const double *)P->d_sq, P->d_inv,
                           (int32_t)n);
        return;
    }
    hipLaunchKernelGGL(k_state_out, grid1(n), dim3(kBlock), 0, stream, d_q, (const double *)P->d_x, n,
                       P->d_lag, P->d_inv, (int32_t)n, total);
}

int rapid_like(rr_plan *P, Mode mode, double *q_t, const Rows &io, int64_t T, int64_t nsub, hipStream_t stream,
               bool q_on_host)
{
    const int64_t n = P->h.n;
    if (n == 0 || T == 0) return RR_OK;
    decide_wave(P, mode, T * nsub);
    double *d_q = q_t;
    double *tmp = nullptr;
    if (q_on_host) {
        int rc = dev_alloc(&tmp, n);
        if (rc) return rc;
        d_q = tmp;
        hipError_t e = hipMemcpyAsync(d_q, q_t, n * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return fail(RR_E_HIP, hipGetErrorString(e)); }
    }
    int rc = launch_state_in(P, mode, d_q, stream);
    if (rc == RR_OK) rc = route_core(P, mode, T, nsub, io, stream);
    if (rc == RR_OK) {
        launch_state_out(P, mode, d_q, T * nsub, stream);
        if (q_on_host) {
            hipError_t e = hipMemcpyAsync(q_t, d_q, n * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
        }
    }
    if (tmp) { (void)hipStreamSynchronize(stream); (void)hipFree(tmp); }
    return rc;
}

int unit_like(rr_plan *P, double *q_ch, double *q_full, const Rows &io, int64_t T, int64_t nsub,
              hipStream_t stream, bool q_on_host)
{
    const int64_t n = P->h.n, ni = (int64_t)P->h.inner_pos.size();
    if (n == 0 || T == 0) return RR_OK;
    decide_wave(P, Mode::Unit, T * nsub);
    double *d_qch = q_ch, *d_qfull = q_full, *tmp = nullptr;
    if (q_on_host) {
        int rc = dev_alloc(&tmp, 2 * std::max<int64_t>(ni, 1));
        if (rc) return rc;
        d_qch = tmp; d_qfull = tmp + std::max<int64_t>(ni, 1);
        hipError_t e = hipMemcpyAsync(d_qch, q_ch, ni * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_qfull, q_full, ni * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return fail(RR_E_HIP, hipGetErrorString(e)); }
    }
    hipError_t e0 = hipMemsetAsync(P->d_x, 0, 3 * n * sizeof(double), stream);
    if (e0 != hipSuccess) { if (tmp) (void)hipFree(tmp); return fail(RR_E_HIP, hipGetErrorString(e0)); }
    if (ni > 0)
        hipLaunchKernelGGL(k_unit_state_in, grid1(ni), dim3(kBlock), 0, stream, P->d_x, P->d_x + n, P->d_x + 2 * n,
                           P->d_qch, (const double *)d_qch, (const double *)d_qfull, P->d_inner_pos, (int32_t)ni);
    const bool wave = use_wave(P, Mode::Unit);
    int rc = RR_OK;
    if (wave) {   // d_x[0] now holds q_full in engine order, d_qch the channel discharge
        const int64_t hr = wave_hist_rows(P);
        rc = ensure_cap(&P->d_hist, &P->hist_cap, hr * n);
        if (rc == RR_OK)
            hipLaunchKernelGGL(k_wave_unit_state_in, grid1(n), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_qch,
                               P->d_hist, (int32_t)hr, (const double *)P->d_x, P->d_child_ptr, (int32_t)n);
    }
    if (rc == RR_OK) rc = route_core(P, Mode::Unit, T, nsub, io, stream);
    if (rc == RR_OK && ni > 0) {
        if (wave)
            hipLaunchKernelGGL(k_wave_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                               (const double *)P->d_sq, (const double *)P->d_qch, P->d_inner_pos, (int32_t)ni);
        else
        hipLaunchKernelGGL(k_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                           (const double *)P->d_x, n, (const double *)P->d_qch, P->d_lag, P->d_inner_pos,
                           (int32_t)ni, T * nsub);
        if (q_on_host) {
            hipError_t e = hipMemcpyAsync(q_ch, d_qch, ni * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(q_full, d_qfull, ni * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
        }
    }
    if (tmp) { (void)hipStreamSynchronize(stream); (void)hipFree(tmp); }
    return rc;
}

int uh_convolve_core(const double *d_kernel, double *d_state, const double *d_lateral, double *d_out, int64_t T,
                     int64_t n_ks, int64_t n, hipStream_t stream)
{
    if (T < 1 || n_ks < 1 || n < 0) return fail(RR_E_INVALID, "rr_uh_convolve: need T >= 1, n_ks >= 1, n >= 0");
    if (n == 0) return RR_OK;
    if (n_ks > 0x7FFFFFFF || T > 0x7FFFFFFFLL * 8) return fail(RR_E_INVALID, "rr_uh_convolve: sizes out of range");
    constexpr int TB = 8;
    double *d_tail = nullptr;
    int rc = dev_alloc(&d_tail, n_ks * n);
    if (rc) return rc;
    if (n_ks <= 57 && T >= 64) {
        // long series: register-resident taps + LDS window; split time only as far as needed to fill the chip
        const int64_t blocks_x = (n + kUhThreads - 1) / kUhThreads;
        int64_t segs = std::max<int64_t>(1, std::min<int64_t>(T / 256, (2048 + blocks_x - 1) / blocks_x));
        const int64_t seg_rows = ((T + segs - 1) / segs + 63) / 64 * 64;      // segments start at multiples of every NK
        segs = (T + seg_rows - 1) / seg_rows;
        dim3 g((unsigned)blocks_x, (unsigned)segs);
#define RR_UH_LAUNCH(NK_, NT_, R_, D_)                                                                             \
        do {                                                                                                       \
            const size_t lds = (size_t)NK_ * kUhThreads * sizeof(double);                                          \
            (void)hipFuncSetAttribute((const void *)k_uh_convolve_ring<NK_, NT_, R_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_uh_convolve_ring<NK_, NT_, R_, D_>), g, dim3(kUhThreads), lds, stream, d_kernel, (const double *)d_state,     \
                               d_lateral, d_out, T, (int32_t)n_ks, n, seg_rows);                                   \
        } while (0)
#define RR_UH_LAUNCH_X(...) RR_UH_LAUNCH(__VA_ARGS__)
        if (n_ks <= 5) RR_UH_LAUNCH(8, 5, 4, 2);            // NK >= NT + R - 1 window slots
        else if (n_ks <= 13) RR_UH_LAUNCH(16, 13, 4, 2);
        else if (n_ks <= 24) RR_UH_LAUNCH(32, 24, 8, 2);
        else if (n_ks <= 29) RR_UH_LAUNCH(32, 29, 4, 2);
        else if (n_ks <= 48) RR_UH_LAUNCH_X(RR_UH48);
        else RR_UH_LAUNCH(64, 57, 8, 2);
#undef RR_UH_LAUNCH
#undef RR_UH_LAUNCH_X
    } else {
        dim3 g((unsigned)((n + kBlock - 1) / kBlock), (unsigned)((T + TB - 1) / TB));
        hipLaunchKernelGGL(k_uh_convolve<TB>, g, dim3(kBlock), 0, stream, d_kernel, (const double *)d_state, d_lateral,
                           d_out, T, (int32_t)n_ks, n);
    }
    dim3 gt((unsigned)((n + kBlock - 1) / kBlock), (unsigned)n_ks);
    hipLaunchKernelGGL(k_uh_tail, gt, dim3(kBlock), 0, stream, d_kernel, (const double *)d_state, d_lateral, d_tail,
                       T, (int32_t)n_ks, n);
    hipError_t e = hipMemcpyAsync(d_state, d_tail, (size_t)n_ks * n * sizeof(double), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // d_tail is freed below
    (void)hipFree(d_tail);
    if (e != hipSuccess) return fail(RR_E_HIP, hipGetErrorString(e));
    HIPCHK(hipGetLastError());
    return RR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

extern "C" {

int rr_version(void) { return RR_VERSION_NUM; }

const char *rr_last_error(void) { return g_err.c_str(); }

int rr_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

void rr_plan_destroy(rr_plan *P)
{
    if (!P) return;
    if (P->device >= 0 && hipSetDevice(P->device) == hipSuccess) {
        void *ptrs[] = {P->d_child_ptr, P->d_lag, P->d_perm, P->d_inv, P->d_inner_pos, P->d_bidx, P->d_hwc, P->d_w, P->d_c2,
                        P->d_c1row, P->d_sq, P->d_ss, P->d_si, P->d_hist, P->d_colmeta, P->d_c4_params,
                        P->d_c3, P->d_c4, P->d_x, P->d_isum, P->d_qch, P->d_ring, P->d_stage, P->d_mrows,
                        P->d_slot_a[0], P->d_slot_a[1], P->d_slot_b[0], P->d_slot_b[1], P->d_m_index[0], P->d_m_index[1]};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        for (hipEvent_t e : P->ev) (void)hipEventDestroy(e);
        if (P->ev_first) (void)hipEventDestroy(P->ev_first);
        if (P->ev_last) (void)hipEventDestroy(P->ev_last);
    }
    delete P;
}

int rr_plan_create(int64_t n, const int32_t *csc_indptr, const int32_t *csc_indices, int device, rr_plan **out)
{
    if (!out) return fail(RR_E_INVALID, "rr_plan_create: null out");
    *out = nullptr;
    rr_plan *P = new (std::nothrow) rr_plan();
    if (!P) return fail(RR_E_ALLOC, "rr_plan_create: out of memory");
    if (const char *e = getenv("RR_CHUNK_ROWS")) P->chunk_rows = std::max(1, atoi(e));        // tuning knobs
    if (const char *e = getenv("RR_PERM_ROWS_PER_BLOCK")) P->perm_rows_per_block = std::max(1, atoi(e));
    if (const char *e = getenv("RR_WAVE")) { P->wave_enabled = atoi(e) != 0; P->wave_forced = atoi(e) == 1; }
    if (const char *e = getenv("RR_REC")) P->rec_enabled = atoi(e) != 0;
    if (const char *e = getenv("RR_WAVE_K")) P->wave_K = std::max(1, atoi(e));
    std::string err;
    int rc = rr::build_host_plan(n, csc_indptr, csc_indices, P->h, err);
    if (rc) { delete P; return fail(rc, err); }
    {   // time-tiled schedule: 2,048-position blocks (1,024 for small networks), see wave_kernel()
        int threads = 1024;
        if (const char *e = getenv("RR_WAVE_THREADS")) threads = atoi(e) == 512 ? 512 : 1024;
        int ppt = threads == 1024 ? 2 : 4;
        if (n <= 256 * 1024) ppt /= 2;
        if (const char *e = getenv("RR_WAVE_PPT")) {
            const int v = atoi(e);
            if ((threads == 1024 && (v == 1 || v == 2)) || (threads == 512 && (v == 2 || v == 4))) ppt = v;
        }
        P->wave_threads = threads;
        P->wave_ppt = ppt;
        const int64_t bs = (int64_t)ppt * threads;
        P->wave_nb = (n + bs - 1) / bs;
        int64_t jmax = 0, halo_max = 0;
        for (int64_t b = 0; b < P->wave_nb; ++b) {
            const int64_t first_up = std::min<int64_t>(P->h.child_ptr[b * bs], b * bs);
            jmax = std::max(jmax, b - first_up / bs);
            halo_max = std::max(halo_max, b * bs - first_up);
        }
        P->wave_jmax = jmax;
        const int small = threads == 1024 ? 2 : 4;     // halo registers per thread: 2,048 or 4,096 positions
        P->wave_hpt = halo_max <= (int64_t)small * threads ? small : 2 * small;
        P->wave_lh = bs + std::max<int64_t>(64, (halo_max + 63) / 64 * 64);      // own positions + the widest halo
        if (halo_max > (int64_t)2 * small * threads) P->wave_enabled = false;   // a level wider than the LDS halo: stream with k_tick
        if (P->wave_K % 2) ++P->wave_K;
    }
    if (device != RR_DEVICE_NONE) {
        int count = rr_device_count();
        if (device < 0 || device >= count) {
            delete P;
            return fail(RR_E_NO_DEVICE, "rr_plan_create: HIP device " + std::to_string(device) + " not available (" +
                                            std::to_string(count) + " visible)");
        }
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { delete P; return fail(RR_E_HIP, hipGetErrorString(e)); }
        P->device = device;
        if (!getenv("RR_WAVE_K") && n > 0) {
            // The time-tiled schedule keeps (depth + blocks * K) rows of the work ring alive.  Keep that under a
            // third of the card's memory: shrink K, and below K = 4 stream with k_tick (ring = depth rows only).
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
                P->dev_total_bytes = total_b;
                const int64_t budget_rows = (int64_t)(total_b / 3) / (n * (int64_t)sizeof(double));
                int64_t k = (budget_rows - P->h.depth - 64) / std::max<int64_t>(1, P->wave_nb);
                k = std::min<int64_t>(P->wave_K, k) & ~(int64_t)1;
                // record mode needs K = 16 and a ring of (blocks * 16 + 2 depth) ticks; it may take five eighths of
                // the card (session_begin), which a deep 1M-reach part of a partitioned network needs (152 GB)
                const int64_t rec_bytes = (((int64_t)P->wave_nb * kRec + 2 * (int64_t)P->h.depth) / kRec + 32) * kRec * n * (int64_t)sizeof(double);
                if (P->rec_enabled && P->wave_K == kRec && rec_bytes <= (int64_t)(total_b / 8 * 5)) k = kRec;
                if (k < 4) P->wave_enabled = false; else P->wave_K = k;
            }
        }
        for (int v = 0; v < 4; ++v) {   // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel
            hipError_t ea = hipFuncSetAttribute((const void *)wave_kernel(P->wave_threads, P->wave_ppt, P->wave_hpt, (v & 1) != 0, (v & 2) != 0),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)(2 * P->wave_lh * sizeof(double)));
            if (ea != hipSuccess) P->wave_enabled = false;
        }
        // record mode needs its (larger) LDS image to fit the CU: 160 KB minus nothing else resident
        const size_t rec_lds = wave_rec_lds_bytes(P->wave_lh, P->wave_threads, P->wave_ppt);
        if (rec_lds > 160 * 1024) P->rec_enabled = false;
        for (int v = 0; v < 2 && P->rec_enabled; ++v)
            if (hipFuncSetAttribute((const void *)wave_rec_kernel(P->wave_threads, P->wave_ppt, P->wave_hpt, v != 0),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rec_lds) != hipSuccess) {
                (void)hipGetLastError();
                P->rec_enabled = false;
            }
        const rr::HostPlan &H = P->h;
        const int64_t ni = (int64_t)H.inner_pos.size();
        rc = dev_alloc(&P->d_child_ptr, n + 1);
        if (!rc) rc = dev_alloc(&P->d_lag, n);
        if (!rc) rc = dev_alloc(&P->d_perm, n);
        if (!rc) rc = dev_alloc(&P->d_inv, n);
        if (!rc) rc = dev_alloc(&P->d_inner_pos, ni);
        if (!rc) rc = dev_alloc(&P->d_hwc, n);
        if (!rc) rc = dev_alloc(&P->d_bidx, n);
        if (!rc) rc = dev_alloc(&P->d_c4_params, n);
        if (!rc) rc = dev_alloc(&P->d_colmeta, n);
        if (!rc) {
            std::vector<int2> cm(n);
            for (int64_t i = 0; i < n; ++i) cm[i] = make_int2(H.inv[i], H.lag[H.inv[i]]);
            rc = dev_upload(P->d_colmeta, cm);
        }
        if (!rc) rc = dev_alloc(&P->d_c1row, n);
        if (!rc) rc = dev_alloc(&P->d_sq, n);
        if (!rc) rc = dev_alloc(&P->d_ss, n);
        if (!rc) rc = dev_alloc(&P->d_si, n);
        if (!rc) rc = dev_alloc(&P->d_w, n);
        if (!rc) rc = dev_alloc(&P->d_c2, n);
        if (!rc) rc = dev_alloc(&P->d_c3, n);
        if (!rc) rc = dev_alloc(&P->d_c4, n);
        if (!rc) rc = dev_alloc(&P->d_x, 3 * n);
        if (!rc) rc = dev_alloc(&P->d_isum, n);
        if (!rc) rc = dev_alloc(&P->d_qch, n);
        if (!rc) rc = dev_upload(P->d_child_ptr, H.child_ptr);
        if (!rc) rc = dev_upload(P->d_lag, H.lag);
        if (!rc) rc = dev_upload(P->d_perm, H.perm);
        if (!rc) rc = dev_upload(P->d_inv, H.inv);
        if (!rc) rc = dev_upload(P->d_inner_pos, H.inner_pos);
        if (!rc) rc = dev_upload(P->d_hwc, H.hw_children);
        if (!rc) {
            const int32_t *pis[2] = {H.perm.data(), H.inv.data()};
            for (int w = 0; w < 2 && !rc; ++w) {
                rr::TiledPermutation tp;
                rr::build_tiled_permutation(pis[w], n, kPermE * kPermThreads, tp);
                rc = dev_alloc(&P->d_slot_a[w], n);
                if (!rc) rc = dev_alloc(&P->d_slot_b[w], n);
                if (!rc) rc = dev_alloc(&P->d_m_index[w], n);
                if (!rc) rc = dev_upload(P->d_slot_a[w], tp.slot_a);
                if (!rc) rc = dev_upload(P->d_slot_b[w], tp.slot_b);
                if (!rc) rc = dev_upload(P->d_m_index[w], tp.m_index);
            }
        }
        if (rc) { rr_plan_destroy(P); return rc; }
    }
    *out = P;
    return RR_OK;
}

int rr_plan_info(const rr_plan *P, int64_t info[8])
{
    if (!P || !info) return fail(RR_E_INVALID, "rr_plan_info: null argument");
    info[0] = P->h.n; info[1] = P->h.n_edges; info[2] = P->h.depth; info[3] = P->h.widest_level;
    info[4] = P->h.n_headwaters; info[5] = P->h.n_outlets; info[6] = P->h.identity ? 1 : 0; info[7] = P->device;
    return RR_OK;
}

int rr_plan_layout(const rr_plan *P, int32_t *perm, int32_t *lag, int32_t *child_ptr)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_layout: null plan");
    const size_t n = (size_t)P->h.n;
    if (perm && n) std::memcpy(perm, P->h.perm.data(), n * sizeof(int32_t));
    if (lag && n) std::memcpy(lag, P->h.lag.data(), n * sizeof(int32_t));
    if (child_ptr) std::memcpy(child_ptr, P->h.child_ptr.data(), (n + 1) * sizeof(int32_t));
    return RR_OK;
}

int rr_plan_set_options(rr_plan *P, int64_t rows_per_chunk, int64_t sample_every)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_set_options: null plan");
    if (rows_per_chunk > 0) P->chunk_rows = rows_per_chunk;
    if (sample_every >= 0) P->sample_every = sample_every;
    return RR_OK;
}

int rr_plan_set_coeffs(rr_plan *P, const double *lhs_off_data, const double *c2, const double *c3, const double *c4_dt)
{
    int rc = need_device(P);
    if (rc) return rc;
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n;
    if (n > 0 && (!c2 || !c3 || (H.n_edges > 0 && !lhs_off_data)))
        return fail(RR_E_INVALID, "rr_plan_set_coeffs: null coefficient array");
    std::vector<double> w(n), a2(n), a3(n), a4(n, 0.0);
    for (int64_t p = 0; p < n; ++p) {
        const int32_t i = H.perm[p];
        const int32_t e = H.edge_of[i];
        w[p] = e >= 0 ? -lhs_off_data[e] : 0.0;
        a2[p] = c2[i];
        a3[p] = c3[i];
        if (c4_dt) a4[p] = c4_dt[i];
    }
    // the time-tiled kernel keeps ONE upstream weight per reach; per-edge weights (never produced by the
    // reference's callers) fall back to the streaming kernel
    std::vector<double> c1row(n, 0.0);
    bool uniform = true;
    for (int64_t p = 0; p < n; ++p) {
        const int32_t u0 = H.child_ptr[p], u1 = H.child_ptr[p + 1];
        if (u1 > u0) c1row[p] = w[u0];
        for (int32_t u = u0 + 1; u < u1; ++u) if (w[u] != w[u0]) uniform = false;
    }
    P->weights_uniform = uniform;
    rc = dev_upload(P->d_w, w);
    if (!rc) rc = dev_upload(P->d_c1row, c1row);
    if (!rc) rc = dev_upload(P->d_c2, a2);
    if (!rc) rc = dev_upload(P->d_c3, a3);
    if (!rc) rc = dev_upload(P->d_c4, a4);
    if (!rc && c4_dt && n > 0) {
        hipError_t e = hipMemcpy(P->d_c4_params, c4_dt, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
    }
    if (rc) return rc;
    P->coeffs_set = true;
    P->has_c4 = c4_dt != nullptr;
    return RR_OK;
}

int rr_plan_profile(rr_plan *P, double prof[10])
{
    if (!P || !prof) return fail(RR_E_INVALID, "rr_plan_profile: null argument");
    for (int k = 0; k < 10; ++k) prof[k] = 0.0;
    prof[0] = (double)P->prof_launches;
    prof[8] = (double)P->prof_brackets;
    prof[9] = P->ses.wave ? (double)P->wave_K : 1.0;
    prof[7] = (double)P->prof_reach_steps;
    if (P->device < 0 || !P->ev_first || P->prof_launches == 0) return RR_OK;
    HIPCHK(hipSetDevice(P->device));
    HIPCHK(hipEventSynchronize(P->ev_last));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, P->ev_first, P->ev_last));
    prof[6] = ms;
    double sum = 0, mn = 1e300, mx = 0, reaches = 0;
    const double per = P->prof_brackets ? (double)P->prof_samples / (double)P->prof_brackets : 1.0;   // ticks per bracket
    for (int64_t k = 0; k < P->prof_brackets; ++k) {
        HIPCHK(hipEventElapsedTime(&ms, P->ev[2 * k], P->ev[2 * k + 1]));
        sum += ms; mn = std::min<double>(mn, ms / per); mx = std::max<double>(mx, ms / per);
        reaches += (double)P->ev_reaches[k];
    }
    prof[1] = (double)P->prof_samples; prof[2] = sum; prof[3] = P->prof_samples ? mn : 0.0; prof[4] = mx;
    prof[5] = reaches;
    return RR_OK;
}

// ---- partitioned networks: boundary reaches + streaming calls ----

int rr_plan_set_boundary(rr_plan *P, int64_t n_ghost, const int64_t *ghost_reaches, int64_t n_export,
                         const int64_t *export_reaches)
{
    int rc = need_device(P);
    if (rc) return rc;
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n;
    if (n_ghost < 0 || n_export < 0 || (n_ghost > 0 && !ghost_reaches) || (n_export > 0 && !export_reaches))
        return fail(RR_E_INVALID, "rr_plan_set_boundary: bad argument");
    if (P->ses.open) return fail(RR_E_STATE, "rr_plan_set_boundary: a routing call is open");
    std::vector<int32_t> lag(H.lag), bidx(n, 0);
    P->ghost_pos.assign(n_ghost, 0);
    int64_t gmin = H.depth, emax = 0;
    for (int64_t g = 0; g < n_ghost; ++g) {
        const int64_t i = ghost_reaches[g];
        if (i < 0 || i >= n) return fail(RR_E_INVALID, "rr_plan_set_boundary: ghost reach out of range");
        const int32_t p = H.inv[i];
        if (H.child_ptr[p + 1] != H.child_ptr[p] || (lag[p] & kGhostBit))
            return fail(RR_E_INVALID, "rr_plan_set_boundary: a ghost reach must be a headwater of this part, listed once");
        lag[p] |= kGhostBit;
        bidx[p] = (int32_t)g;
        P->ghost_pos[g] = p;
        gmin = std::min<int64_t>(gmin, H.lag[p]);
    }
    for (int64_t e = 0; e < n_export; ++e) {
        const int64_t i = export_reaches[e];
        if (i < 0 || i >= n) return fail(RR_E_INVALID, "rr_plan_set_boundary: export reach out of range");
        const int32_t p = H.inv[i];
        if (lag[p] & (kGhostBit | kExportBit))
            return fail(RR_E_INVALID, "rr_plan_set_boundary: an export reach must be a real reach, listed once");
        lag[p] |= kExportBit;
        bidx[p] = (int32_t)e;
        emax = std::max<int64_t>(emax, H.lag[p]);
    }
    rc = dev_upload(P->d_lag, lag);
    if (!rc) rc = dev_upload(P->d_bidx, bidx);
    if (rc) return rc;
    P->n_ghost = n_ghost; P->n_export = n_export;
    P->ghost_min_lag = n_ghost ? gmin : 0;
    P->export_max_lag = n_export ? emax : 0;
    {
        const int64_t bs = (int64_t)P->wave_ppt * P->wave_threads, K = P->wave_K;
        int64_t slack = (int64_t)H.depth + P->wave_nb * K, skew = 0;
        for (int64_t g = 0; g < n_ghost; ++g) { const int64_t p = H.inv[ghost_reaches[g]]; slack = std::min(slack, (p / bs) * K + H.lag[p]); }
        for (int64_t e = 0; e < n_export; ++e) { const int64_t p = H.inv[export_reaches[e]]; skew = std::max(skew, (p / bs) * K + H.lag[p]); }
        P->wave_ghost_slack = n_ghost ? slack : 0;
        P->wave_export_skew = n_export ? skew : 0;
    }
    return RR_OK;
}

int rr_stream_begin(rr_plan *P, int has_lateral, const double *q_t, const double *lateral, int64_t lat_rows,
                    double *discharge, int64_t out_rows, int64_t T, int64_t nsub, const double *ghost_series,
                    double *export_series, void *stream)
{
    int rc = check_route_args(P, has_lateral != 0, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!q_t || !discharge || out_rows < 1 || (has_lateral && (!lateral || lat_rows < 1))))
        return fail(RR_E_INVALID, "rr_stream_begin: null array or empty row count");
    Rows io; io.dev_in = has_lateral ? lateral : nullptr; io.rows_in = lat_rows; io.dev_out = discharge; io.rows_out = out_rows;
    const Mode mode = has_lateral ? Mode::Rapid : Mode::Muskingum;
    if (P->ses.open) return fail(RR_E_STATE, "a routing call is already open on this plan (rr_stream_end it first)");
    decide_wave(P, mode, T * nsub);
    if (P->h.n > 0 && T > 0) {
        rc = launch_state_in(P, mode, q_t, (hipStream_t)stream);
        if (rc) return rc;
    }
    return session_begin(P, mode, T, nsub, io, (hipStream_t)stream, ghost_series, export_series);
}

int rr_stream_advance(rr_plan *P, int64_t lateral_rows_ready, int64_t ghost_substeps_ready, int64_t *export_substeps_ready)
{
    int rc = need_device(P);
    if (rc) return rc;
    return session_advance(P, lateral_rows_ready, ghost_substeps_ready, export_substeps_ready);
}

int rr_stream_end(rr_plan *P, double *q_t)
{
    int rc = need_device(P);
    if (rc) return rc;
    const int64_t total = P->ses.total;
    hipStream_t stream = P->ses.stream;
    rc = session_end(P);
    if (rc) return rc;
    if (P->h.n > 0 && total > 0 && q_t) launch_state_out(P, P->ses.mode, q_t, total, stream);
    return RR_OK;
}

int rr_partition_forest(int64_t n, const int32_t *csc_indptr, const int32_t *csc_indices, int32_t n_parts,
                        int32_t *part_of, int64_t *part_sizes)
{
    if (!part_of || n_parts < 1) return fail(RR_E_INVALID, "rr_partition_forest: bad argument");
    rr::HostPlan H;
    std::string err;
    int rc = rr::build_host_plan(n, csc_indptr, csc_indices, H, err);
    if (rc) return fail(rc, err);
    std::vector<int32_t> down(n, -1);
    for (int64_t i = 0; i < n; ++i) if (H.edge_of[i] >= 0) down[i] = csc_indices[H.edge_of[i]];
    std::vector<int64_t> sizes;
    rr::partition_forest(down, n_parts, part_of, sizes);
    if (part_sizes) for (int32_t k = 0; k < n_parts; ++k) part_sizes[k] = k < (int32_t)sizes.size() ? sizes[k] : 0;
    return RR_OK;
}

// ---- device-pointer entry points ----

int rr_rapid_route_dev(rr_plan *P, double *q_t, const double *qlateral, int64_t ql_rows, double *discharge,
                       int64_t out_rows, int64_t T, int64_t nsub, void *stream)
{
    int rc = check_route_args(P, true, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!q_t || !qlateral || !discharge || ql_rows < 1 || out_rows < 1))
        return fail(RR_E_INVALID, "rr_rapid_route_dev: null array or empty row count");
    Rows io; io.dev_in = qlateral; io.rows_in = ql_rows; io.dev_out = discharge; io.rows_out = out_rows;
    return rapid_like(P, Mode::Rapid, q_t, io, T, nsub, (hipStream_t)stream, false);
}

int rr_muskingum_route_dev(rr_plan *P, double *q_t, double *discharge, int64_t out_rows, int64_t n_out,
                           int64_t n_per_out, void *stream)
{
    int rc = check_route_args(P, false, n_out, n_per_out);
    if (rc) return rc;
    if (P->h.n > 0 && n_out > 0 && (!q_t || !discharge || out_rows < 1))
        return fail(RR_E_INVALID, "rr_muskingum_route_dev: null array or empty row count");
    Rows io; io.dev_out = discharge; io.rows_out = out_rows;
    return rapid_like(P, Mode::Muskingum, q_t, io, n_out, n_per_out, (hipStream_t)stream, false);
}

int rr_unit_route_dev(rr_plan *P, double *q_ch, double *q_full, const double *conv, int64_t conv_rows,
                      double *discharge, int64_t out_rows, int64_t T, int64_t nsub, void *stream)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!conv || !discharge || conv_rows < 1 || out_rows < 1 ||
                                (!P->h.inner_pos.empty() && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_unit_route_dev: null array or empty row count");
    Rows io; io.dev_in = conv; io.rows_in = conv_rows; io.dev_out = discharge; io.rows_out = out_rows;
    return unit_like(P, q_ch, q_full, io, T, nsub, (hipStream_t)stream, false);
}

int rr_uh_convolve_dev(int device, const double *kernel, double *state, const double *lateral, double *out,
                       int64_t T, int64_t n_ks, int64_t n, void *stream)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_uh_convolve: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (n > 0 && (!kernel || !state || !lateral || !out)) return fail(RR_E_INVALID, "rr_uh_convolve: null array");
    return uh_convolve_core(kernel, state, lateral, out, T, n_ks, n, (hipStream_t)stream);
}

// ---- host-pointer entry points (the reference's kernel boundary) ----

int rr_rapid_route(rr_plan *P, double *q_t, const double *qlateral, double *discharge, int64_t T, int64_t nsub)
{
    int rc = check_route_args(P, true, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!q_t || !qlateral || !discharge)) return fail(RR_E_INVALID, "rr_rapid_route: null array");
    Rows io; io.host_in = qlateral; io.host_out = discharge;
    rc = rapid_like(P, Mode::Rapid, q_t, io, T, nsub, nullptr, true);
    if (rc == RR_OK) HIPCHK(hipStreamSynchronize(nullptr));
    return rc;
}

int rr_muskingum_route(rr_plan *P, double *q_t, double *discharge, int64_t n_out, int64_t n_per_out)
{
    int rc = check_route_args(P, false, n_out, n_per_out);
    if (rc) return rc;
    if (P->h.n > 0 && n_out > 0 && (!q_t || !discharge)) return fail(RR_E_INVALID, "rr_muskingum_route: null array");
    Rows io; io.host_out = discharge;
    rc = rapid_like(P, Mode::Muskingum, q_t, io, n_out, n_per_out, nullptr, true);
    if (rc == RR_OK) HIPCHK(hipStreamSynchronize(nullptr));
    return rc;
}

int rr_unit_route(rr_plan *P, double *q_ch, double *q_full, const double *conv, double *discharge, int64_t T,
                  int64_t nsub)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!conv || !discharge || (!P->h.inner_pos.empty() && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_unit_route: null array");
    Rows io; io.host_in = conv; io.host_out = discharge;
    rc = unit_like(P, q_ch, q_full, io, T, nsub, nullptr, true);
    if (rc == RR_OK) HIPCHK(hipStreamSynchronize(nullptr));
    return rc;
}

int rr_uh_convolve(int device, const double *kernel, double *state, const double *lateral, double *out, int64_t T,
                   int64_t n_ks, int64_t n)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_uh_convolve: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (T < 1 || n_ks < 1 || n < 0) return fail(RR_E_INVALID, "rr_uh_convolve: need T >= 1, n_ks >= 1, n >= 0");
    if (n == 0) return RR_OK;
    if (!kernel || !state || !lateral || !out) return fail(RR_E_INVALID, "rr_uh_convolve: null array");
    double *d_k = nullptr, *d_s = nullptr, *d_l = nullptr, *d_o = nullptr;
    int rc = dev_alloc(&d_k, n_ks * n);
    if (!rc) rc = dev_alloc(&d_s, n_ks * n);
    if (!rc) rc = dev_alloc(&d_l, T * n);
    if (!rc) rc = dev_alloc(&d_o, T * n);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpy(d_k, kernel, (size_t)n_ks * n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_s, state, (size_t)n_ks * n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_l, lateral, (size_t)T * n * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
    }
    if (!rc) rc = uh_convolve_core(d_k, d_s, d_l, d_o, T, n_ks, n, nullptr);
    if (!rc) {
        e = hipMemcpy(out, d_o, (size_t)T * n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(state, d_s, (size_t)n_ks * n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
    }
    for (double *p : {d_k, d_s, d_l, d_o}) if (p) (void)hipFree(p);
    return rc;
}

int rr_resample_cast_dev(int device, const double *discharge, int64_t num_rows, int64_t n, int64_t factor, float *out,
                         void *stream)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_resample_cast_dev: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (num_rows < 0 || n < 0 || factor < 1 || factor > 0x7FFFFFFF || (factor > 0 && num_rows % factor != 0))
        return fail(RR_E_INVALID, "rr_resample_cast_dev: rows must be a multiple of factor >= 1");
    const int64_t out_rows = num_rows / factor;
    if (out_rows == 0 || n == 0) return RR_OK;
    if (!discharge || !out) return fail(RR_E_INVALID, "rr_resample_cast_dev: null array");
    if (out_rows > 65535) {   // grid.y limit: go in slabs
        for (int64_t o0 = 0; o0 < out_rows; o0 += 65535) {
            const int64_t rows = std::min<int64_t>(65535, out_rows - o0);
            dim3 g((unsigned)((n + kBlock - 1) / kBlock), (unsigned)rows);
            hipLaunchKernelGGL(k_resample_cast, g, dim3(kBlock), 0, (hipStream_t)stream, discharge + o0 * factor * n,
                               out + o0 * n, n, rows, (int32_t)factor);
        }
    } else {
        dim3 g((unsigned)((n + kBlock - 1) / kBlock), (unsigned)out_rows);
        hipLaunchKernelGGL(k_resample_cast, g, dim3(kBlock), 0, (hipStream_t)stream, discharge, out, n, out_rows,
                           (int32_t)factor);
    }
    HIPCHK(hipGetLastError());
    return RR_OK;
}

int rr_runoff_to_qlateral_dev(int device, int64_t n_rivers, int64_t n_points, int64_t T, const int32_t *indptr,
                              const int32_t *indices, const double *weights, const void *runoff, int runoff_is_f32,
                              int64_t stride_t, int64_t stride_p, const double *area, int flags, double *qlateral, void *stream)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_runoff_to_qlateral_dev: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (n_rivers < 0 || n_points < 0 || T < 0 || n_rivers > 0x7FFFFFFFLL || n_points > 0x7FFFFFFFLL)
        return fail(RR_E_INVALID, "rr_runoff_to_qlateral_dev: sizes out of range");
    if (n_rivers == 0 || T == 0) return RR_OK;
    if (!indptr || !indices || !weights || !runoff || !qlateral) return fail(RR_E_INVALID, "rr_runoff_to_qlateral_dev: null array");
    const int64_t chunks = (T + kRunoffRows - 1) / kRunoffRows;
    if (chunks > 65535) return fail(RR_E_UNSUPPORTED, "rr_runoff_to_qlateral_dev: more than 1,048,560 time steps in one call");
    dim3 g((unsigned)((n_rivers + kBlock - 1) / kBlock), (unsigned)chunks);
    const bool vec = stride_t == 1 && stride_p % kRunoffRows == 0 && ((uintptr_t)runoff & 15) == 0;
#define RR_RUNOFF_LAUNCH(RT_, VEC_)                                                                                \
    hipLaunchKernelGGL((k_runoff_to_qlateral<RT_, VEC_>), g, dim3(kBlock), 0, (hipStream_t)stream, indptr, indices, weights,  \
                       (const RT_ *)runoff, stride_t, stride_p, area, flags, qlateral, n_rivers, T)
    if (runoff_is_f32) { if (vec) RR_RUNOFF_LAUNCH(float, true); else RR_RUNOFF_LAUNCH(float, false); }
    else { if (vec) RR_RUNOFF_LAUNCH(double, true); else RR_RUNOFF_LAUNCH(double, false); }
#undef RR_RUNOFF_LAUNCH
    HIPCHK(hipGetLastError());
    return RR_OK;
}

int rr_runoff_to_qlateral(int device, int64_t n_rivers, int64_t n_points, int64_t T, const int32_t *indptr,
                          const int32_t *indices, const double *weights, const void *runoff, int runoff_is_f32,
                          int64_t stride_t, int64_t stride_p, const double *area, int flags, double *qlateral)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_runoff_to_qlateral: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (n_rivers < 0 || n_points < 0 || T < 0) return fail(RR_E_INVALID, "rr_runoff_to_qlateral: negative size");
    if (n_rivers == 0 || T == 0) return RR_OK;
    if (!indptr || !indices || !weights || !runoff || !qlateral) return fail(RR_E_INVALID, "rr_runoff_to_qlateral: null array");
    // the runoff block spans max over (t, p) of t * stride_t + p * stride_p elements
    if (stride_t < 0 || stride_p < 0) return fail(RR_E_INVALID, "rr_runoff_to_qlateral: negative stride");
    const int64_t nnz = indptr[n_rivers];
    // point-major blocks may pad their rows (stride_p > T): the whole padded image is copied
    const int64_t elems = stride_t == 1 && stride_p >= T ? std::max<int64_t>(1, n_points) * stride_p
                                                        : (T - 1) * stride_t + (n_points > 0 ? (n_points - 1) * stride_p : 0) + 1;
    const size_t esz = runoff_is_f32 ? sizeof(float) : sizeof(double);
    int32_t *d_indptr = nullptr, *d_indices = nullptr;
    double *d_w = nullptr, *d_area = nullptr, *d_out = nullptr;
    void *d_runoff = nullptr;
    int rc = dev_alloc(&d_indptr, n_rivers + 1);
    if (!rc) rc = dev_alloc(&d_indices, std::max<int64_t>(1, nnz));
    if (!rc) rc = dev_alloc(&d_w, std::max<int64_t>(1, nnz));
    if (!rc && area) rc = dev_alloc(&d_area, n_rivers);
    if (!rc) rc = dev_alloc(&d_out, T * n_rivers);
    if (!rc && hipMalloc(&d_runoff, (size_t)elems * esz) != hipSuccess) rc = fail(RR_E_ALLOC, "rr_runoff_to_qlateral: device allocation failed");
    auto release = [&]() {
        (void)hipFree(d_indptr); (void)hipFree(d_indices); (void)hipFree(d_w); (void)hipFree(d_area); (void)hipFree(d_out); (void)hipFree(d_runoff);
    };
    if (rc) { release(); return rc; }
    hipError_t e = hipMemcpy(d_indptr, indptr, (size_t)(n_rivers + 1) * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(d_indices, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(d_w, weights, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && area) e = hipMemcpy(d_area, area, (size_t)n_rivers * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_runoff, runoff, (size_t)elems * esz, hipMemcpyHostToDevice);
    if (e != hipSuccess) { release(); return fail(RR_E_HIP, hipGetErrorString(e)); }
    rc = rr_runoff_to_qlateral_dev(device, n_rivers, n_points, T, d_indptr, d_indices, d_w, d_runoff, runoff_is_f32, stride_t,
                                   stride_p, d_area, flags, d_out, nullptr);
    if (!rc) {
        e = hipMemcpy(qlateral, d_out, (size_t)(T * n_rivers) * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
    }
    release();
    return rc;
}

// ---- device helpers ----

int rr_dev_malloc(int device, int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(RR_E_INVALID, "rr_dev_malloc: bad argument");
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_malloc: no such HIP device");
    HIPCHK(hipSetDevice(device));
    hipError_t e = hipMalloc(out, (size_t)std::max<int64_t>(bytes, 1));
    if (e != hipSuccess) return fail(RR_E_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    return RR_OK;
}

int rr_dev_free(int device, void *ptr)
{
    if (!ptr) return RR_OK;
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_free: no such HIP device");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(ptr));
    return RR_OK;
}

int rr_dev_upload(int device, void *dst_dev, const void *src_host, int64_t bytes)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_upload: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (bytes > 0) HIPCHK(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return RR_OK;
}

int rr_dev_download(int device, void *dst_host, const void *src_dev, int64_t bytes)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_download: no such HIP device");
    HIPCHK(hipSetDevice(device));
    if (bytes > 0) HIPCHK(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return RR_OK;
}

int rr_dev_synchronize(int device)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_synchronize: no such HIP device");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return RR_OK;
}

}  // extern "C"
