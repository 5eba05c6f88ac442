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
#include <limits>
#include <string>
#include <thread>
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
constexpr int32_t kTileGhostBit = rr::kTileGhost;     // tile layout only: position mirrors a reach another tile owns
constexpr int32_t kTileExportBit = rr::kTileExport;   // tile layout only: reach is mirrored by a ghost, values also go to the export ring
constexpr int32_t kLagMask = kTileExportBit - 1;

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

// ---- time-tiled routing over subtree tiles ----
//
// k_tick streams ~88 B per reach-step because nothing survives from one tick to the next.  k_tile lets one workgroup
// advance one TILE (rr_plan.hpp: at most TH * PPT positions, closed under "upstream", ghosts mirroring the reaches other
// tiles own) by K = 16 * KC routing ticks: coefficients and the tile's discharges of the previous tick sit in LDS
// (double-buffered, one LDS-only barrier per tick), so HBM sees only the lateral read and the discharge write of every
// reach-step plus the tile's state and coefficients once per task.  Task (tile, macro-chunk m) needs (tile, m - 1) and
// the tiles its ghosts mirror at the same macro-chunk, all of which have a lower level: launch d runs the tasks
// (tile, d - level(tile)) and nothing inside a launch depends on anything else inside it.
//
// Lateral inflow and discharge travel as RECORDS indexed by tick (kRec = 16 ticks, 128 bytes):
//     rec[(tick / 16) % chunks][position][tick % 16]
// holding c4dt * lateral on the way in and the clamped discharge on the way out, in place (k_rec_in / k_rec_out move
// whole records to and from params order).  A GHOST's record slot is written by the tile that owns the mirrored reach
// (its unclamped discharge, 8 bytes per tick; a ghost has the lag of its reach, so the ticks line up), and the ghost
// receives its record like any other position and republishes it: no load, wait or branch of its own.

struct TileArgs {
    const int32_t *tile_ptr, *tile_level, *tile_lag_lo, *tile_lag_hi;
    const int32_t *lag, *cfirst, *xpos;
    const uint32_t *ccnt;
    const double *c1row, *c2, *c3;        // c1row: the (uniform) weight of a reach's upstream terms
    double *sq, *ss, *si, *sqch;          // carried state: discharge, sum of upstream discharges one tick back, interval sum, channel discharge
    const int32_t *bidx;                  // slot of an export reach in the boundary series another GPU reads (multi-GPU)
    double *exports;
    int32_t n_export;
    double *rec;                          // record ring [rec_chunks][np][16]
    Div32 rec_chunks;
#ifdef RR_WAVE_TRACE
    long long *trace; int32_t trace_diag;   // development build: per-block timestamps of one launch (profiles/microbench/wave_dbg.py)
#endif
    int32_t np, t_first, t_last, KC, diag, n_macro, total, has_lat;
    Div32 nsub;
    double inv_nsub;
};

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its global loads/stores, so record
// prefetches stay in flight across ticks (__syncthreads() would drain vmcnt every tick).
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Predicated global accesses without a branch: a raw buffer access whose byte offset is pushed past the end of the
// buffer is dropped by the bounds check (loads return zero).  Unlike `if (cond) *ptr = v` the instruction is always
// issued, so hipcc can count it in vmcnt and an in-order wait for an older prefetch does not have to assume the worst.
constexpr uint32_t kBufferFlags = 0x00020000;      // gfx9 raw buffer, 32-bit data format
constexpr uint32_t kDropAccess = 0xFFFFFFF0u;      // offset outside any buffer this file creates (< 4 GiB - 16)
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

constexpr int kRec = 16;
// Records move between HBM and their owning lanes through a per-wave LDS transpose: a lane owns a position (its record
// lives in registers), but a memory instruction in which every lane touches 16 bytes of a different record costs L2 one
// request per lane.  Through the transpose four neighbouring lanes load or store the 64 contiguous bytes of one half
// record: a quarter of the requests.
constexpr int kStageStride = 10;   // doubles per position in the staging area: 64 bytes + 16 of padding (bank spread, skip flag)
constexpr int kStageLanes = 32;    // positions transposed at a time: half a wave (2.5 KiB of staging per wave)
// Lanes of one wave exchange data through its staging area without a workgroup barrier: a wave's LDS instructions
// execute in order.  The compiler still has to be told that other lanes wrote (it would reuse earlier reads).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS in doubles: X[2][TH] | stage[waves][kStageLanes * kStageStride]: the tile's discharges of the last two ticks and the
// per-wave transpose areas.
constexpr size_t tile_lds_bytes(int threads)
{
    return (size_t)(2 * (int64_t)threads + (threads / 64) * kStageLanes * kStageStride) * sizeof(double);
}

// One task: KC record chunks of one tile, one position per thread.  R[16] is the record the ticks work on, in place
// (lateral in, discharge out); N[16] receives the NEXT chunk's record while the 16 ticks of this one run, so inside a task
// HBM traffic and tick arithmetic overlap and only the first chunk's load is exposed.  Whole 128-byte records are
// requested at once (a half record would cost the fabric a full line: measured, FETCH_SIZE 1.8x).
template <int TH, bool UNIT, bool SUB>
__global__ __launch_bounds__(TH, 4) void k_tile(const TileArgs a)      // 16 waves per CU: 1,024 / TH workgroups of 128 VGPRs
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t total = a.total, K = a.KC * kRec;
    double *stage = lds + 2 * TH + (size_t)(tid >> 6) * (kStageLanes * kStageStride);   // this wave's transpose area
    auto ring = [&](int32_t chunk) { return make_rsrc(a.rec + (int64_t)a.rec_chunks.mod((uint32_t)chunk) * a.np * kRec, (uint32_t)a.np * 128u); };   // np < 2^25

    // A workgroup takes the tiles t_last - blockIdx.x - g * gridDim.x, g = 0, 1, ... of this launch (highest level first:
    // the few tiles with ghosts start in the first round), so that the first record and the state of its NEXT tile are
    // requested while the current one still ticks.  A tile none of whose positions is active during its task has nothing
    // to do (pipeline fill and drain; a ghost has the lag of the reach it mirrors, so it is idle exactly when its owner
    // did not write its record) and is skipped.
    struct Task { int32_t tile, m, b0, b1; };
    auto select = [&](int32_t from, Task &t) {
        for (int32_t c = from; c >= a.t_first; c -= (int32_t)gridDim.x) {
            const int32_t m = a.diag - a.tile_level[c];
            if (m < 0 || m >= a.n_macro) continue;
            if (m * K >= a.tile_lag_hi[c] + total || (m + 1) * K <= a.tile_lag_lo[c]) continue;
            t.tile = c; t.m = m; t.b0 = a.tile_ptr[c]; t.b1 = a.tile_ptr[c + 1];
            return true;
        }
        return false;
    };
    Task cur;
    if (!select(a.t_last - (int32_t)blockIdx.x, cur)) return;
#ifdef RR_WAVE_TRACE
    bool trace = a.trace && a.diag == a.trace_diag && tid == 0;
    const bool trace_wg = trace;
    long long *tq = a.trace + (int64_t)cur.tile * 16;
#define RR_TRACE(i) do { if (trace) tq[i] = wall_clock64(); } while (0)
#else
#define RR_TRACE(i) do { } while (0)
#endif
    RR_TRACE(0);

    // Four lanes fetch (store) the four 16-byte pieces of one 64-byte sector: in flight a lane's N[] holds OTHER
    // positions' pieces; receive() hands them to their owners through the wave's staging area.
    double R[kRec], N[kRec];
    // load j of a record: (i = j / 2: half wave and group of 16 positions, half = j % 2: which 64-byte sector), so the two
    // sectors of a 128-byte line are requested by consecutive loads
    auto issue_load = [&](__amdgpu_buffer_rsrc_t src, int32_t b0, int32_t b1, int j, bool real) {
        const int32_t t = fresh(tid), ln = t & 63;      // addresses are rebuilt at every use, not kept in registers across the task
        const int i = j >> 1, half = j & 1;
        const int32_t pos = min(b0 + (t - ln) + (i >> 1) * kStageLanes + 16 * (i & 1) + (ln >> 2), b1 - 1);
        load_f64x2(src, real ? (uint32_t)pos * 128u + (uint32_t)(half * 64 + (ln & 3) * 16) : kDropAccess,
                   N[8 * half + 2 * i], N[8 * half + 2 * i + 1]);
    };
    auto receive = [&]() {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    reinterpret_cast<double2 *>(stage + (16 * g + (lane >> 2)) * kStageStride)[lane & 3] =
                        make_double2(N[8 * half + 2 * (2 * h + g)], N[8 * half + 2 * (2 * h + g) + 1]);
                wave_lds_fence();
                if (lane / kStageLanes == h) {
                    const double2 *src = reinterpret_cast<const double2 *>(stage + (lane % kStageLanes) * kStageStride);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const double2 v = src[j]; R[8 * half + 2 * j] = v.x; R[8 * half + 2 * j + 1] = v.y; }
                }
                wave_lds_fence();
            }
    };

    // State of a position.  A slot past the end of the tile keeps lag -1: no upstream range, never active, publishes 0.0
    // to nobody.  up: LDS slot of the first upstream value (low 16 bits), number of upstream positions (high 16 bits).
    struct State { int32_t lg, up, xp, uh; double c1, c2, c3, s_prev, qch, isum, q; };
    auto load_state = [&](const Task &t, State &s) {
        s.lg = -1; s.up = 0; s.xp = 0; s.uh = 0;
        s.c1 = s.c2 = s.c3 = s.s_prev = s.qch = s.isum = s.q = 0.0;
        const int32_t p = t.b0 + tid;
        if (p < t.b1) {
            const uint32_t cc = a.ccnt[p];
            const int32_t first_up = a.cfirst[p] - t.b0;
            s.lg = a.lag[p]; s.up = first_up | (int32_t)((cc & 0xFFFFu) << 16);
            s.xp = a.xpos[p];
            if (UNIT) { s.uh = first_up + (int32_t)(cc >> 16); s.qch = a.sqch[p]; }
            if (SUB) s.isum = a.si[p];
            s.s_prev = a.ss[p];
            s.q = a.sq[p]; s.c1 = a.c1row[p]; s.c2 = a.c2[p]; s.c3 = a.c3[p];
        }
    };
    // The first tile: state and coefficients are requested BEFORE the record: memory operations retire in order, so the
    // wait for them leaves the (much larger) record load in flight.
    State st;
    load_state(cur, st);
    __amdgpu_buffer_rsrc_t rec_cur = ring(cur.m * a.KC);
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_load(rec_cur, cur.b0, cur.b1, j, true);
    // everything but the 8 record loads has arrived (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 |
    // vmcnt[5:4] << 14); a real s_waitcnt, so hipcc knows that no state register is pending inside the tick loop
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
    RR_TRACE(1);
    receive();          // the first tile's first record: the only record load nothing overlaps
    RR_TRACE(2);
    const bool has_lat = a.has_lat != 0;   // channel-only routing: the records only carry discharge

    for (;;) {
        int32_t lg = st.lg, up = st.up, xp = st.xp, uh = st.uh, sub = 0;
        double c1 = st.c1, c2 = st.c2, c3 = st.c3, s_prev = st.s_prev, qch = st.qch, isum = st.isum;
        const int32_t b0 = cur.b0, tau_begin = cur.m * K;
        lds[(size_t)((tau_begin + 1) & 1) * TH + tid] = st.q;       // tick tau_begin reads the buffer of tick tau_begin - 1
        if (SUB && lg >= 0) {      // phase of the position's sub-step counter at the first tick of the task
            const int32_t ts0 = tau_begin - (lg & kLagMask);
            const uint32_t r = a.nsub.mod((uint32_t)(ts0 < 0 ? -ts0 : ts0));
            sub = ts0 >= 0 ? (int32_t)r : (r ? (int32_t)(a.nsub.d - r) : 0);
        }
        Task nxt;
        const bool has_next = select(cur.tile - (int32_t)gridDim.x, nxt);

        // Eight slots of the record are final: write that 64-byte sector.  Half a wave at a time parks its sectors in
        // the wave's staging area, then all 64 lanes store them, four lanes per sector.
        auto store_half = [&](__amdgpu_buffer_rsrc_t dst, int half) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (lane / kStageLanes == h) {
                    double2 *mine = reinterpret_cast<double2 *>(stage + (lane % kStageLanes) * kStageStride);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mine[j] = make_double2(R[8 * half + 2 * j], R[8 * half + 2 * j + 1]);
                    reinterpret_cast<int32_t *>(mine + 4)[0] = (lg < 0 || (lg & (kGhostBit | kTileGhostBit))) ? 1 : 0;   // not this tile's to write
                }
                wave_lds_fence();
                const int32_t t = fresh(tid), ln = t & 63;
                const uint32_t first = (uint32_t)(b0 + (t - ln) + h * kStageLanes) * 128u + (uint32_t)half * 64u;
#pragma unroll
                for (int g = 0; g < 2; ++g) {       // lanes 4i .. 4i+3: the four 16-byte pieces of position 16 g + i
                    const int pm = 16 * g + (ln >> 2), piece = ln & 3;
                    const double2 *theirs = reinterpret_cast<const double2 *>(stage + pm * kStageStride);
                    const double2 v = theirs[piece];
                    const bool skip = reinterpret_cast<const int32_t *>(theirs + 4)[0] != 0;
                    store_f64x2(dst, skip ? kDropAccess : first + (uint32_t)pm * 128u + (uint32_t)piece * 16u, v);
                }
                wave_lds_fence();
            }
        };
        auto ticks = [&](int32_t tau0, int half, __amdgpu_buffer_rsrc_t rec_next, int32_t nb0, int32_t nb1, bool more) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int s = 8 * half + s8;
                // the next record (this tile's next chunk, or the next tile's first) is requested one load per tick: a CU
                // accepts only so many requests at a time, and a wave that waits to issue its loads cannot tick
                if (half == 0) issue_load(rec_next, nb0, nb1, s8, more);
                const int32_t tau = tau0 + s;
                const double *rd = lds + (size_t)((tau + 1) & 1) * TH;
                double *wr = lds + (size_t)(tau & 1) * TH;
                const int32_t t = fresh(tid), lgk = fresh(lg), upk = fresh(up);
                const int32_t u0 = upk & 0xFFFF, u1 = u0 + (int32_t)((uint32_t)upk >> 16);
                double qk = rd[t];       // own discharge one tick back
                double s_cur = 0.0, s_hw = 0.0;
                if (UNIT) {   // headwater tributaries come first in the upstream range
                    for (int32_t u = u0; u < uh; ++u) s_hw += rd[u];
                    for (int32_t u = uh; u < u1; ++u) s_cur += rd[u];
                } else {
                    for (int32_t u = u0; u < u1; ++u) s_cur += rd[u];
                }
                const int32_t ts = tau - (lgk & kLagMask);
                if (ts >= 0 && ts < total) {
                    const double lat = has_lat ? R[s] : 0.0;
                    double outv = 0.0;
                    bool routed = false;
                    if (lgk & (kGhostBit | kTileGhostBit)) {
                        qk = R[s];        // a ghost republishes what its owner computed
                    } else if (UNIT) {
                        if (u0 == u1) {
                            qk = lat;        // headwater: discharge = lateral, the record slot already holds it (unclamped, un-averaged)
                        } else {
                            const double r = __builtin_fma(c1, s_hw + s_cur, __builtin_fma(c2, s_hw + s_prev, c3 * qch));
                            qch = r;
                            qk = r + lat;
                            outv = qk; routed = true;
                        }
                        if (lgk & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[b0 + t]] = qk;
                    } else {
                        // explicit fma: every copy of this tick must round identically (split run == joint run)
                        qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, qk, lat)));
                        outv = qk; routed = true;
                        if (lgk & kExportBit) a.exports[(int64_t)ts * a.n_export + a.bidx[b0 + t]] = qk;
                    }
                    if (routed) {
                        if (SUB) {      // mean over the sub-steps of a row, written to the slot of the row's last sub-step
                            const double acc = (sub == 0 ? 0.0 : isum) + outv;
                            isum = acc;
                            if (sub + 1 == (int32_t)a.nsub.d) { const double v = acc * a.inv_nsub; R[s] = v > 0.0 ? v : 0.0; }
                        } else {
                            R[s] = outv > 0.0 ? outv : 0.0;
                        }
                    }
                }
                if (SUB) sub = sub + 1 == (int32_t)a.nsub.d ? 0 : sub + 1;
                s_prev = s_cur;
                wr[t] = qk;
                // a reach mirrored by a ghost of another tile: 8 bytes into the ghost's record, always issued (see store_f64)
                store_f64(rec_cur, (lgk >= 0 && (lgk & kTileExportBit)) ? (uint32_t)fresh(xp) * 128u + (uint32_t)s * 8u : kDropAccess, qk);
                barrier_lds();
            }
        };

        barrier_lds();      // the buffer of tick tau_begin - 1 is in place (and every wave has left the previous tile)
        for (int32_t cc = 0; cc < a.KC; ++cc) {
            const int32_t chunk = cur.m * a.KC + cc, tau0 = chunk * kRec;
            const bool last = cc + 1 == a.KC;
            // what arrives during this chunk: the tile's next chunk, or -- in its last one -- the first chunk of the next tile
            const __amdgpu_buffer_rsrc_t rec_next = ring(last ? nxt.m * a.KC : chunk + 1);
            const int32_t nb0 = last ? nxt.b0 : b0, nb1 = last ? nxt.b1 : cur.b1;
            ticks(tau0, 0, rec_next, nb0, nb1, !last || has_next);
            if (cc == 0) RR_TRACE(3);
            store_half(rec_cur, 0);
            ticks(tau0, 1, rec_next, nb0, nb1, false);
            if (cc == 0) RR_TRACE(6);
            store_half(rec_cur, 1);
            if (cc == 0) RR_TRACE(7);
            if (last && has_next) load_state(nxt, st);      // small, and only the wait for it is exposed between two tiles
            receive();      // the record that has had 16 ticks to arrive (zeros after the last chunk of the last tile)
            if (cc == 0) RR_TRACE(8);
            rec_cur = rec_next;
        }
        RR_TRACE(12);
        if (lg >= 0) {
            const int32_t p = b0 + tid;
            a.sq[p] = lds[(size_t)((tau_begin + K - 1) & 1) * TH + tid]; a.ss[p] = s_prev;
            if (UNIT) a.sqch[p] = qch;
            if (SUB) a.si[p] = isum;
        }
        RR_TRACE(13);
#ifdef RR_WAVE_TRACE
        trace = false;      // the first tile of the workgroup only
        if (!has_next && trace_wg) tq[14] = wall_clock64();     // ... and when the workgroup leaves
#endif
        if (!has_next) break;
        cur = nxt;
    }
#undef RR_TRACE
}

// sq = q0 at every position (a ghost starts from the state of the reach it mirrors), ss = sum of the upstream q0
__global__ __launch_bounds__(kBlock) void k_tile_state_in(double *sq, double *ss, double *si, const double *q_t, const int32_t *perm,
                                                          const int32_t *cfirst, const uint32_t *ccnt, int32_t np)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= np) return;
    double s = 0.0;
    const int32_t u0 = cfirst[p], u1 = u0 + (int32_t)(ccnt[p] & 0xFFFFu);
    for (int32_t u = u0; u < u1; ++u) s += q_t[perm[u]];
    sq[p] = q_t[perm[p]]; ss[p] = s; si[p] = 0.0;
}

// UnitMuskingum state for the time-tiled kernel: published discharge = q_full on inner reaches (0 on headwaters until
// their first tick), ss = sum over the INNER tributaries only (the headwater ones come first), qch = channel discharge.
// full[i] / chan[i]: q_full / q_ch scattered to params order, zeros on headwaters (k_unit_scatter).
__global__ __launch_bounds__(kBlock) void k_tile_unit_state_in(double *sq, double *ss, double *si, double *sqch, const double *full,
                                                               const double *chan, const int32_t *perm, const int32_t *cfirst,
                                                               const uint32_t *ccnt, int32_t np)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= np) return;
    double s = 0.0;
    const uint32_t cc = ccnt[p];
    const int32_t u0 = cfirst[p] + (int32_t)(cc >> 16), u1 = cfirst[p] + (int32_t)(cc & 0xFFFFu);
    for (int32_t u = u0; u < u1; ++u) s += full[perm[u]];
    sq[p] = full[perm[p]]; ss[p] = s; si[p] = 0.0; sqch[p] = chan[perm[p]];
}

__global__ __launch_bounds__(kBlock) void k_unit_scatter(double *full, double *chan, const double *q_full, const double *q_ch,
                                                         const int32_t *inner_idx, int32_t n_inner)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    full[inner_idx[k]] = q_full[k]; chan[inner_idx[k]] = q_ch[k];
}

__global__ __launch_bounds__(kBlock) void k_tile_unit_state_out(double *q_ch, double *q_full, const double *sq, const double *sqch,
                                                                const int32_t *inner_idx, const int32_t *inv, int32_t n_inner)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    const int32_t p = inv[inner_idx[k]];
    q_full[k] = sq[p];
    q_ch[k] = sqch[p];
}

__global__ __launch_bounds__(kBlock) void k_tile_state_out(double *q_t, const double *sq, const int32_t *inv, int32_t n)
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
constexpr int kUhTailThreads = 64;

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

// Carry-over tail, IN PLACE: state[s, i] <- buf[T + s, i] for s < n_ks - 1, 0 for s = n_ks - 1 (UnitHydrograph.py:103-105),
// where buf[m] = sum_{k} kernel[k] lateral[m - k] (+ the old state[m] when m < n_ks).  One thread owns a basin and walks s
// upwards: row s is written after row T + s > s has been read, so no second buffer (and no allocation, copy or
// synchronisation inside an enqueue-only call) is needed.  NK > 0: taps and the last n_ks - 1 lateral rows sit in
// registers (static indices, n_ks <= NK); NK == 0: any n_ks, straight from memory.
template <int NK>
__global__ __launch_bounds__(kUhTailThreads) void k_uh_tail(const double *__restrict__ kernel, double *__restrict__ state,
                                                            const double *__restrict__ lateral, int64_t T, int32_t n_ks, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kUhTailThreads + threadIdx.x;
    if (i >= n) return;
    if (NK > 0) {
        double kv[NK > 0 ? NK : 1], lat[NK > 0 ? NK : 1];      // lat[j] = lateral[T - 1 - j]
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            kv[k] = k < n_ks ? kernel[(int64_t)k * n + i] : 0.0;
            lat[k] = (k < n_ks - 1 && k < T) ? lateral[(T - 1 - k) * n + i] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const int64_t m = T + s;
            double acc = (s < n_ks && m < n_ks) ? state[m * n + i] : 0.0;
#pragma unroll
            for (int k = s + 1; k < NK; ++k) acc += kv[k] * lat[k - s - 1];      // taps beyond n_ks and rows before 0 are zeros
            if (s < n_ks) state[(int64_t)s * n + i] = s == n_ks - 1 ? 0.0 : acc;
        }
    } else {
        for (int32_t s = 0; s < n_ks; ++s) {
            const int64_t m = T + s;
            double acc = m < n_ks ? state[m * n + i] : 0.0;
            for (int32_t k = s + 1; k < n_ks; ++k) {
                const int64_t tt = m - k;
                if (tt < 0) break;
                acc += kernel[(int64_t)k * n + i] * lateral[tt * n + i];
            }
            state[(int64_t)s * n + i] = s == n_ks - 1 ? 0.0 : acc;
        }
    }
}

// ---- record permutation (one pass each way), see k_tile ----
// Column i of the params-order rows is position p = inv[i] with lag L = 16 * sh + o.  Tick-row r of that column (tick-row =
// routing sub-step: runoff row r / nsub, sub-step r % nsub) is slot (r + L) % 16 of record (r + L) / 16, so the 128 tick-rows
// [128 j - o, 128 j + 128 - o) are exactly the eight records 8 j + sh .. 8 j + sh + 7.  k_rec_in reads the runoff rows
// behind the 143 tick-rows [128 j - 15, 128 j + 128) of a 32-column tile coalesced into LDS (all loads in flight before the
// first LDS write) and writes eight whole 128-byte records per column (8 lanes x 16 B per record), every sub-step slot of a
// row holding the row's lateral value; k_rec_out reads nine records per column the same way and writes the tile's rows of
// the batch coalesced: the slot of a row's LAST sub-step holds the row's mean discharge.
#ifndef RR_REC_BATCH
#define RR_REC_BATCH 8
#define RR_REC_COLS 32
#endif
#ifndef RR_REC_THREADS
#define RR_REC_THREADS 256
#endif
constexpr int kRecCols = RR_REC_COLS, kRecBatch = RR_REC_BATCH, kRecThreads = RR_REC_THREADS;
constexpr int kRecRows = 16 * kRecBatch;    // tick-rows of one batch

struct RecPermArgs {
    double *rec;
    Div32 rec_chunks;
    int64_t n, np, T, total, batch;   // T runoff rows, total = T * nsub tick-rows
    Div32 nsub;
    const int2 *colmeta;      // per params column: {position, lag}
    const double *scale;      // c4dt in PARAMS order (RapidMuskingum: the ring holds c4dt * lateral) or NULL
    RowView rows;             // params-order rows (source of k_rec_in, destination of k_rec_out)
    float *rows32;            // k_rec_out: float32 destination with `factor` rows averaged (router post-processing), or NULL
    Div32 factor;
};

constexpr int kRecTileRows = 16 * kRecBatch + 15;      // tick-rows behind one batch of records
constexpr int kRecTileLd = kRecCols + 1;

// Second half of the in-pass: the LDS tile (row = runoff row - row_first, kRecTileLd doubles per row) becomes records.
template <bool SUB, int THREADS = kRecThreads>
__device__ __forceinline__ void write_records(const RecPermArgs &a, const double *tile, int64_t col0, int64_t tick_first, int64_t row_first)
{
    constexpr int R = kRecTileRows;
    const int tid = threadIdx.x;
    constexpr int IT = kRecCols * kRecBatch * 8 / THREADS;
    int2 meta[IT];
    double f[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {     // all metadata loads first: they are independent
        const int64_t i = col0 + (it * THREADS + tid) / (8 * kRecBatch);
        meta[it] = i < a.n ? a.colmeta[i] : make_int2(-1, 0);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t i = col0 + (it * THREADS + tid) / (8 * kRecBatch);
        f[it] = (a.scale && i < a.n) ? a.scale[i] : 1.0;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int piece = it * THREADS + tid;       // (column, record, 16-byte part): 8 consecutive lanes = one record
        const int c = piece / (8 * kRecBatch), k = (piece >> 3) % kRecBatch, part = piece & 7;
        const int32_t p = meta[it].x;
        if (p < 0) continue;
        const int32_t lag = meta[it].y;
        const int o = lag & 15;
        const uint32_t chunk = (uint32_t)kRecBatch * (uint32_t)a.batch + (uint32_t)(lag >> 4) + k;
        const int r = 15 - o + 16 * k + 2 * part;       // tick-row tick_first + r
        double v0, v1;
        if (SUB) {
            const int64_t t0 = tick_first + r, t1 = t0 + 1;
            uint32_t s;
            const int r0 = t0 < 0 ? 0 : (int)((int64_t)a.nsub.div((uint32_t)t0, s) - row_first);
            const int r1 = t1 < 0 ? 0 : (int)((int64_t)a.nsub.div((uint32_t)t1, s) - row_first);
            v0 = tile[min(r0, R - 1) * kRecTileLd + c] * f[it]; v1 = tile[min(r1, R - 1) * kRecTileLd + c] * f[it];
        } else {
            v0 = tile[r * kRecTileLd + c] * f[it]; v1 = tile[(r + 1) * kRecTileLd + c] * f[it];
        }
        double2 *dst = reinterpret_cast<double2 *>(a.rec + ((int64_t)a.rec_chunks.mod(chunk) * a.np + p) * kRec) + part;
        *dst = make_double2(v0, v1);
    }
}

template <bool SUB>
__global__ __launch_bounds__(kRecThreads) void k_rec_in(const RecPermArgs a)
{
    constexpr int R = kRecTileRows;
    __shared__ double tile[R * kRecTileLd];
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * kRecCols;
    const int64_t tick_first = kRecRows * a.batch - 15;                 // may be negative in the first batch
    uint32_t sub_unused;
    const int64_t row_first = SUB ? (int64_t)a.nsub.div((uint32_t)(tick_first < 0 ? 0 : tick_first), sub_unused) : tick_first;
    {   // all row loads in flight first (branch-free: out-of-range rows/columns are clamped and zeroed afterwards)
        constexpr int RPT = (R + kRecThreads / kRecCols - 1) / (kRecThreads / kRecCols);
        const int c = tid % kRecCols, r0 = tid / kRecCols;
        const int64_t i = min(col0 + c, a.n - 1);
        const int need = SUB ? (int)((uint32_t)(R - 1) / a.nsub.d) + 2 : R;     // runoff rows behind the batch's tick-rows
        double v[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int64_t t = row_first + min(r0 + q * (kRecThreads / kRecCols), need - 1);
            v[q] = a.rows.row(t < 0 ? 0 : (t >= a.T ? a.T - 1 : t))[i];
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = r0 + q * (kRecThreads / kRecCols);
            const int64_t t = row_first + r;
            if (r < R) tile[r * kRecTileLd + c] = (t >= 0 && t < a.T && col0 + c < a.n) ? v[q] : 0.0;
        }
    }
    __syncthreads();
    write_records<SUB>(a, tile, col0, tick_first, row_first);
}

// The in-pass with the unit-hydrograph convolution fused in (UnitHydrograph.py:93-107, direct form): the tile is COMPUTED
// from the runoff-depth rows instead of loaded, so the convolved lateral never exists as (T, n) rows in HBM (written by the
// convolution kernel, read again by k_rec_in: 16 B per value).  The block loads the depth rows behind its 143 tick-rows plus
// the n_ks - 1 before them and the kernel's taps into LDS; thread (column, group of 18 rows) pulls its window of 18 + NK - 1
// depth values into registers and accumulates 18 outputs x NK taps with static indices (NK = n_ks padded with zero taps);
// the outputs replace the depth tile in LDS and leave as records.  out[t] = [t < n_ks] state[t] + sum_k kernel[k] depth[t - k].
struct UhArgs {
    const double *kernel, *state;     // (n_ks, n) taps and carried-in state, params order
    int32_t n_ks;
};
constexpr int kUhInThreads = 256;       // 8 groups of rows x 32 columns: 18 outputs per thread, windows of 18 + NK - 1 depth values (512 threads x 9 rows: 20 % slower)
constexpr int kUhRowsPerThread = (kRecTileRows + kUhInThreads / kRecCols - 1) / (kUhInThreads / kRecCols);
constexpr size_t rec_in_uh_lds_bytes(int nk) { return (size_t)((kRecTileRows + nk - 1) + nk) * kRecTileLd * sizeof(double); }

template <bool SUB, int NK>
__global__ __launch_bounds__(kUhInThreads) void k_rec_in_uh(const RecPermArgs a, const UhArgs u)
{
    constexpr int R = kRecTileRows, G = kUhInThreads / kRecCols, RP = kUhRowsPerThread, W = RP + NK - 1;
    extern __shared__ __attribute__((aligned(16))) double uh_lds[];
    double *dt = uh_lds;                                   // [R + NK - 1][kRecTileLd] depth rows row_first - (NK - 1) ...
    double *tp = uh_lds + (R + NK - 1) * kRecTileLd;       // [NK][kRecTileLd] taps
    const int tid = threadIdx.x, c = tid % kRecCols, g = tid / kRecCols;
    const int64_t col0 = (int64_t)blockIdx.x * kRecCols;
    const int64_t tick_first = kRecRows * a.batch - 15;
    uint32_t sub_unused;
    const int64_t row_first = SUB ? (int64_t)a.nsub.div((uint32_t)(tick_first < 0 ? 0 : tick_first), sub_unused) : tick_first;
    const int need = SUB ? (int)((uint32_t)(R - 1) / a.nsub.d) + 2 : R;
    const int64_t i = min(col0 + c, a.n - 1);
    const bool live = col0 + c < a.n;
    {   // all loads in flight first (branch-free: out-of-range rows/columns are clamped and zeroed afterwards)
        constexpr int DPT = (R + NK - 1 + G - 1) / G, TPT = (NK + G - 1) / G;
        double dv[DPT], tv[TPT];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int64_t t = row_first - (NK - 1) + min(g + q * G, need + NK - 2);
            dv[q] = a.rows.row(t < 0 ? 0 : (t >= a.T ? a.T - 1 : t))[i];
        }
#pragma unroll
        for (int q = 0; q < TPT; ++q) tv[q] = u.kernel[(int64_t)min(g + q * G, u.n_ks - 1) * a.n + i];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int r = g + q * G;
            const int64_t t = row_first - (NK - 1) + r;
            if (r < R + NK - 1) dt[r * kRecTileLd + c] = (live && t >= 0 && t < a.T) ? dv[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            const int k = g + q * G;
            if (k < NK) tp[k * kRecTileLd + c] = (live && k < u.n_ks) ? tv[q] : 0.0;
        }
    }
    __syncthreads();
    const int rb = g * RP;      // first output row of this thread
    double acc[RP];
    if (rb < need) {
        double win[W];
#pragma unroll
        for (int q = 0; q < W; ++q) win[q] = dt[min(rb + q, R + NK - 2) * kRecTileLd + c];      // depth row (row_first + rb + q - (NK - 1))
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int64_t t = row_first + rb + j;
            acc[j] = (live && t >= 0 && t < u.n_ks && t < a.T) ? u.state[t * a.n + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const double tap = tp[k * kRecTileLd + c];
#pragma unroll
            for (int j = 0; j < RP; ++j) acc[j] = __builtin_fma(tap, win[j + (NK - 1) - k], acc[j]);
        }
    }
    __syncthreads();      // every window is in registers: the depth tile's space now takes the outputs
    if (rb < need) {
#pragma unroll
        for (int j = 0; j < RP; ++j) if (rb + j < R) dt[(rb + j) * kRecTileLd + c] = acc[j];
    }
    __syncthreads();
    write_records<SUB, kUhInThreads>(a, dt, col0, tick_first, row_first);
}

// OUT32: the router's post-processing fused in (TransformMuskingum.py:128-142): mean over `factor` consecutive rows
// (sequential sum, one division, as numpy reduces a strided axis) and the float32 cast; 128 % (factor * nsub) == 0.
template <bool SUB, bool OUT32>
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
        v[it] = *(reinterpret_cast<const double2 *>(a.rec + ((int64_t)a.rec_chunks.mod(chunk) * a.np + p) * kRec) + part);
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
    const int64_t tick0 = kRecRows * a.batch;
    if (OUT32) {
        // output row q averages runoff rows [q * factor, (q + 1) * factor), each the slot of its last sub-step
        const int step = (int)(a.factor.d * (SUB ? a.nsub.d : 1u));          // tick-rows per output row, divides 128
        const int64_t q0 = tick0 / step;
        for (int q = tid / kRecCols; q < kRecRows / step; q += kRecThreads / kRecCols) {
            if ((q0 + q + 1) * step > a.total) break;
            const int nsub = SUB ? (int)a.nsub.d : 1;
            double acc = recs[c][o + q * step + nsub - 1];
            for (int j = 1; j < (int)a.factor.d; ++j) acc += recs[c][o + q * step + j * nsub + nsub - 1];
            a.rows32[(q0 + q) * a.n + i] = (float)(a.factor.d > 1 ? acc / (double)a.factor.d : acc);
        }
        return;
    }
    for (int r = tid / kRecCols; r < kRecRows; r += kRecThreads / kRecCols) {
        const int64_t tick = tick0 + r;
        if (tick >= a.total) break;
        if (SUB) {
            uint32_t s;
            const uint32_t t = a.nsub.div((uint32_t)tick, s);
            if (s + 1 == a.nsub.d) a.rows.row(t)[i] = recs[c][o + r];
        } else {
            a.rows.row(tick)[i] = recs[c][o + r];
        }
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

// The in-pass with the gridded-runoff aggregation fused in (one sub-step per row): thread (river i, record k of the batch)
// computes the 16 rows of ONE record of river i -- rows [128 j + 16 k - o, + 16), o = lag % 16: the record boundaries of a
// river follow its lag -- exactly as k_runoff_to_qlateral computes its 16-row chunks (same gather, same rounding, same
// post-processing), times c4dt, and the block's 256 records leave through LDS eight lanes per record.  The catchment
// inflow never exists as (T, n) rows in HBM.
struct RunoffArgs {
    const int32_t *indptr, *indices;
    const double *weights, *area;
    const void *runoff;
    int64_t stride_t, stride_p;
    int32_t flags, is_f32;
};
constexpr int kRunoffInThreads = 256;

template <typename RT>
__global__ __launch_bounds__(kRunoffInThreads) void k_rec_in_runoff(const RecPermArgs a, const RunoffArgs g)
{
    __shared__ double stage[kRunoffInThreads][kRec + 1];
    __shared__ int64_t slot[kRunoffInThreads];      // record index (chunk % chunks) * np + position, -1: no record
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kRunoffInThreads + tid;
    const int k = blockIdx.y;
    const bool cumulative = g.flags & RR_RUNOFF_CUMULATIVE, force_positive = g.flags & RR_RUNOFF_FORCE_POSITIVE,
               keep_nan = g.flags & RR_RUNOFF_KEEP_NAN;
    slot[tid] = -1;
    if (i < a.n) {
        const int2 meta = a.colmeta[i];
        const int32_t lag = meta.y, o = lag & 15;
        const uint32_t chunk = (uint32_t)kRecBatch * (uint32_t)a.batch + (uint32_t)(lag >> 4) + (uint32_t)k;
        const int64_t t0 = kRecRows * a.batch + 16 * k - o;      // first row of the record; rows outside [0, T) hold zeros
        double acc[kRec + 1];        // slot 0: row t0 - 1 (cumulative input only)
#pragma unroll
        for (int j = 0; j <= kRec; ++j) acc[j] = 0.0;
        const RT *base = static_cast<const RT *>(g.runoff);
        // A record's 16 rows start wherever the river's lag puts them, so the gather cannot be 16-byte aligned; with the
        // block point-major the window is still one run of memory and gfx950 takes 16-byte loads at element alignment.
        // Records that stick out of the grid point's (padded) row -- the first of a river, the last -- go row by row.
        constexpr int VL = 16 / (int)sizeof(RT);
        typedef RT VecU __attribute__((ext_vector_type(VL), aligned(sizeof(RT))));
        const bool inside = g.stride_t == 1 && t0 >= 1 && t0 + kRec <= g.stride_p;
        for (int32_t e = g.indptr[i]; e < g.indptr[i + 1]; ++e) {
            const double w = g.weights[e];
            const RT *src = base + (int64_t)g.indices[e] * g.stride_p;
            if (inside) {
                if (cumulative) acc[0] = __dadd_rn(acc[0], __dmul_rn(w, (double)src[t0 - 1]));
                VecU x[kRec / VL];
#pragma unroll
                for (int q = 0; q < kRec / VL; ++q) x[q] = *reinterpret_cast<const VecU *>(src + t0 + q * VL);      // rows past T: padding, zeroed below
#pragma unroll
                for (int q = 0; q < kRec / VL; ++q)
#pragma unroll
                    for (int v = 0; v < VL; ++v) acc[1 + q * VL + v] = __dadd_rn(acc[1 + q * VL + v], __dmul_rn(w, (double)x[q][v]));
            } else {
#pragma unroll
                for (int j = 0; j <= kRec; ++j) {
                    const int64_t t = t0 - 1 + j;
                    if (t >= 0 && t < a.T && (j > 0 || cumulative)) acc[j] = __dadd_rn(acc[j], __dmul_rn(w, (double)src[t * g.stride_t]));
                }
            }
        }
        const double area = g.area ? g.area[i] : 1.0, f = a.scale ? a.scale[i] : 1.0;
#pragma unroll
        for (int j = 1; j <= kRec; ++j) {
            const int64_t t = t0 - 1 + j;
            double v = (cumulative && t > 0) ? acc[j] - acc[j - 1] : acc[j];
            if (force_positive) v = v < 0.0 ? 0.0 : v;      // np.clip leaves NaN alone, as does this comparison
            if (v != v && !keep_nan) v = 0.0;
            if (g.area) v = v * area;
            stage[tid][j - 1] = (t >= 0 && t < a.T) ? v * f : 0.0;
        }
        slot[tid] = (int64_t)a.rec_chunks.mod(chunk) * a.np + meta.x;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int piece = it * kRunoffInThreads + tid, r = piece >> 3, part = piece & 7;      // 8 consecutive lanes = one record
        const int64_t sl = slot[r];
        if (sl < 0) continue;
        reinterpret_cast<double2 *>(a.rec + sl * kRec)[part] = make_double2(stage[r][2 * part], stage[r][2 * part + 1]);
    }
}

// Device copy rate probe (bench.py reports it beside the nominal HBM peak): 16 bytes per lane, grid-stride.
__global__ __launch_bounds__(kBlock) void k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t count)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {      // four loads in flight per lane
        const double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < count; i += stride) dst[i] = src[i];
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
    float *dev_out32 = nullptr;       // instead of dev_out: float32 rows, each the mean of out_factor routed rows
    int64_t out_factor = 1;
    // UnitMuskingum with the convolution fused into the in-pass: dev_in holds runoff DEPTH rows, the lateral inflow is
    // computed on the way into the records (k_rec_in_uh)
    const double *uh_kernel = nullptr, *uh_state = nullptr;
    int64_t uh_nks = 0;
    // RapidMuskingum fed by gridded runoff: no lateral rows at all, the weights product runs in the in-pass (k_rec_in_runoff)
    const RunoffArgs *runoff = nullptr;
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
    bool wave = false;            // time-tiled k_tile over records instead of per-tick k_tick over rows
    int64_t KC = 1;               // record chunks per task: K = 16 * KC ticks
    int64_t rec_chunks = 0, in_batches = 0, n_in_batches = 0, out_batches = 0, n_out_batches = 0;
    int64_t ticks_stored = 0;     // tick-rows that have left the record ring
    int64_t out_limit = std::numeric_limits<int64_t>::max();   // rows the caller's output ring can take (host pipeline)
    int64_t diag = 0, n_diags = 0, n_macro = 0;
    int64_t ghost_batches = 0;    // batches of the boundary (ghost) series turned into records
    int64_t ghost_slack = 0, export_skew = 0;      // boundary reaches of a partitioned network in the time-tiled schedule (level skew included)
    TileArgs ta{};
    bool bracket_open = false;
    int64_t bracket_reaches = 0;
    size_t max_samples = 0;
};

// ---- host-pointer calls: PCIe pipeline around the time-tiled kernel ----
//
// The reference's kernel boundary hands over numpy arrays in pageable host memory.  hipMemcpy from pageable memory moves
// 22 GB/s here, and one direction at a time; registering the caller's arrays costs 43 ms per GB; pinned memory moves
// 49 GB/s each way at once (profiles/microbench/host_copy.hip).  So rows travel in chunks of 64 through three pinned
// buffers per direction, filled and emptied by eight copy threads each, while the DMA engines move the neighbouring
// chunks and the GPU routes what has arrived: caller -> pinned -> device staging ring -> records -> tiles -> records ->
// device staging ring -> pinned -> caller, every stage overlapping the others.  The open routing call is the streaming
// session the partitioned path uses (rows become ready chunk by chunk).
struct HostPipe {
    static constexpr int kPinned = 3, kCopyThreads = 8;
    int64_t chunk_rows = 64, ring_chunks = 8;
    double *pin_in[kPinned] = {nullptr, nullptr, nullptr}, *pin_out[kPinned] = {nullptr, nullptr, nullptr};
    double *dev_in = nullptr, *dev_out = nullptr;
    int64_t pin_cap = 0, dev_cap = 0;      // doubles per pinned buffer / per device ring
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<hipEvent_t> ev_h2d, ev_d2h, ev_adv;
    void destroy()
    {
        for (int k = 0; k < kPinned; ++k) { if (pin_in[k]) (void)hipHostFree(pin_in[k]); if (pin_out[k]) (void)hipHostFree(pin_out[k]); pin_in[k] = pin_out[k] = nullptr; }
        if (dev_in) (void)hipFree(dev_in);
        if (dev_out) (void)hipFree(dev_out);
        dev_in = dev_out = nullptr; pin_cap = dev_cap = 0;
        for (auto *v : {&ev_h2d, &ev_d2h, &ev_adv}) { for (hipEvent_t e : *v) (void)hipEventDestroy(e); v->clear(); }
        if (s_h2d) (void)hipStreamDestroy(s_h2d);
        if (s_d2h) (void)hipStreamDestroy(s_d2h);
        s_h2d = s_d2h = nullptr;
    }
};

struct rr_plan {
    rr::HostPlan h;
    int device = RR_DEVICE_NONE;
    bool coeffs_set = false, has_c4 = false;
    int64_t chunk_rows = 16, sample_every = 0;

    // streaming kernel (k_tick): lag-ordered layout of rr::HostPlan
    int32_t *d_child_ptr = nullptr, *d_lag = nullptr, *d_perm = nullptr, *d_inv = nullptr, *d_inner_pos = nullptr;
    int32_t *d_bidx = nullptr;   // ghost / export slot of flagged positions
    uint16_t *d_hwc = nullptr;
    double *d_w = nullptr, *d_c1row_h = nullptr, *d_c2 = nullptr, *d_c3 = nullptr, *d_c4 = nullptr;
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

    // time-tiled routing (k_tile): subtree tiles of rr::TilePlan
    rr::TilePlan tp;
    bool wave_enabled = true, wave_forced = false, wave_now = false, weights_uniform = false;
    int wave_threads = 1024, wave_ppt = 2;
    int64_t wave_K = 0;          // ticks per task (multiple of 16); 0 = chosen per call
    int64_t next_KC = 1, next_chunks = 0;   // decide_wave: task length and record ring of the call about to start
    int32_t *d_tile_ptr = nullptr, *d_tile_level = nullptr, *d_tile_lag_lo = nullptr, *d_tile_lag_hi = nullptr;
    int32_t *d_tlag = nullptr, *d_cfirst = nullptr, *d_xpos = nullptr, *d_tperm = nullptr, *d_tinv = nullptr;
    int32_t *d_tbidx = nullptr, *d_inner_idx = nullptr;
    uint32_t *d_ccnt = nullptr;
    double *d_c1row = nullptr, *d_tc2 = nullptr, *d_tc3 = nullptr, *d_sq = nullptr, *d_ss = nullptr, *d_si = nullptr, *d_sqch = nullptr;
    double *d_full = nullptr, *d_chan = nullptr;   // UnitMuskingum state scattered to params order
    int2 *d_colmeta = nullptr;   // per params column {position, lag}
    int2 *d_ghostmeta = nullptr; // the same per boundary ghost (column of the ghost series)
    double *d_c4_params = nullptr;   // c4dt in params order (scale of the record permutation)
    size_t dev_total_bytes = 0;
    int cu_count = 256;

    // boundary reaches of a partitioned network
    int64_t n_ghost = 0, n_export = 0;
    int64_t ghost_min_lag = 0, export_max_lag = 0;
    std::vector<int32_t> ghost_reach, export_reach;   // params indices, in the caller's order

    Session ses;
    HostPipe pipe;      // staging of the host-pointer entry points (allocated at first use)

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

// Record chunks per task.  A longer task amortises the load of the tile's state and of its first half chunk, which
// nothing overlaps; every tile level adds one task of skew to the pipeline and to the record ring.
int64_t pick_KC(const rr_plan *P, int64_t total_ticks)
{
    if (P->wave_K > 0) return std::max<int64_t>(1, P->wave_K / kRec);
    return total_ticks >= 4096 ? 4 : (total_ticks >= 512 ? 2 : 1);
}

// Which routing kernel a call uses.  The time-tiled schedule needs device rows, one upstream weight per reach, a
// network that tiles (rr::TilePlan) and room for its record ring; its fill and drain cost (levels x K) ticks more than
// the streaming kernel's, a few launches, so only calls of a handful of sub-steps stream.  RR_WAVE=1 forces it where it
// applies, RR_WAVE=0 forbids it.
//
// Records are indexed by tick = tick-row + lag, modulo the ring, per position: a position's slots never hold another
// position's data, so what the ring must cover is one position's tick-rows in flight.  Rows enter for all columns at once
// (ahead of the level-0 tiles) and leave for all columns at once (after the last level has passed their tick + depth), so
// every position keeps depth + levels * K tick-rows plus the batching of the two permutation passes; that its window sits
// lag ticks later than a headwater's does not widen it.  The ring may take five eighths of the card; a deep network that
// does not fit gets shorter tasks, then the streaming kernel.
bool decide_wave(rr_plan *P, Mode mode, int64_t total, bool host_rows)
{
    bool ok = P->wave_enabled && P->tp.ok && P->weights_uniform && P->h.n > 0 && !host_rows && P->tp.np < (int64_t{1} << 25);
    if (ok && !P->wave_forced) ok = total >= 32;
    if (ok) {
        const int64_t dmax = P->h.depth - 1, np = P->tp.np, levels = P->tp.n_levels;
        const int64_t all_chunks = kRecBatch * ((total + 14) / kRecRows + 2) + (dmax >> 4) + 2;
        ok = false;
        for (int64_t KC = pick_KC(P, total + dmax); KC >= 1; KC /= 2) {
            static const int64_t extra = getenv("RR_RING_EXTRA") ? atoll(getenv("RR_RING_EXTRA")) : 0;      // measurements: a larger ring than needed
            const int64_t chunks = std::min<int64_t>(all_chunks, (dmax + levels * KC * kRec) / kRec + 4 * kRecBatch + extra);
            const int64_t bytes = chunks * kRec * np * (int64_t)sizeof(double);
            if (P->dev_total_bytes > 0 && bytes > (int64_t)(P->dev_total_bytes / 8 * 5)) continue;
            if (ensure_cap(&P->d_ring, &P->ring_cap, chunks * kRec * np) != RR_OK) { (void)hipGetLastError(); continue; }
            P->next_KC = KC; P->next_chunks = chunks;
            ok = true;
            break;
        }
    }
    P->wave_now = ok;
    return ok;
}

bool use_wave(const rr_plan *P, Mode) { return P->wave_now; }

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
    S.wave = use_wave(P, mode);
    S.direct = H.identity && !host_io && !S.wave;   // engine order == params order: the streaming kernel reads the caller's arrays
    const int64_t C = std::max<int64_t>(1, P->chunk_rows);

    P->prof_launches = P->prof_samples = P->prof_brackets = 0;
    P->prof_reach_steps = n * S.total;
    P->ev_reaches.clear();
    P->last_stream = stream;
    S.open = true;
    if (n == 0 || S.total == 0) return RR_OK;
    if (P->n_ghost > 0 && !ghost_series) { S.open = false; return fail(RR_E_INVALID, "plan has ghost reaches but no ghost series was given"); }
    if (P->n_export > 0 && !export_series) { S.open = false; return fail(RR_E_INVALID, "plan has export reaches but no export series was given"); }
    if (io.dev_out32 && !S.wave) { S.open = false; return fail(RR_E_UNSUPPORTED, "float32 output needs the time-tiled kernel"); }

    int rc = RR_OK;
    if (S.wave) { S.KC = P->next_KC; S.rec_chunks = P->next_chunks; }     // ring sized and allocated by decide_wave
    if (S.wave) {
        const rr::TilePlan &TP = P->tp;
        const int64_t K = S.KC * kRec;
        S.n_macro = (S.total_ticks + K - 1) / K;
        S.n_diags = S.n_macro + TP.n_levels - 1;
        S.n_in_batches = (S.total + 14) / kRecRows + 1;      // of the lateral rows (if any) and of the boundary series (if any)
        S.n_out_batches = (S.total + kRecRows - 1) / kRecRows;
        // external boundary reaches: a ghost in a tile of level l at lag L is read for sub-steps below (diag - l + 1) K - L,
        // an export reach there has produced the sub-steps below (diag - l) K - L
        S.export_skew = 0;
        S.ghost_slack = P->ghost_reach.empty() ? 0 : S.total_ticks + (int64_t)TP.n_levels * K;
        for (int32_t i : P->ghost_reach) { const int32_t p = TP.inv[i]; S.ghost_slack = std::min<int64_t>(S.ghost_slack, (int64_t)TP.tile_level[TP.tile_of[p]] * K + (TP.lag[p] & kLagMask)); }
        for (int32_t i : P->export_reach) { const int32_t p = TP.inv[i]; S.export_skew = std::max<int64_t>(S.export_skew, (int64_t)TP.tile_level[TP.tile_of[p]] * K + (TP.lag[p] & kLagMask)); }
        if (io.dev_out32) {
            const int64_t step = io.out_factor * nsub;
            if (io.out_factor < 1 || kRecRows % step != 0 || T % io.out_factor != 0) { S.open = false; return fail(RR_E_UNSUPPORTED, "float32 output: factor * sub-steps must divide 128 and factor the number of rows"); }
        }
        TileArgs &w = S.ta;
        w.tile_ptr = P->d_tile_ptr; w.tile_level = P->d_tile_level; w.tile_lag_lo = P->d_tile_lag_lo; w.tile_lag_hi = P->d_tile_lag_hi;
        w.lag = P->d_tlag; w.cfirst = P->d_cfirst; w.xpos = P->d_xpos; w.ccnt = P->d_ccnt;
        w.c1row = P->d_c1row; w.c2 = P->d_tc2; w.c3 = P->d_tc3;
        w.sq = P->d_sq; w.ss = P->d_ss; w.si = P->d_si; w.sqch = P->d_sqch;
        w.bidx = P->d_tbidx; w.exports = export_series; w.n_export = (int32_t)P->n_export;
        w.rec = P->d_ring; w.rec_chunks = Div32((uint32_t)S.rec_chunks);
#ifdef RR_WAVE_TRACE
        w.trace = nullptr; w.trace_diag = -1;
        if (getenv("RR_WAVE_TRACE_DIAG")) {
            static long long *tbuf = nullptr;
            if (!tbuf) (void)hipMalloc(&tbuf, 8 * 16 * 4096);
            (void)hipMemset(tbuf, 0, 8 * 16 * 4096);
            w.trace = tbuf; w.trace_diag = atoi(getenv("RR_WAVE_TRACE_DIAG"));
        }
#endif
        w.np = (int32_t)TP.np; w.KC = (int32_t)S.KC; w.n_macro = (int32_t)S.n_macro; w.total = (int32_t)S.total;
        w.has_lat = S.has_in ? 1 : 0; w.nsub = Div32((uint32_t)nsub); w.inv_nsub = 1.0 / (double)nsub;
    }
    if (getenv("RR_VERBOSE"))
        fprintf(stderr, "rr: n=%lld T=%lld nsub=%lld tiled=%d K=%lld tiles=%d levels=%d block=%d ghosts=%lld ring_chunks=%lld (%.1f GB) lds=%zu\n",
                (long long)n, (long long)T, (long long)nsub, (int)S.wave, (long long)(S.KC * kRec), P->tp.n_tiles, P->tp.n_levels, P->tp.block,
                (long long)P->tp.n_ghost, (long long)S.rec_chunks, S.wave ? (double)S.rec_chunks * kRec * P->tp.np * 8 / 1e9 : 0.0,
                tile_lds_bytes(P->wave_threads));
    if (!S.wave) {
        // work ring in engine order: lateral rows come in, discharge rows overwrite them in place; rows stay until the
        // outlet-most reaches have passed them
        const int64_t lag_rows = (dmax + nsub - 1) / nsub;
        S.ring_rows = S.direct ? 0 : std::min<int64_t>(T, lag_rows + 2 * C + 2);
        if (S.ring_rows > 0xFFFFFFFFLL || T > 0x7FFFFFFFLL) { S.open = false; return fail(RR_E_INVALID, "route: too many time rows"); }
        rc = RR_OK;
        if (!S.direct) rc = ensure_cap(&P->d_ring, &P->ring_cap, S.ring_rows * n);
        if (!rc && !S.direct) rc = ensure_cap(&P->d_mrows, &P->mrows_cap, C * n);
        if (!rc && host_io) rc = ensure_cap(&P->d_stage, &P->stage_cap, C * n);
        if (rc) { S.open = false; return rc; }
        TickArgs &a = S.a;
        a.child_ptr = P->d_child_ptr; a.lag = P->d_lag; a.w = P->d_w; a.c2 = P->d_c2; a.c3 = P->d_c3; a.c4 = P->d_c4;
        a.c1row = P->weights_uniform ? P->d_c1row_h : nullptr;
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
    }
    S.max_samples = P->sample_every >= kSampleGroup ? (size_t)std::min<int64_t>(4096, S.total_ticks / P->sample_every + 1) : 0;
    if (S.wave && S.max_samples > 0) S.max_samples = (size_t)std::min<int64_t>(4096, S.n_diags / 4 + 1);     // every fourth launch
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

typedef void (*tile_kernel_t)(const TileArgs);

// Tile = one position per thread; 16 waves per CU whose two record buffers fill the register file, as 1, 2 or 4 workgroups.
tile_kernel_t tile_kernel(int threads, bool unit, bool sub)
{
#define RR_TILE_PICK(T_) (unit ? (sub ? (tile_kernel_t)k_tile<T_, true, true> : (tile_kernel_t)k_tile<T_, true, false>)   \
                               : (sub ? (tile_kernel_t)k_tile<T_, false, true> : (tile_kernel_t)k_tile<T_, false, false>))
    return threads == 256 ? RR_TILE_PICK(256) : (threads == 512 ? RR_TILE_PICK(512) : RR_TILE_PICK(1024));
#undef RR_TILE_PICK
}

// Launch d of the time-tiled schedule: the tasks (tile, macro-chunk d - level) of every tile whose macro-chunk exists.
// Tiles are stored by level, so they are one contiguous range; a tile with no active position returns at once.
int session_launch_diag(rr_plan *P, int64_t d)
{
    Session &S = P->ses;
    const rr::TilePlan &TP = P->tp;
    const int64_t l_lo = std::max<int64_t>(0, d - (S.n_macro - 1)), l_hi = std::min<int64_t>(TP.n_levels - 1, d);
    if (l_hi < l_lo) { ++P->prof_launches; return RR_OK; }
    int64_t t_lo = TP.level_start[l_lo], t_hi = TP.level_start[l_hi + 1];
    // tiles are sorted by their smallest lag inside a level; while the pipeline fills, the tiles of level 0 that
    // have not started yet are a suffix of it
    const int64_t K = S.KC * kRec;
    if (l_lo == 0) {
        const int64_t end0 = TP.level_start[1];
        int64_t hi = std::min<int64_t>(t_hi, end0);
        while (hi > t_lo && (d + 1) * K <= TP.tile_lag_lo[hi - 1]) --hi;
        if (t_hi <= end0) t_hi = hi;     // only level 0 in this launch: trim; otherwise the idle ones just return
    }
    if (t_hi <= t_lo) { ++P->prof_launches; return RR_OK; }
    TileArgs &w = S.ta;
    w.diag = (int32_t)d; w.t_first = (int32_t)t_lo; w.t_last = (int32_t)t_hi - 1;
    // every fourth launch is bracketed by HIP events, full or not (fill and drain launches run fewer tiles), so the
    // sampled average is the average rocprofv3 reports for the kernel; the reach-ticks of a sample are those of the
    // tiles it launched
    const bool sample = S.max_samples > 0 && (P->prof_launches % 4) == 0 && (size_t)P->prof_brackets < S.max_samples;
    if (sample) HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets], S.stream));
    // one workgroup per resident slot (16 waves per CU); each walks its share of the launch's tiles
    const dim3 g((unsigned)std::min<int64_t>(t_hi - t_lo, (int64_t)P->cu_count * (1024 / P->wave_threads)));
    const size_t lds_bytes = tile_lds_bytes(P->wave_threads);
    hipLaunchKernelGGL(tile_kernel(P->wave_threads, S.mode == Mode::Unit, S.nsub > 1), g, dim3((unsigned)P->wave_threads), lds_bytes, S.stream, w);
    if (sample) {
        HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets + 1], S.stream));
        P->ev_reaches.push_back((int64_t)(TP.tile_ptr[t_hi] - TP.tile_ptr[t_lo]) * K);
        P->prof_samples += K;
        ++P->prof_brackets;
    }
    ++P->prof_launches;
    return RR_OK;
}

// Boundary inflow of a partitioned network: the ghost series (total sub-steps x ghosts, row = sub-step) is a matrix of
// tick-rows like the lateral rows, and its columns become the records of the ghost positions by the same pass.
void launch_ghost_permute(rr_plan *P, int64_t batch)
{
    Session &S = P->ses;
    RecPermArgs ra{};
    ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = P->n_ghost; ra.np = P->tp.np; ra.T = S.total; ra.total = S.total;
    ra.batch = batch; ra.nsub = Div32(1u); ra.colmeta = P->d_ghostmeta; ra.scale = nullptr;
    ra.rows = RowView{const_cast<double *>(S.ghost_series), P->n_ghost, 0, (uint32_t)S.total};
    ra.factor = Div32(1u);
    hipLaunchKernelGGL(k_rec_in<false>, dim3((unsigned)((P->n_ghost + kRecCols - 1) / kRecCols)), dim3(kRecThreads), 0, S.stream, ra);
}

typedef void (*rec_in_uh_t)(const RecPermArgs, const UhArgs);
constexpr int kUhFusedMaxTaps = 64;
int uh_padded_taps(int64_t n_ks) { return n_ks <= 16 ? 16 : (n_ks <= 48 ? 48 : 64); }
rec_in_uh_t rec_in_uh_kernel(bool sub, int64_t n_ks)
{
    const int nk = uh_padded_taps(n_ks);
#define RR_UHIN_PICK(NK_) (sub ? (rec_in_uh_t)k_rec_in_uh<true, NK_> : (rec_in_uh_t)k_rec_in_uh<false, NK_>)
    return nk == 16 ? RR_UHIN_PICK(16) : (nk == 48 ? RR_UHIN_PICK(48) : RR_UHIN_PICK(64));
#undef RR_UHIN_PICK
}

void launch_rec_permute(rr_plan *P, bool in, int64_t batch)
{
    Session &S = P->ses;
    const int64_t n = P->h.n;
    RecPermArgs ra{};
    ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = n; ra.np = P->tp.np; ra.T = S.T; ra.total = S.total; ra.batch = batch;
    ra.nsub = Div32((uint32_t)S.nsub);
    ra.colmeta = P->d_colmeta;
    ra.scale = (in && S.mode == Mode::Rapid) ? P->d_c4_params : nullptr;
    ra.rows = in ? RowView{const_cast<double *>(S.io.dev_in), n, 0, (uint32_t)S.io.rows_in}
                 : RowView{S.io.dev_out, n, 0, (uint32_t)std::max<int64_t>(1, S.io.rows_out)};
    ra.rows32 = in ? nullptr : S.io.dev_out32;
    ra.factor = Div32((uint32_t)std::max<int64_t>(1, S.io.out_factor));
    const dim3 g((unsigned)((n + kRecCols - 1) / kRecCols));
    const bool sub = S.nsub > 1;
    if (in && S.io.runoff) {
        const dim3 gr((unsigned)((n + kRunoffInThreads - 1) / kRunoffInThreads), (unsigned)kRecBatch);
        if (S.io.runoff->is_f32) hipLaunchKernelGGL(k_rec_in_runoff<float>, gr, dim3(kRunoffInThreads), 0, S.stream, ra, *S.io.runoff);
        else hipLaunchKernelGGL(k_rec_in_runoff<double>, gr, dim3(kRunoffInThreads), 0, S.stream, ra, *S.io.runoff);
    } else if (in && S.io.uh_kernel) {
        UhArgs ua{S.io.uh_kernel, S.io.uh_state, (int32_t)S.io.uh_nks};
        hipLaunchKernelGGL(rec_in_uh_kernel(sub, S.io.uh_nks), g, dim3(kUhInThreads), rec_in_uh_lds_bytes(uh_padded_taps(S.io.uh_nks)), S.stream, ra, ua);
    } else if (in) {
        if (sub) hipLaunchKernelGGL(k_rec_in<true>, g, dim3(kRecThreads), 0, S.stream, ra);
        else hipLaunchKernelGGL(k_rec_in<false>, g, dim3(kRecThreads), 0, S.stream, ra);
    } else if (ra.rows32) {
        if (sub) hipLaunchKernelGGL((k_rec_out<true, true>), g, dim3(kRecThreads), 0, S.stream, ra);
        else hipLaunchKernelGGL((k_rec_out<false, true>), g, dim3(kRecThreads), 0, S.stream, ra);
    } else {
        if (sub) hipLaunchKernelGGL((k_rec_out<true, false>), g, dim3(kRecThreads), 0, S.stream, ra);
        else hipLaunchKernelGGL((k_rec_out<false, false>), g, dim3(kRecThreads), 0, S.stream, ra);
    }
}

// Time-tiled schedule: batches of 128 tick-rows become records as soon as their rows are there and their ring slots
// are free, launches run while their input is present, finished batches leave.
int session_advance_tile(rr_plan *P, int64_t rows_ready, int64_t ghost_ready, int64_t *export_ready)
{
    Session &S = P->ses;
    const int64_t dmax = P->h.depth - 1, levels = P->tp.n_levels, K = S.KC * kRec;
    const int64_t ticks_ready = std::min(rows_ready, S.T) * S.nsub;
    for (;;) {
        bool progressed = false;
        // one batch of 128 tick-rows -> records; a record slot is recycled only after every tick-row it can hold has left.
        // Lateral rows and boundary sub-steps (the ghost series of a partitioned network) advance separately: a ghost in a
        // tile of level l at lag L is first read (l K + L) ticks into the schedule, so the boundary may trail the rows.
        // Batch j writes, for a position of lag L, the records whose last tick-row lies in the batch: chunks up to
        // (128 (j + 1) + L) / 16.  One ring revolution earlier that slot held the same position's tick-rows up to
        // 128 (j + 1) - 16 rec_chunks + 15, whatever L is: those must have left.
        auto slot_free = [&](int64_t j) {
            const int64_t must_have_left = kRecRows * (j + 1) + kRec - kRec * S.rec_chunks;
            return must_have_left <= 0 || S.ticks_stored >= std::min(S.total, must_have_left);
        };
        if (S.has_in && S.in_batches < S.n_in_batches && ticks_ready >= std::min(kRecRows * (S.in_batches + 1), S.total) && slot_free(S.in_batches)) {
            launch_rec_permute(P, true, S.in_batches);
            ++S.in_batches;
            progressed = true;
        }
        if (P->n_ghost > 0 && S.ghost_batches < S.n_in_batches && (!S.has_in || S.ghost_batches < S.in_batches) &&      // after the lateral batch: that one writes zeros into the ghosts' records
            ghost_ready >= std::min(kRecRows * (S.ghost_batches + 1), S.total) && slot_free(S.ghost_batches)) {
            launch_ghost_permute(P, S.ghost_batches);
            ++S.ghost_batches;
            progressed = true;
        }
        auto loaded_ticks = [&](int64_t batches) { return batches >= S.n_in_batches ? S.total : std::max<int64_t>(0, kRecRows * batches - 15); };
        const int64_t have = S.has_in ? loaded_ticks(S.in_batches) : S.total;
        const int64_t have_ghost = P->n_ghost > 0 ? loaded_ticks(S.ghost_batches) : S.total;
        S.rows_loaded = have / S.nsub;
        // launch d runs macro-chunk d of the tiles of level 0: ticks below (d + 1) K need the tick-rows below that
        int64_t launched = 0;
        const int64_t batch = std::max<int64_t>(1, kRecRows / K);
        while (S.diag < S.n_diags && launched < batch) {
            const int64_t need_ticks = std::min((S.diag + 1) * K, S.total);
            if (have < need_ticks) break;
            if (have_ghost < std::min(std::max<int64_t>(0, (S.diag + 1) * K - S.ghost_slack), S.total)) break;
            // the tasks of this launch overwrite records in place: nothing they write may still be waiting to leave from
            // one ring revolution earlier (their chunks are at most (d + 1) KC - 1)
            const int64_t top = std::min(S.diag + 1, S.n_macro) * S.KC - 1;
            if (top >= S.rec_chunks && S.ticks_stored < std::min(S.total, kRec * (top - S.rec_chunks + 1))) break;
            int rc = session_launch_diag(P, S.diag);
            if (rc) return rc;
            ++S.diag; ++launched;
            progressed = true;
        }
        // the tiles of the last level have finished macro-chunk diag - levels; every other tile is further along
        const int64_t m_done = S.diag - levels;
        int64_t done = 0;
        if (S.diag >= S.n_diags) done = S.total;
        else if (m_done >= 0) done = std::max<int64_t>(0, (m_done + 1) * K - dmax);
        done = std::min(done, S.total);
        while (S.out_batches < S.n_out_batches && done >= std::min(kRecRows * (S.out_batches + 1), S.total) &&
               (std::min(kRecRows * (S.out_batches + 1), S.total) + S.nsub - 1) / S.nsub <= S.out_limit) {
            launch_rec_permute(P, false, S.out_batches);
            ++S.out_batches;
            S.ticks_stored = std::min(S.total, kRecRows * S.out_batches);
            progressed = true;
        }
        if (!progressed) break;
    }
    S.rows_stored = S.ticks_stored / S.nsub;
    if (S.diag >= S.n_diags) S.tau = S.total_ticks;
    if (export_ready) {
        const int64_t e = S.diag >= S.n_diags ? S.total : S.diag * K - S.export_skew;
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
    if (S.wave) return session_advance_tile(P, rows_ready, std::min(ghost_ready, S.total), export_ready);
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
    if (S.wave && S.ta.trace) {
        std::vector<long long> hbuf(16 * 4096);
        (void)hipStreamSynchronize(S.stream);
        (void)hipMemcpy(hbuf.data(), S.ta.trace, hbuf.size() * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("RR_WAVE_TRACE_FILE") ? getenv("RR_WAVE_TRACE_FILE") : "/tmp/wave_trace.txt", "w")) {
            for (int b = 0; b < 4096; ++b)
                if (hbuf[16 * b]) { fprintf(f, "%d", b); for (int k = 0; k < 15; ++k) fprintf(f, " %lld", hbuf[16 * b + k]); fprintf(f, "\n"); }
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
        const int64_t np = P->tp.np;
        hipLaunchKernelGGL(k_tile_state_in, grid1(np), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_si, d_q, P->d_tperm,
                           P->d_cfirst, P->d_ccnt, (int32_t)np);
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
        hipLaunchKernelGGL(k_tile_state_out, grid1(n), dim3(kBlock), 0, stream, d_q, (const double *)P->d_sq, P->d_tinv,
                           (int32_t)n);
        return;
    }
    hipLaunchKernelGGL(k_state_out, grid1(n), dim3(kBlock), 0, stream, d_q, (const double *)P->d_x, n,
                       P->d_lag, P->d_inv, (int32_t)n, total);
}

void parallel_copy(double *dst, const double *src, size_t count, int threads, std::vector<std::thread> &pool)
{
    const size_t per = ((count + threads - 1) / threads + 511) / 512 * 512;
    for (int t = 0; t < threads; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= count) break;
        pool.emplace_back([=] { std::memcpy(dst + o, src + o, std::min(per, count - o) * sizeof(double)); });
    }
}

int host_pipe_prepare(rr_plan *P)
{
    HostPipe &H = P->pipe;
    const int64_t n = P->h.n;
    // chunks of about half a gigabyte: long enough for the DMA engines to reach their rate, short enough to pipeline
    H.chunk_rows = std::max<int64_t>(16, std::min<int64_t>(4096, ((int64_t{1} << 29) / (n * 8) + 15) / 16 * 16));
    if (n * 8 * 64 <= (int64_t{1} << 30)) H.chunk_rows = std::max<int64_t>(H.chunk_rows, 64);
    H.ring_chunks = std::max<int64_t>(8, (2 * kRecRows + 15) / H.chunk_rows + 6);      // a batch of 128 rows + its 15-row overlap stays readable
    const int64_t pin_need = H.chunk_rows * n, dev_need = H.ring_chunks * H.chunk_rows * n;
    if (H.pin_cap < pin_need || H.dev_cap < dev_need) {
        H.destroy();
        for (int k = 0; k < HostPipe::kPinned; ++k) {
            if (hipHostMalloc((void **)&H.pin_in[k], (size_t)pin_need * 8, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&H.pin_out[k], (size_t)pin_need * 8, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError(); H.destroy();
                return fail(RR_E_ALLOC, "host pipeline: pinned staging buffers could not be allocated");
            }
        }
        if (hipMalloc((void **)&H.dev_in, (size_t)dev_need * 8) != hipSuccess || hipMalloc((void **)&H.dev_out, (size_t)dev_need * 8) != hipSuccess) {
            (void)hipGetLastError(); H.destroy();
            return fail(RR_E_ALLOC, "host pipeline: device staging rings could not be allocated");
        }
        H.pin_cap = pin_need; H.dev_cap = dev_need;
    }
    if (!H.s_h2d) { HIPCHK(hipStreamCreateWithFlags(&H.s_h2d, hipStreamNonBlocking)); HIPCHK(hipStreamCreateWithFlags(&H.s_d2h, hipStreamNonBlocking)); }
    return RR_OK;
}

// Routes T rows between host arrays (host_in may be NULL: channel-only) through the pipeline above.  State arrays are
// already on the device and the tile state is loaded; returns when host_out is complete.
int route_host_pipelined(rr_plan *P, Mode mode, int64_t T, int64_t nsub, const double *host_in, double *host_out, hipStream_t stream)
{
    int rc = host_pipe_prepare(P);
    if (rc) return rc;
    HostPipe &H = P->pipe;
    constexpr int kPinned = HostPipe::kPinned;
    const int64_t n = P->h.n, C = H.chunk_rows, NR = H.ring_chunks, nchunks = (T + C - 1) / C;
    auto grow = [&](std::vector<hipEvent_t> &v, size_t count) {
        while (v.size() < count) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false; v.push_back(e); }
        return true;
    };
    if (!grow(H.ev_h2d, (size_t)nchunks) || !grow(H.ev_d2h, (size_t)nchunks) || !grow(H.ev_adv, (size_t)4 * nchunks + 64))
        return fail(RR_E_HIP, "host pipeline: event creation failed");
    Rows io;
    io.dev_in = host_in ? H.dev_in : nullptr; io.rows_in = NR * C; io.dev_out = H.dev_out; io.rows_out = NR * C;
    rc = session_begin(P, mode, T, nsub, io, stream, nullptr, nullptr);
    if (rc) return rc;
    Session &S = P->ses;
    auto rows_of = [&](int64_t c) { return std::min(C, T - c * C); };
    std::vector<int64_t> adv_loaded;      // rows that were records after the a-th advance (ev_adv[a] marks it on the stream)
    int64_t filled = 0, h2d_issued = 0, d2h_issued = 0, copied_out = 0;      // chunks through each stage
    std::vector<std::thread> pool;
    auto bail = [&](int code, const std::string &msg) {
        for (auto &t : pool) t.join();
        (void)hipStreamSynchronize(H.s_h2d); (void)hipStreamSynchronize(H.s_d2h); (void)hipStreamSynchronize(stream);
        P->ses.open = false;
        return fail(code, msg);
    };
#define RR_PIPE(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return bail(RR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)
    // One iteration: the copy threads fill the next pinned input chunk and empty the oldest downloaded output chunk while
    // this thread enqueues the upload of the chunk filled last time, the routing it enables and the downloads it completes.
    while (copied_out < nchunks) {
        pool.clear();
        bool progressed = false;
        const bool fill = host_in && filled < nchunks && filled < h2d_issued + kPinned;
        if (fill) {
            if (filled >= kPinned) RR_PIPE(hipEventSynchronize(H.ev_h2d[filled - kPinned]));      // the buffer's previous chunk has left
            parallel_copy(H.pin_in[filled % kPinned], host_in + filled * C * n, (size_t)(rows_of(filled) * n), HostPipe::kCopyThreads, pool);
        }
        bool empty = false;
        if (copied_out < d2h_issued) {      // only a download that HAS arrived: waiting for one here would stall the uploads behind it
            const hipError_t q = hipEventQuery(H.ev_d2h[copied_out]);
            if (q == hipSuccess) empty = true;
            else if (q != hipErrorNotReady) return bail(RR_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
        if (empty) {
            parallel_copy(host_out + copied_out * C * n, H.pin_out[copied_out % kPinned], (size_t)(rows_of(copied_out) * n), HostPipe::kCopyThreads, pool);
        }
        // upload of a chunk filled earlier, into the ring slot whose previous occupant has become records
        if (host_in && h2d_issued < filled) {
            const int64_t c = h2d_issued;
            bool slot_ready = true;
            if (c >= NR) {
                const int64_t need = std::min(T, (c - NR + 1) * C);
                size_t a = 0;
                while (a < adv_loaded.size() && adv_loaded[a] < need) ++a;
                if (a < adv_loaded.size()) RR_PIPE(hipStreamWaitEvent(H.s_h2d, H.ev_adv[a], 0));
                else slot_ready = false;      // the routing has to get further first (see the advance below)
            }
            if (slot_ready) {
                RR_PIPE(hipMemcpyAsync(H.dev_in + (c % NR) * C * n, H.pin_in[c % kPinned], (size_t)(rows_of(c) * n) * 8, hipMemcpyHostToDevice, H.s_h2d));
                RR_PIPE(hipEventRecord(H.ev_h2d[c], H.s_h2d));
                RR_PIPE(hipStreamWaitEvent(stream, H.ev_h2d[c], 0));
                ++h2d_issued;
                progressed = true;
            }
        }
        // route what has arrived; output rows land in the device ring at row % (NR * C), so no batch may be written before
        // the rows it overwrites are on their way to the host
        if (S.tau < S.total_ticks || S.rows_stored < T) {
            const int64_t ready = host_in ? std::min(T, h2d_issued * C) : T;
            S.out_limit = std::min(T, d2h_issued * C) + NR * C;
            if (d2h_issued > 0) RR_PIPE(hipStreamWaitEvent(stream, H.ev_d2h[d2h_issued - 1], 0));
            const int64_t before_diag = S.diag, before_in = S.in_batches, before_out = S.out_batches;
            rc = session_advance(P, ready, S.total, nullptr);
            if (rc) { for (auto &t : pool) t.join(); P->ses.open = false; return rc; }
            if (S.diag != before_diag || S.in_batches != before_in || S.out_batches != before_out) {
                progressed = true;
                if (adv_loaded.size() < H.ev_adv.size() - 1) {
                    RR_PIPE(hipEventRecord(H.ev_adv[adv_loaded.size()], stream));
                    adv_loaded.push_back(S.rows_loaded);
                }
            }
        }
        // downloads of the output chunks that are complete; a pinned buffer is free once the copy threads have emptied it
        while (d2h_issued < nchunks && std::min(T, (d2h_issued + 1) * C) <= S.rows_stored && d2h_issued < copied_out + kPinned) {
            const int64_t k = d2h_issued;
            RR_PIPE(hipEventRecord(H.ev_adv.back(), stream));      // everything enqueued so far on the routing stream
            RR_PIPE(hipStreamWaitEvent(H.s_d2h, H.ev_adv.back(), 0));
            RR_PIPE(hipMemcpyAsync(H.pin_out[k % kPinned], H.dev_out + (k % NR) * C * n, (size_t)(rows_of(k) * n) * 8, hipMemcpyDeviceToHost, H.s_d2h));
            RR_PIPE(hipEventRecord(H.ev_d2h[k], H.s_d2h));
            ++d2h_issued;
            progressed = true;
        }
        for (auto &t : pool) t.join();
        if (fill) ++filled;
        if (empty) ++copied_out;
        if (!progressed && !fill && !empty) {
            if (copied_out < d2h_issued) RR_PIPE(hipEventSynchronize(H.ev_d2h[copied_out]));      // nothing else to do but wait for it
            else return bail(RR_E_STATE, "host pipeline: no stage can make progress");
        }
    }
#undef RR_PIPE
    S.out_limit = std::numeric_limits<int64_t>::max();
    return session_end(P);
}

int rapid_like(rr_plan *P, Mode mode, double *q_t, const Rows &io_in, int64_t T, int64_t nsub, hipStream_t stream,
               bool q_on_host)
{
    const int64_t n = P->h.n;
    if (n == 0 || T == 0) return RR_OK;
    const Rows &io = io_in;
    const bool host_rows = io.host_in != nullptr || io.host_out != nullptr;
    // host rows reach the time-tiled kernel through the PCIe pipeline's device rings; where it does not apply they are
    // routed chunk by chunk by the streaming kernel
    if (!decide_wave(P, mode, T * nsub, false) && host_rows) decide_wave(P, mode, T * nsub, true);
    const bool piped = host_rows && P->wave_now;
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
    if (rc == RR_OK) rc = piped ? route_host_pipelined(P, mode, T, nsub, io.host_in, io.host_out, stream) : route_core(P, mode, T, nsub, io, stream);
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

// UnitMuskingum state (channel discharge and full discharge of the reaches that have upstream reaches) into the layout of the
// kernel this call runs, and back.
int unit_state_in(rr_plan *P, const double *d_qch, const double *d_qfull, hipStream_t stream)
{
    const int64_t n = P->h.n, ni = (int64_t)P->h.inner_pos.size();
    hipError_t e0 = hipSuccess;
    if (use_wave(P, Mode::Unit)) {   // q_full / q_ch scattered to params order (zeros on headwaters), then gathered position by position
        e0 = hipMemsetAsync(P->d_full, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(P->d_chan, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess && ni > 0)
            hipLaunchKernelGGL(k_unit_scatter, grid1(ni), dim3(kBlock), 0, stream, P->d_full, P->d_chan, d_qfull, d_qch, P->d_inner_idx, (int32_t)ni);
        if (e0 == hipSuccess)
            hipLaunchKernelGGL(k_tile_unit_state_in, grid1(P->tp.np), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_si, P->d_sqch,
                               (const double *)P->d_full, (const double *)P->d_chan, P->d_tperm, P->d_cfirst, P->d_ccnt, (int32_t)P->tp.np);
    } else {
        e0 = hipMemsetAsync(P->d_x, 0, 3 * n * sizeof(double), stream);
        if (e0 == hipSuccess && ni > 0)
            hipLaunchKernelGGL(k_unit_state_in, grid1(ni), dim3(kBlock), 0, stream, P->d_x, P->d_x + n, P->d_x + 2 * n,
                               P->d_qch, d_qch, d_qfull, P->d_inner_pos, (int32_t)ni);
    }
    return e0 == hipSuccess ? RR_OK : fail(RR_E_HIP, hipGetErrorString(e0));
}

void unit_state_out(rr_plan *P, double *d_qch, double *d_qfull, int64_t total, hipStream_t stream)
{
    const int64_t n = P->h.n, ni = (int64_t)P->h.inner_pos.size();
    if (ni == 0) return;
    if (use_wave(P, Mode::Unit))
        hipLaunchKernelGGL(k_tile_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                           (const double *)P->d_sq, (const double *)P->d_sqch, P->d_inner_idx, P->d_tinv, (int32_t)ni);
    else
        hipLaunchKernelGGL(k_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                           (const double *)P->d_x, n, (const double *)P->d_qch, P->d_lag, P->d_inner_pos, (int32_t)ni, total);
}

int unit_like(rr_plan *P, double *q_ch, double *q_full, const Rows &io_in, int64_t T, int64_t nsub,
              hipStream_t stream, bool q_on_host, double *d_q_final = nullptr, double *uh_state_inout = nullptr)
{
    const int64_t n = P->h.n, ni = (int64_t)P->h.inner_pos.size();
    if (n == 0 || T == 0) return RR_OK;
    const Rows &io = io_in;
    const bool host_rows = io.host_in != nullptr || io.host_out != nullptr;
    if (!decide_wave(P, Mode::Unit, T * nsub, false) && host_rows) decide_wave(P, Mode::Unit, T * nsub, true);
    const bool piped = host_rows && P->wave_now;
    double *d_qch = q_ch, *d_qfull = q_full, *tmp = nullptr;
    if (q_on_host) {
        int rc = dev_alloc(&tmp, 2 * std::max<int64_t>(ni, 1));
        if (rc) return rc;
        d_qch = tmp; d_qfull = tmp + std::max<int64_t>(ni, 1);
        hipError_t e = hipMemcpyAsync(d_qch, q_ch, ni * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_qfull, q_full, ni * sizeof(double), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return fail(RR_E_HIP, hipGetErrorString(e)); }
    }
    const bool wave = use_wave(P, Mode::Unit);
    int rc = unit_state_in(P, d_qch, d_qfull, stream);
    if (rc) { if (tmp) (void)hipFree(tmp); return rc; }
    if (rc == RR_OK) rc = piped ? route_host_pipelined(P, Mode::Unit, T, nsub, io.host_in, io.host_out, stream) : route_core(P, Mode::Unit, T, nsub, io, stream);
    if (rc == RR_OK && wave && d_q_final)      // every reach: a headwater's state is its last lateral inflow, an inner reach's q_full
        hipLaunchKernelGGL(k_tile_state_out, grid1(n), dim3(kBlock), 0, stream, d_q_final, (const double *)P->d_sq, P->d_tinv, (int32_t)n);
    if (rc == RR_OK && io.uh_kernel && uh_state_inout) {      // carry-over state of the fused convolution, in place, after every batch has read the old one
        const dim3 gt((unsigned)((n + kUhTailThreads - 1) / kUhTailThreads));
        const int32_t nks = (int32_t)io.uh_nks;
        if (nks <= 16) hipLaunchKernelGGL(k_uh_tail<16>, gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in, T, nks, n);
        else if (nks <= 48) hipLaunchKernelGGL(k_uh_tail<48>, gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in, T, nks, n);
        else hipLaunchKernelGGL(k_uh_tail<0>, gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in, T, nks, n);
    }
    if (rc == RR_OK && ni > 0) {
        unit_state_out(P, d_qch, d_qfull, T * nsub, stream);
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
    // carry-over state, in place, after the rows above have read the old one (same stream)
    const dim3 gt((unsigned)((n + kUhTailThreads - 1) / kUhTailThreads));
    if (n_ks <= 16) hipLaunchKernelGGL(k_uh_tail<16>, gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n);
    else if (n_ks <= 48) hipLaunchKernelGGL(k_uh_tail<48>, gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n);
    else hipLaunchKernelGGL(k_uh_tail<0>, gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n);
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
        void *ptrs[] = {P->d_child_ptr, P->d_lag, P->d_perm, P->d_inv, P->d_inner_pos, P->d_bidx, P->d_hwc, P->d_w, P->d_c1row_h, P->d_c2,
                        P->d_c1row, P->d_tc2, P->d_tc3, P->d_sq, P->d_ss, P->d_si, P->d_sqch, P->d_full, P->d_chan,
                        P->d_tile_ptr, P->d_tile_level, P->d_tile_lag_lo, P->d_tile_lag_hi, P->d_tlag, P->d_cfirst, P->d_xpos,
                        P->d_tperm, P->d_tinv, P->d_tbidx, P->d_inner_idx, P->d_ccnt, P->d_colmeta, P->d_ghostmeta, P->d_c4_params,
                        P->d_c3, P->d_c4, P->d_x, P->d_isum, P->d_qch, P->d_ring, P->d_stage, P->d_mrows,
                        P->d_slot_a[0], P->d_slot_a[1], P->d_slot_b[0], P->d_slot_b[1], P->d_m_index[0], P->d_m_index[1]};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        P->pipe.destroy();
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
    if (const char *e = getenv("RR_WAVE_K")) P->wave_K = std::max(kRec, atoi(e) / kRec * kRec);
    std::string err;
    int rc = rr::build_host_plan(n, csc_indptr, csc_indices, P->h, err);
    if (rc) { delete P; return fail(rc, err); }
    {   // time-tiled schedule: tiles of 512 positions, one per thread, two workgroups per CU (while the waves of one
        // wait to issue their record loads the other one ticks: measured 368 ms per year at 1M reaches against 391 with
        // one 1,024-thread workgroup and 382 with four 256-thread ones)
        P->wave_threads = 512;
        if (const char *e = getenv("RR_WAVE_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) P->wave_threads = v; }
        P->wave_ppt = 1;
        int32_t block = P->wave_threads;
        if (const char *e = getenv("RR_TILE_BLOCK")) block = std::max(8, std::min(block, atoi(e)));     // tests: many small tiles
        std::vector<int32_t> lag_of((size_t)n);
        for (int64_t i = 0; i < n; ++i) lag_of[i] = P->h.lag[P->h.inv[i]];
        rr::build_tile_plan(P->h.down, lag_of, block, P->tp);
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
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) P->dev_total_bytes = total_b;
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) P->cu_count = cus;
            if (const char *e2 = getenv("RR_TILE_SLOTS")) P->cu_count = std::max(1, atoi(e2));      // tests: few workgroups, many tiles each
        }
        for (int v = 0; v < 4; ++v)    // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel
            if (hipFuncSetAttribute((const void *)tile_kernel(P->wave_threads, (v & 1) != 0, (v & 2) != 0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)tile_lds_bytes(P->wave_threads)) != hipSuccess) {
                (void)hipGetLastError();
                P->wave_enabled = false;
            }
        for (int v = 0; v < 2; ++v)
            (void)hipFuncSetAttribute((const void *)rec_in_uh_kernel(v != 0, 64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rec_in_uh_lds_bytes(64));
        (void)hipGetLastError();
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
        if (!rc && P->tp.ok) {      // tile layout of the time-tiled kernel
            const rr::TilePlan &TP = P->tp;
            const int64_t np = TP.np;
            std::vector<int2> cm(n);
            for (int64_t i = 0; i < n; ++i) cm[i] = make_int2(TP.inv[i], TP.lag[TP.inv[i]] & kLagMask);
            std::vector<int32_t> inner_idx;
            inner_idx.reserve(ni);
            for (int64_t i = 0; i < n; ++i) if (H.child_ptr[H.inv[i] + 1] > H.child_ptr[H.inv[i]]) inner_idx.push_back((int32_t)i);
            rc = dev_alloc(&P->d_colmeta, n);
            if (!rc) rc = dev_upload(P->d_colmeta, cm);
            if (!rc) rc = dev_alloc(&P->d_inner_idx, ni);
            if (!rc) rc = dev_upload(P->d_inner_idx, inner_idx);
            if (!rc) rc = dev_alloc(&P->d_tile_ptr, (int64_t)TP.tile_ptr.size());
            if (!rc) rc = dev_upload(P->d_tile_ptr, TP.tile_ptr);
            if (!rc) rc = dev_alloc(&P->d_tile_level, TP.n_tiles);
            if (!rc) rc = dev_upload(P->d_tile_level, TP.tile_level);
            if (!rc) rc = dev_alloc(&P->d_tile_lag_lo, TP.n_tiles);
            if (!rc) rc = dev_upload(P->d_tile_lag_lo, TP.tile_lag_lo);
            if (!rc) rc = dev_alloc(&P->d_tile_lag_hi, TP.n_tiles);
            if (!rc) rc = dev_upload(P->d_tile_lag_hi, TP.tile_lag_hi);
            if (!rc) rc = dev_alloc(&P->d_tlag, np);
            if (!rc) rc = dev_upload(P->d_tlag, TP.lag);
            if (!rc) rc = dev_alloc(&P->d_cfirst, np);
            if (!rc) rc = dev_upload(P->d_cfirst, TP.cfirst);
            if (!rc) rc = dev_alloc(&P->d_ccnt, np);
            if (!rc) rc = dev_upload(P->d_ccnt, TP.ccnt);
            if (!rc) rc = dev_alloc(&P->d_xpos, np);
            if (!rc) rc = dev_upload(P->d_xpos, TP.xpos);
            if (!rc) rc = dev_alloc(&P->d_tperm, np);
            if (!rc) rc = dev_upload(P->d_tperm, TP.perm);
            if (!rc) rc = dev_alloc(&P->d_tinv, n);
            if (!rc) rc = dev_upload(P->d_tinv, TP.inv);
            if (!rc) rc = dev_alloc(&P->d_tbidx, np);
            if (!rc) rc = dev_alloc(&P->d_c1row, np);
            if (!rc) rc = dev_alloc(&P->d_tc2, np);
            if (!rc) rc = dev_alloc(&P->d_tc3, np);
            if (!rc) rc = dev_alloc(&P->d_sq, np);
            if (!rc) rc = dev_alloc(&P->d_ss, np);
            if (!rc) rc = dev_alloc(&P->d_si, np);
            if (!rc) rc = dev_alloc(&P->d_sqch, np);
            if (!rc) rc = dev_alloc(&P->d_full, n);
            if (!rc) rc = dev_alloc(&P->d_chan, n);
        }
        if (!rc) rc = dev_alloc(&P->d_w, n);
        if (!rc) rc = dev_alloc(&P->d_c1row_h, n);
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

int rr_plan_tile_info(const rr_plan *P, int64_t info[8])
{
    if (!P || !info) return fail(RR_E_INVALID, "rr_plan_tile_info: null argument");
    const rr::TilePlan &T = P->tp;
    info[0] = T.ok ? 1 : 0; info[1] = T.block; info[2] = T.np; info[3] = T.n_ghost; info[4] = T.n_tiles; info[5] = T.n_levels;
    info[6] = P->wave_threads; info[7] = 0;
    return RR_OK;
}

int rr_plan_tile_layout(const rr_plan *P, int32_t *tile_ptr, int32_t *tile_level, int32_t *perm, int32_t *lag, int32_t *cfirst,
                        uint32_t *ccnt, int32_t *xpos)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_tile_layout: null plan");
    const rr::TilePlan &T = P->tp;
    if (!T.ok) return fail(RR_E_UNSUPPORTED, "rr_plan_tile_layout: this network does not tile (a reach has more upstream reaches than a tile holds)");
    auto copy = [](auto *dst, const auto &src) { if (dst && !src.empty()) std::memcpy(dst, src.data(), src.size() * sizeof(src[0])); };
    copy(tile_ptr, T.tile_ptr); copy(tile_level, T.tile_level); copy(perm, T.perm); copy(lag, T.lag); copy(cfirst, T.cfirst);
    copy(ccnt, T.ccnt); copy(xpos, T.xpos);
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
    if (!rc) rc = dev_upload(P->d_c1row_h, c1row);
    if (!rc && P->tp.ok) {      // the same in tile order; a ghost computes nothing
        const rr::TilePlan &TP = P->tp;
        std::vector<double> t1(TP.np, 0.0), t2(TP.np, 0.0), t3(TP.np, 0.0);
        for (int64_t p = 0; p < TP.np; ++p) {
            if (TP.lag[p] & kTileGhostBit) continue;
            const int32_t i = TP.perm[p];
            t1[p] = c1row[H.inv[i]]; t2[p] = c2[i]; t3[p] = c3[i];
        }
        rc = dev_upload(P->d_c1row, t1);
        if (!rc) rc = dev_upload(P->d_tc2, t2);
        if (!rc) rc = dev_upload(P->d_tc3, t3);
    }
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
    prof[9] = P->ses.wave ? (double)(P->ses.KC * kRec) : 1.0;
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
    int64_t gmin = H.depth, emax = 0;
    for (int64_t g = 0; g < n_ghost; ++g) {
        const int64_t i = ghost_reaches[g];
        if (i < 0 || i >= n) return fail(RR_E_INVALID, "rr_plan_set_boundary: ghost reach out of range");
        const int32_t p = H.inv[i];
        if (lag[p] & kGhostBit) return fail(RR_E_INVALID, "rr_plan_set_boundary: a ghost reach is listed twice");
        lag[p] |= kGhostBit;
        bidx[p] = (int32_t)g;
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
    P->ghost_reach.assign(n_ghost, 0); P->export_reach.assign(n_export, 0);
    for (int64_t g = 0; g < n_ghost; ++g) P->ghost_reach[g] = (int32_t)ghost_reaches[g];
    for (int64_t e = 0; e < n_export; ++e) P->export_reach[e] = (int32_t)export_reaches[e];
    if (P->tp.ok) {      // the same flags and slots in the tile layout
        const rr::TilePlan &TP = P->tp;
        std::vector<int32_t> tlag(TP.lag), tbidx(TP.np, 0);
        for (int64_t g = 0; g < n_ghost; ++g) { const int32_t p = TP.inv[ghost_reaches[g]]; tlag[p] |= kGhostBit; tbidx[p] = (int32_t)g; }
        for (int64_t e = 0; e < n_export; ++e) { const int32_t p = TP.inv[export_reaches[e]]; tlag[p] |= kExportBit; tbidx[p] = (int32_t)e; }
        rc = dev_upload(P->d_tlag, tlag);
        if (!rc) rc = dev_upload(P->d_tbidx, tbidx);
        if (P->d_ghostmeta) { (void)hipFree(P->d_ghostmeta); P->d_ghostmeta = nullptr; }
        std::vector<int2> gm((size_t)n_ghost);
        for (int64_t g = 0; g < n_ghost; ++g) { const int32_t p = TP.inv[ghost_reaches[g]]; gm[g] = make_int2(p, TP.lag[p] & kLagMask); }
        if (!rc) rc = dev_alloc(&P->d_ghostmeta, n_ghost);
        if (!rc) rc = dev_upload(P->d_ghostmeta, gm);
        if (rc) return rc;
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
    decide_wave(P, mode, T * nsub, false);
    if (P->h.n > 0 && T > 0) {
        rc = launch_state_in(P, mode, q_t, (hipStream_t)stream);
        if (rc) return rc;
    }
    return session_begin(P, mode, T, nsub, io, (hipStream_t)stream, ghost_series, export_series);
}

int rr_stream_begin_unit(rr_plan *P, const double *q_ch, const double *q_full, const double *lateral, int64_t lat_rows,
                         double *discharge, int64_t out_rows, int64_t T, int64_t nsub, const double *ghost_series,
                         double *export_series, void *stream)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    const int64_t ni = (int64_t)P->h.inner_pos.size();
    if (P->h.n > 0 && T > 0 && (!lateral || !discharge || lat_rows < 1 || out_rows < 1 || (ni > 0 && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_stream_begin_unit: null array or empty row count");
    Rows io; io.dev_in = lateral; io.rows_in = lat_rows; io.dev_out = discharge; io.rows_out = out_rows;
    if (P->ses.open) return fail(RR_E_STATE, "a routing call is already open on this plan (rr_stream_end_unit it first)");
    decide_wave(P, Mode::Unit, T * nsub, false);
    if (P->h.n > 0 && T > 0) {
        rc = unit_state_in(P, q_ch, q_full, (hipStream_t)stream);
        if (rc) return rc;
    }
    return session_begin(P, Mode::Unit, T, nsub, io, (hipStream_t)stream, ghost_series, export_series);
}

int rr_stream_end_unit(rr_plan *P, double *q_ch, double *q_full)
{
    int rc = need_device(P);
    if (rc) return rc;
    if (P->ses.open && P->ses.mode != Mode::Unit) return fail(RR_E_STATE, "rr_stream_end_unit: the open call is not a UnitMuskingum call");
    const int64_t total = P->ses.total;
    hipStream_t stream = P->ses.stream;
    rc = session_end(P);
    if (rc) return rc;
    if (P->h.n > 0 && total > 0 && q_ch && q_full) unit_state_out(P, q_ch, q_full, total, stream);
    return RR_OK;
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

// float32 output fused into the record pass (k_rec_out): applies when the call is time-tiled and factor x sub-steps divides
// the 128 tick-rows of a batch; otherwise RR_E_UNSUPPORTED and the caller uses the float64 form + rr_resample_cast_dev.
static int f32_output_applies(rr_plan *P, Mode mode, int64_t T, int64_t nsub, int64_t factor)
{
    if (factor < 1 || T % factor != 0) return fail(RR_E_INVALID, "float32 output: the number of rows must be a multiple of factor >= 1");
    if (kRecRows % (factor * nsub) != 0) return fail(RR_E_UNSUPPORTED, "float32 output: factor x sub-steps must divide 128");
    if (!decide_wave(P, mode, T * nsub, false)) return fail(RR_E_UNSUPPORTED, "float32 output needs the time-tiled kernel, which this call does not get");
    return RR_OK;
}

int rr_rapid_route_f32_dev(rr_plan *P, double *q_t, const double *qlateral, int64_t ql_rows, float *discharge32, int64_t T,
                           int64_t nsub, int64_t factor, void *stream)
{
    int rc = check_route_args(P, true, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!q_t || !qlateral || !discharge32 || ql_rows < 1))
        return fail(RR_E_INVALID, "rr_rapid_route_f32_dev: null array or empty row count");
    if (P->h.n == 0 || T == 0) return RR_OK;
    rc = f32_output_applies(P, Mode::Rapid, T, nsub, factor);
    if (rc) return rc;
    Rows io; io.dev_in = qlateral; io.rows_in = ql_rows; io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor;
    return rapid_like(P, Mode::Rapid, q_t, io, T, nsub, (hipStream_t)stream, false);
}

int rr_muskingum_route_f32_dev(rr_plan *P, double *q_t, float *discharge32, int64_t n_out, int64_t n_per_out, void *stream)
{
    int rc = check_route_args(P, false, n_out, n_per_out);
    if (rc) return rc;
    if (P->h.n > 0 && n_out > 0 && (!q_t || !discharge32)) return fail(RR_E_INVALID, "rr_muskingum_route_f32_dev: null array");
    if (P->h.n == 0 || n_out == 0) return RR_OK;
    rc = f32_output_applies(P, Mode::Muskingum, n_out, n_per_out, 1);
    if (rc) return rc;
    Rows io; io.dev_out32 = discharge32; io.out_factor = 1; io.rows_out = n_out;
    return rapid_like(P, Mode::Muskingum, q_t, io, n_out, n_per_out, (hipStream_t)stream, false);
}

int rr_unit_route_f32_dev(rr_plan *P, double *q_ch, double *q_full, const double *conv, int64_t conv_rows, float *discharge32,
                          int64_t T, int64_t nsub, int64_t factor, void *stream)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    if (P->h.n > 0 && T > 0 && (!conv || !discharge32 || conv_rows < 1 || (!P->h.inner_pos.empty() && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_unit_route_f32_dev: null array or empty row count");
    if (P->h.n == 0 || T == 0) return RR_OK;
    rc = f32_output_applies(P, Mode::Unit, T, nsub, factor);
    if (rc) return rc;
    Rows io; io.dev_in = conv; io.rows_in = conv_rows; io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor;
    return unit_like(P, q_ch, q_full, io, T, nsub, (hipStream_t)stream, false);
}

int rr_rapid_route_runoff_dev(rr_plan *P, double *q_t, int64_t n_points, const int32_t *indptr, const int32_t *indices,
                              const double *weights, const void *runoff, int runoff_is_f32, int64_t stride_t, int64_t stride_p,
                              const double *area, int flags, double *discharge, float *discharge32, int64_t factor, int64_t T,
                              void *stream)
{
    int rc = check_route_args(P, true, T, 1);
    if (rc) return rc;
    const bool f32 = discharge32 != nullptr;
    if (P->h.n > 0 && T > 0 && (!q_t || !indptr || !indices || !weights || !runoff || (!discharge && !discharge32) || (discharge && discharge32)))
        return fail(RR_E_INVALID, "rr_rapid_route_runoff_dev: null array, or both or neither output");
    if (n_points < 0 || stride_t < 0 || stride_p < 0) return fail(RR_E_INVALID, "rr_rapid_route_runoff_dev: negative size or stride");
    if (P->h.n == 0 || T == 0) return RR_OK;
    if (f32) { rc = f32_output_applies(P, Mode::Rapid, T, 1, factor); if (rc) return rc; }
    else if (!decide_wave(P, Mode::Rapid, T, false)) return fail(RR_E_UNSUPPORTED, "rr_rapid_route_runoff_dev needs the time-tiled kernel, which this call does not get");
    RunoffArgs ga{indptr, indices, weights, area, runoff, stride_t, stride_p, (int32_t)flags, runoff_is_f32 ? 1 : 0};
    Rows io; io.dev_in = P->d_c4_params; io.rows_in = 1;      // (no lateral rows: dev_in only has to be non-NULL for the executor)
    if (f32) { io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor; }
    else { io.dev_out = discharge; io.rows_out = T; }
    io.runoff = &ga;
    return rapid_like(P, Mode::Rapid, q_t, io, T, 1, (hipStream_t)stream, false);
}

int rr_unit_route_uh_dev(rr_plan *P, double *q_ch, double *q_full, double *q_final, const double *uh_kernel, double *uh_state,
                         int64_t n_ks, const double *depth, double *discharge, float *discharge32, int64_t factor, int64_t T,
                         int64_t nsub, void *stream)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    const bool f32 = discharge32 != nullptr;
    if (P->h.n > 0 && T > 0 && (!depth || !uh_kernel || !uh_state || (!discharge && !discharge32) || (discharge && discharge32) || n_ks < 1 ||
                                (!P->h.inner_pos.empty() && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_unit_route_uh_dev: null array, both or neither output, or n_ks < 1");
    if (P->h.n == 0 || T == 0) return RR_OK;
    if (n_ks > kUhFusedMaxTaps) return fail(RR_E_UNSUPPORTED, "rr_unit_route_uh_dev: more than 64 kernel steps: convolve with rr_uh_convolve_dev, then rr_unit_route_dev");
    if (f32) { rc = f32_output_applies(P, Mode::Unit, T, nsub, factor); if (rc) return rc; }
    else if (!decide_wave(P, Mode::Unit, T * nsub, false)) return fail(RR_E_UNSUPPORTED, "rr_unit_route_uh_dev needs the time-tiled kernel, which this call does not get");
    Rows io; io.dev_in = depth; io.rows_in = T;
    if (f32) { io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor; }
    else { io.dev_out = discharge; io.rows_out = T; }
    io.uh_kernel = uh_kernel; io.uh_state = uh_state; io.uh_nks = n_ks;
    return unit_like(P, q_ch, q_full, io, T, nsub, (hipStream_t)stream, false, q_final, uh_state);
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

int rr_copy_bandwidth(int device, int64_t bytes, int reps, double *gbps)
{
    if (!gbps || bytes < 1024 || reps < 1) return fail(RR_E_INVALID, "rr_copy_bandwidth: bad argument");
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_copy_bandwidth: no such HIP device");
    HIPCHK(hipSetDevice(device));
    const int64_t count = bytes / 16;
    double2 *a = nullptr, *b = nullptr;
    int rc = dev_alloc(&a, count);
    if (!rc) rc = dev_alloc(&b, count);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    if (!rc) {
        hipError_t e = hipMemset(a, 0, (size_t)count * 16);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        const dim3 g((unsigned)std::min<int64_t>((count + kBlock - 1) / kBlock, 256 * 16));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_copy16, g, dim3(kBlock), 0, nullptr, (const double2 *)a, b, count);      // warm-up
            e = hipEventRecord(e0, nullptr);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16, g, dim3(kBlock), 0, nullptr, (const double2 *)a, b, count);
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            // the runtime's own device-to-device copy: whichever of the two is faster is "the achievable rate"
            float ms2 = 0.f;
            if (e == hipSuccess) e = hipMemcpyAsync(b, a, (size_t)count * 16, hipMemcpyDeviceToDevice, nullptr);
            if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
            for (int r = 0; r < reps && e == hipSuccess; ++r) e = hipMemcpyAsync(b, a, (size_t)count * 16, hipMemcpyDeviceToDevice, nullptr);
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms2, e0, e1);
            if (e == hipSuccess && ms2 > 0.f && ms2 < ms) ms = ms2;
        }
        if (e != hipSuccess) rc = fail(RR_E_HIP, hipGetErrorString(e));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc) return rc;
    *gbps = 2.0 * (double)count * 16.0 * reps / ((double)ms * 1e-3) / 1e9;      // bytes read + bytes written
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
