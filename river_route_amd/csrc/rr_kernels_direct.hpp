// rr_kernels_direct.hpp -- the direct row path: routing straight from and to the caller's (time, reach) rows where the params order
// numbers small subtrees contiguously (rr_plan.hpp: DirectPlan).
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

// k_tile moves every lateral value and every discharge through the record ring, and the two record passes move them once more
// each: 51.9 B per reach-step where 16 are compulsory.  They exist because a tile's positions are scattered over the params
// order.  In a depth-first post-order a subtree is a run of consecutive columns, so a tile can be a column RANGE [c0, c0 + nc):
// lane = column, and the tile reads and writes row segments itself.
//
// Lane with lag L routes row r at tick r + L (section 3: its upstream lanes are one tick ahead), so at one tick the lanes of a
// tile work on different rows: delay = L - (smallest lag of the tile) rows behind the first.  A row segment is loaded once, one
// coalesced 8-byte load per lane PF ticks ahead, and parked in an LDS window F[slot][lane] of span + 1 rows (span = largest delay):
// lane reads F[own slot] when its turn comes, writes its discharge back in place, and when the slowest lane has done a row the
// segment is stored, coalesced.  Each lane touches only its own column of the window: no synchronisation beyond the tick's barrier.
//
// One task = one tile x K rows [m K, (m + 1) K): lanes start one after the other (K + span ticks, the first and last `span` of them
// with idle lanes), so the window is empty at both ends and nothing but the lanes' last discharge is carried from task to task.
// A tile holds whole small subtrees, so it has no ghost, no level and no dependency on another tile: every launch runs every tile.
//
// The skeleton (reaches with large or tall subtrees: 4-5 %) keeps records and k_tile.  Its columns lie between the subtrees'
// columns (HOLES): the hole lane forwards its column's lateral inflow, scaled, into the skeleton position's record; the outlet lane
// of a small subtree writes its discharge into the record of the ghost that mirrors it in the skeleton; k_rec_out, given the
// holes' columns, patches the output rows from the skeleton's records afterwards (the direct tile stores 0 there).
struct DirectTile { int32_t c0, nc, lag_lo, span; };
struct DirectArgs {
    const DirectTile *tiles;
    int32_t n_tiles;
    const int4 *lane;           // per column {delay | kDirectHole, upstream lanes (3 x 10 bits), xinfo, lag}
    const double *coef;         // per column {c1row, c2, c3, c4dt}
    double *q;                  // per column: carried discharge
    const double *in;           // lateral rows (in_rows x n), read cyclically
    double *out;                // discharge rows (out_rows x n), written cyclically
    int64_t n;
    uint32_t in_rows, out_rows;
    double *rec;                // the skeleton's record ring [chunks][np][16] (k_tile's layout)
    uint32_t rec_chunks;
    int32_t np;
    int32_t m, K, total;        // this launch: rows [m K, min((m + 1) K, total))
};
constexpr int32_t kDirectHoleBit = rr::kDirectHole;
constexpr int kDirectLanes = 256, kDirectAhead = 16;
constexpr size_t direct_lds_bytes(int window_rows) { return (size_t)(2 * (kDirectLanes + kTilePad) + (int64_t)window_rows * kDirectLanes) * sizeof(double); }
constexpr int kDirectMaxWindow = 72;      // rows: 2 x 258 + 72 x 256 doubles = 151.6 KB of the CU's 160

__device__ __forceinline__ double load_f64(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x2 bits = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    double v;
    __builtin_memcpy(&v, &bits, sizeof v);
    return v;
}

template <int TH, int PF>
__global__ __launch_bounds__(TH, 1) void k_direct(const DirectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int THP = TH + kTilePad;
    static_assert(PF % 2 == 0, "the parity of a tick is the parity of its place in a chunk");
    char *const X = reinterpret_cast<char *>(lds);                  // [2][THP] discharges of the last two ticks, each followed by a slot that holds 0.0
    char *const F = reinterpret_cast<char *>(lds + 2 * THP);        // [span + 1][TH] the row window
    const int tid = threadIdx.x;
    if (tid < 2) lds[tid * THP + TH] = 0.0;
    const int32_t r0 = a.m * a.K, nrows = min(a.K, a.total - r0);
    const uint32_t row_bytes = (uint32_t)a.n * 8u;                  // n < 2^29 (choose_schedule)

    for (int32_t t = (int32_t)blockIdx.x; t < a.n_tiles; t += (int32_t)gridDim.x) {
        const DirectTile tm = a.tiles[t];
        const bool live = tid < tm.nc;
        const int32_t col = tm.c0 + (live ? tid : 0);
        const int4 lm = a.lane[col];
        const bool hole = !live || (lm.x & kDirectHoleBit) != 0;
        const int32_t delta = hole ? 0x40000000 : lm.x;              // a hole is never active
        const int32_t xpos = live ? lm.z : -1;                        // hole: its skeleton position; outlet of a subtree: its ghost there
        const double c1 = hole ? 0.0 : a.coef[4 * (int64_t)col], c2 = hole ? 0.0 : a.coef[4 * (int64_t)col + 1],
                     c3 = hole ? 0.0 : a.coef[4 * (int64_t)col + 2], c4 = a.coef[4 * (int64_t)col + 3];
        const double q0 = hole ? 0.0 : a.q[col];
        const int32_t up0_b = (lm.y & 0x3FF) == 0x3FF || hole ? TH * 8 : (lm.y & 0x3FF) * 8,
                      up1_b = ((lm.y >> 10) & 0x3FF) == 0x3FF || hole ? TH * 8 : ((lm.y >> 10) & 0x3FF) * 8,
                      up2_b = ((lm.y >> 20) & 0x3FF) == 0x3FF || hole ? TH * 8 : ((lm.y >> 20) & 0x3FF) * 8;
        const uint32_t voff = live ? (uint32_t)col * 8u : kDropAccess;
        const int32_t span = tm.span, wrap = (span + 1) * (TH * 8);   // bytes of the window in use
        // the record slot of this lane's tick: row + lag, i.e. (r0 + k - delay) + lag at tick k (a hole forwards the row that arrives: no delay)
        const bool sends = xpos >= 0;
        uint32_t xchunk = 0, xslot = 0;
        if (sends) {
            const uint32_t tg = (uint32_t)(r0 + lm.w - (hole ? 0 : delta));
            xslot = tg & 15u;
            xchunk = (tg >> 4) % a.rec_chunks;
        }
        const bool wave_sends = __builtin_amdgcn_ballot_w64(sends) != 0;

        __syncthreads();      // every wave has left the previous tile
        *reinterpret_cast<double *>(X + tid * 8) = q0;
        *reinterpret_cast<double *>(X + THP * 8 + tid * 8) = q0;
        // rows r0 ... r0 + PF - 1 on their way
        double P[PF];
        uint32_t rin = (uint32_t)r0 % a.in_rows;
        auto request = [&](int32_t k) {      // row r0 + k, or nothing past the task's rows
            const __amdgpu_buffer_rsrc_t src = make_rsrc(a.in + (int64_t)rin * a.n, row_bytes);
            const double v = load_f64(src, k < nrows ? voff : kDropAccess);
            rin = rin + 1 == a.in_rows ? 0 : rin + 1;
            return v;
        };
#pragma unroll
        for (int j = 0; j < PF; ++j) P[j] = request(j);
        uint32_t rout = (uint32_t)r0 % a.out_rows;
        int32_t in_b = 0;                                              // window slot of the row that arrives this tick, in bytes
        int32_t own_b = delta >= 0x40000000 ? 0 : (delta == 0 ? 0 : wrap - delta * (TH * 8));      // ... of the row this lane routes this tick
        double s_prev = 0.0;
        __syncthreads();

        auto ticks = [&](auto tested, int32_t k0) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int32_t k = k0 + s;
                const int prev = ((s + 1) & 1) * (THP * 8), cur = (s & 1) * (THP * 8);
                // the row that arrives: into the window, the next one requested
                const double lat_in = P[s] * c4;
                P[s] = request(k + PF);
                *reinterpret_cast<double *>(F + in_b + tid * 8) = hole ? 0.0 : lat_in;
                // this lane's tick: _numba_kernels.py:63-84 in gather form, the arithmetic of k_tile's short tick
                const double q_old = *reinterpret_cast<const double *>(X + prev + tid * 8);
                const double s_cur = (*reinterpret_cast<const double *>(X + prev + up0_b) + *reinterpret_cast<const double *>(X + prev + up1_b)) +
                                     *reinterpret_cast<const double *>(X + prev + up2_b);
                double *mine = reinterpret_cast<double *>(F + own_b + tid * 8);
                const double lat = *mine;
                double qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, q_old, lat)));
                bool active = true;
                if (decltype(tested)::value) {
                    active = (uint32_t)(k - delta) < (uint32_t)nrows;
                    qk = active ? qk : q_old;
                    *mine = active ? qk : lat;
                } else {
                    *mine = qk;
                }
                s_prev = s_cur;
                *reinterpret_cast<double *>(X + cur + tid * 8) = qk;
                if (wave_sends) {      // wave-uniform: most waves hold neither a hole nor an outlet
                    const bool now = sends && (hole ? k < nrows : active);
                    if (now) a.rec[((int64_t)xchunk * a.np + xpos) * kRec + xslot] = hole ? lat_in : qk;
                    xslot = (xslot + 1) & 15u;      // the slot moves with the tick
                    if (xslot == 0) xchunk = xchunk + 1 == a.rec_chunks ? 0 : xchunk + 1;
                }
                // the row the slowest lane has just done leaves: it sits where the next row will arrive
                in_b = in_b + TH * 8 == wrap ? 0 : in_b + TH * 8;
                own_b = own_b + TH * 8 == wrap ? 0 : own_b + TH * 8;
                const double done = *reinterpret_cast<const double *>(F + in_b + tid * 8);
                const bool leaves = k >= span && k - span < nrows;
                const __amdgpu_buffer_rsrc_t dst = make_rsrc(a.out + (int64_t)rout * a.n, row_bytes);
                store_f64(dst, leaves ? voff : kDropAccess, done > 0.0 ? done : 0.0);      // the reference's clip at zero (_numba_kernels.py:84)
                if (leaves) rout = rout + 1 == a.out_rows ? 0 : rout + 1;
                barrier_lds();
            }
        };
        const int32_t n_ticks = nrows + span;
        for (int32_t k0 = 0; k0 < n_ticks; k0 += PF) {
            if (k0 >= span && k0 + PF <= nrows) ticks(std::false_type(), k0);      // every lane busy on every tick of the chunk
            else ticks(std::true_type(), k0);
        }
        if (!hole) a.q[col] = *reinterpret_cast<const double *>(X + THP * 8 + tid * 8);      // PF is even: the last tick wrote buffer 1
    }
}

}  // namespace
