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
// tile work on different rows: delay = L - (smallest lag of the tile) rows behind the first.  A row segment is loaded once,
// coalesced, PF ticks ahead, and parked in an LDS window F[slot][lane] of span + 3 rows (span = largest delay): a lane reads
// F[its slot] when its turn comes and writes its discharge back in place; when the slowest lane has done a row the segment is
// stored, coalesced.  Each lane touches only its own column of the window.
//
// One task = one tile x K rows [m K, (m + 1) K): lanes start one after the other (K + span ticks, the first and last `span` of them
// with idle lanes), so the window is empty at both ends and nothing but the lanes' last discharge is carried from task to task.
// A tile holds whole small subtrees, so it has no ghost, no level and no dependency on another tile: every launch runs every tile.
//
// A tick is bound by instruction issue and LDS latency, not by arithmetic: one wave per SIMD issues a vector instruction every
// ~4 cycles, and the first version -- every lane loading, routing, forwarding and storing -- ran 110 instructions and two LDS
// round trips per tick: 816 cycles.  So the workgroup is SPECIALISED: waves 0-3 route (lane = column: five LDS reads, five
// multiply-adds, two LDS writes), waves 4 and 5 bring the rows in (a 16-byte load per lane and tick, 32 rows ahead, scaled by c4dt
// into the window one tick before the first lane needs them), wave 6 takes finished rows out (one tick after the last lane wrote
// them), wave 7 forwards what the skeleton needs: eight instruction streams, two per SIMD, that meet at the tick's barrier.
// What a tick (0.195 us) answers to, measured build against build on one box (profiles/r04_direct_tick_ab.txt): heavier rows-in
// waves slow it down (63 instead of 38 instructions a tick: +34 %; 16 instead of 32 rows in flight: +11 %, a loaded memory latency
// of ~3.5 us), lighter ones do not speed it up (25 instructions, or 20 / 26 on four waves that take turns), nor do a lighter rows-out
// wave or wave 7, nor rows that lie in the Infinity Cache instead of HBM, nor LDS round trips taken off the I/O waves' path; two
// workgroups of 128 lanes per CU are slower than one of 256.  What is left is the routing waves' own recurrence: four LDS reads, two
// additions and three multiply-adds in a chain, two LDS writes, the wait for them and the barrier.
//
// The skeleton (reaches with large or tall subtrees: 5 %) keeps records and k_tile.  Its columns lie between the subtrees'
// columns (HOLES).  A hole's scaled lateral inflow (waves 4, 5) and the discharge of an outlet lane (a small subtree's last reach)
// go into small LDS rings, from which wave 7 writes whole records into the skeleton's record ring: the hole's own position, the
// ghost that mirrors the outlet there; k_rec_out, given the holes' columns, patches the output rows from the skeleton's records afterwards.
struct DirectTile { int32_t c0, nc, lag_lo, span; };      // span: levels between the tile's first and last lane
struct DirectArgs {
    const DirectTile *tiles;
    int32_t n_tiles;
    const int4 *lane;           // per column {level delay | kDirectHole, upstream lanes (3 x 10 bits), xinfo, lag}
    const double *coef;         // per column {c1row, c2, c3, c4dt}
    const int32_t *send_ptr;    // per tile: its senders in send_lane
    const int32_t *send_lane;   // lane | kDirectHole
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
constexpr int kDirectThreads = kDirectLanes + 8 * 64;      // four routing waves + four rows-in + two rows-out + two send: three waves per SIMD
constexpr int kDirectSlack = 6;           // window rows beyond twice the tile's span: the pair being routed by the last lanes, the pair parked
                                          // for the next step, the pair on its way out
constexpr int kDirectMaxWindow = 64;      // rows of the LDS window, 2 span + kDirectSlack
constexpr int kDirectMaxLevels = (kDirectMaxWindow - kDirectSlack) / 2 + 1;      // the tallest small subtree
constexpr int kDirectSenders = rr::kDirectSenders;      // per tile
// LDS in doubles: X[2][258][2]: the lanes' last two PAIRS of discharges | S[senders][32]: what the skeleton needs, a ring of two
// records (32 ticks) per sender on a multiple of its 256 bytes, filled by the lanes that make the values -- the outlet's routing
// lane, the hole's column in the rows-in waves -- at slot = tick % 32 (tick = row + lag, k_tile's record slot) | D[2][256]: a
// slot per lane that takes the store of a lane with nothing to send (no branch, no bank conflict) | F[window rows][256]
constexpr int kDirectStage = kDirectSenders * 2 * kRec + 2 * kDirectLanes;      // doubles
constexpr int kDirectX = (4 * (kDirectLanes + kTilePad) + 2 * kRec - 1) / (2 * kRec) * (2 * kRec);      // doubles
constexpr size_t direct_lds_bytes(int window_rows)
{
    return (size_t)(kDirectX + kDirectStage + (int64_t)window_rows * kDirectLanes) * sizeof(double);
}
static_assert(direct_lds_bytes(kDirectMaxWindow) <= 160 * 1024, "the CU's LDS");

// the reference's clip at zero (_numba_kernels.py:84): x > 0 ? x : 0 in one instruction (the compiler's own max quiets its operand first)
__device__ __forceinline__ double clip0(double x)
{
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ double2 load_f64x2_(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    double2 v;
    __builtin_memcpy(&v, &bits, sizeof v);
    return v;
}

template <int PF>
__global__ __launch_bounds__(kDirectThreads, 1) void k_direct(const DirectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int TH = kDirectLanes, THP = TH + kTilePad;
    static_assert(PF == kRec && kDirectSenders == 64, "waves 10 and 11 serve four senders a lane each, one of them a step: a sender's turn comes every four steps (eight rows)");
    char *const X = reinterpret_cast<char *>(lds);                  // [2][THP] pairs of discharges: the rows of the last two steps, each followed by a lane that holds (0, 0)
    char *const F = reinterpret_cast<char *>(lds + kDirectX + kDirectStage);        // [2 span + 6][TH] the row window
    // (the wave's number in a scalar register: what depends on it stays scalar)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    const int role = wave < 4 ? 0 : (wave < 8 ? 1 : (wave < 10 ? 2 : 3));      // 0 route, 1 rows in (four waves), 2 rows out (two waves), 3 send (two waves)
    if (tid < 2) *reinterpret_cast<double2 *>(X + (tid * THP + TH) * 16) = make_double2(0.0, 0.0);
    const int32_t r0 = a.m * a.K, nrows = min(a.K, a.total - r0);
    const uint32_t row_bytes = (uint32_t)a.n * 8u;                  // n < 2^29 (choose_schedule)
    constexpr int kRowB = TH * 8;                                    // bytes of a window row
    constexpr int kPairB = THP * 16;                                 // bytes of one parity of X
    constexpr int kStageB = kDirectX * 8, kDummyB = kStageB + kDirectSenders * 2 * kRec * 8;      // byte offsets in LDS of S and of the dummy slots
    constexpr int kRingB = 2 * kRec * 8;                             // bytes of a sender's ring
    static_assert(kStageB % kRingB == 0, "a ring starts on a multiple of its size");
    constexpr int kSenderMask = 0x7F;

    for (int32_t t = (int32_t)blockIdx.x; t < a.n_tiles; t += (int32_t)gridDim.x) {
        const DirectTile tm = a.tiles[t];
        const int32_t span = tm.span, wrap = (2 * span + kDirectSlack) * kRowB;     // bytes of the window in use (an even number of rows: a pair never wraps)
        // TWO rows per barrier.  A lane that is d levels below the tile's first lanes runs 2 d rows behind them: at step j it routes
        // rows 2 j - 2 d and 2 j - 2 d + 1, one after the other in registers, with its upstream lanes' discharges of both rows --
        // routed one step earlier -- in hand: the chain LDS read -> arithmetic -> LDS write -> wait -> barrier, which is what a
        // step costs (profiles/r04_direct_tick_ab.txt), is paid once for two rows.  Step j: rows 2 j + 2 and 2 j + 3 are parked
        // (waves 4-7), rows 2 j - 2 d (+ 1) routed (waves 0-3), rows 2 j - 2 - 2 span (+ 1) leave (waves 8, 9), what the
        // skeleton needs is forwarded (wave 10); n_steps of them, in chunks of PF.
        const int32_t n_steps = (nrows - 1 + 2 * span) / 2 + 2;
        __syncthreads();      // every wave has left the previous tile

        if (role == 0) {
            // ---------------------------------------------------------------- routing lanes: lane = column
            const bool live = tid < tm.nc;
            const int32_t col = tm.c0 + (live ? tid : 0);
            const int4 lm = a.lane[col];
            const bool idle = !live || (lm.x & kDirectHoleBit) != 0;      // a hole's column only passes through the window
            const int32_t level = lm.x & rr::kDirectDelayMask;
            const int32_t delta = idle ? 0x40000000 : 2 * level;           // rows behind the tile's first lanes
            // an outlet that feeds the skeleton also puts its discharges into its sender ring, at slot (row + lag) % 32
            const int32_t sender = idle ? 0 : (lm.x >> rr::kDirectSenderShift) & kSenderMask;
            const int32_t ring_b = sender ? kStageB + (sender - 1) * kRingB : -1;
            const bool wave_sends = __builtin_amdgcn_ballot_w64(sender != 0) != 0;
            int32_t slot_b = ((r0 + lm.w - 2 * level) & 31) * 8;      // of the first row of step 0: tick = r0 + row + lag, row = -2 level
            const double c1 = idle ? 0.0 : a.coef[4 * (int64_t)col], c2 = idle ? 0.0 : a.coef[4 * (int64_t)col + 1], c3 = idle ? 0.0 : a.coef[4 * (int64_t)col + 2];
            const double q0 = idle ? 0.0 : a.q[col];
            const int32_t u0 = lm.y & 0x3FF, u1 = (lm.y >> 10) & 0x3FF, u2 = (lm.y >> 20) & 0x3FF;
            const int32_t up0_b = (u0 == 0x3FF || idle ? TH : u0) * 16, up1_b = (u1 == 0x3FF || idle ? TH : u1) * 16, up2_b = (u2 == 0x3FF || idle ? TH : u2) * 16;
            *reinterpret_cast<double2 *>(X + tid * 16) = make_double2(q0, q0);
            *reinterpret_cast<double2 *>(X + kPairB + tid * 16) = make_double2(q0, q0);
            // window slot of the first row this lane routes at step 0: (-delta) mod (2 span + 6), the lane's own column of it
            int32_t own_b = (idle || delta == 0 ? 0 : wrap - delta * kRowB) + tid * 8;
            double s_prev = 0.0, q_last = q0;      // the upstream sum and the lane's own discharge of the row before stay in registers
            __syncthreads();      // the discharges carried in, and rows 0, 1 in the window (waves 4-7)
            auto steps = [&](auto tested, int32_t k0) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const int prev = ((s + 1) & 1) * kPairB, cur = (s & 1) * kPairB;
                    // _numba_kernels.py:63-84 in gather form, the arithmetic of k_tile's short tick operation for operation, twice
                    const double2 v0 = *reinterpret_cast<const double2 *>(X + prev + up0_b), v1 = *reinterpret_cast<const double2 *>(X + prev + up1_b),
                                  v2 = *reinterpret_cast<const double2 *>(X + prev + up2_b);
                    double *mine = reinterpret_cast<double *>(F + own_b);
                    const double lat_a = mine[0], lat_b = mine[TH];
                    const double s_a = (v0.x + v1.x) + v2.x, s_b = (v0.y + v1.y) + v2.y;
                    double q_a = __builtin_fma(c1, s_a, __builtin_fma(c2, s_prev, __builtin_fma(c3, q_last, lat_a)));
                    bool act_a = true, act_b = true;
                    if (decltype(tested)::value) {      // a row outside the task: the lane keeps its discharge and hands the window slot back as it found it
                        const int32_t row_a = 2 * (k0 + s) - delta;
                        act_a = (uint32_t)row_a < (uint32_t)nrows; act_b = (uint32_t)(row_a + 1) < (uint32_t)nrows;
                        mine[0] = act_a ? q_a : lat_a;
                        q_a = act_a ? q_a : q_last;
                    } else {
                        mine[0] = q_a;
                    }
                    double q_b = __builtin_fma(c1, s_b, __builtin_fma(c2, s_a, __builtin_fma(c3, q_a, lat_b)));
                    if (decltype(tested)::value) {
                        mine[TH] = act_b ? q_b : lat_b;
                        q_b = act_b ? q_b : q_a;
                    } else {
                        mine[TH] = q_b;
                    }
                    s_prev = s_b; q_last = q_b;
                    *reinterpret_cast<double2 *>(X + cur + tid * 16) = make_double2(q_a, q_b);
                    if (wave_sends) {      // wave-uniform
                        char *const base = reinterpret_cast<char *>(lds);
                        const int32_t dummy = kDummyB + tid * 8;
                        *reinterpret_cast<double *>(base + (ring_b >= 0 && act_a ? ring_b + slot_b : dummy)) = q_a;
                        *reinterpret_cast<double *>(base + (ring_b >= 0 && act_b ? ring_b + ((slot_b + 8) & (kRingB - 8)) : dummy)) = q_b;
                        slot_b = (slot_b + 16) & (kRingB - 8);
                    }
                    own_b += 2 * kRowB;
                    if (own_b >= wrap + tid * 8) own_b -= wrap;
                    barrier_lds();
                }
            };
            for (int32_t k0 = 0; k0 < n_steps; k0 += PF) {
                if (k0 >= span && 2 * (k0 + PF) <= nrows) steps(std::false_type(), k0);      // every lane busy on both rows of every step of the chunk
                else steps(std::true_type(), k0);
            }
            if (!idle) a.q[col] = q_last;
        } else if (role == 1) {
            // ---------------------------------------------------------------- waves 4-7: rows in.  Wave 4 + h + 2 p: columns 128 h + 2 ln, 128 h + 2 ln + 1 of
            // the rows r with r % 2 == p: one row parked and one requested per step and wave
            const int32_t ca = ((wave - 4) & 1) * (TH / 2) + 2 * ln, par = (wave - 4) >> 1;
            auto c4_of = [&](int32_t c) { return c < tm.nc ? a.coef[4 * (int64_t)(tm.c0 + c) + 3] : 0.0; };
            const double c4a0 = c4_of(ca), c4a1 = c4_of(ca + 1);
            // a hole's scaled lateral inflow also goes into its sender ring, at slot (row + lag) % 32
            auto hole_of = [&](int32_t c, int32_t &ring_b, int32_t &slot0) {
                const int4 lm = a.lane[tm.c0 + (c < tm.nc ? c : 0)];
                const bool hole = c < tm.nc && (lm.x & kDirectHoleBit) != 0 && ((lm.x >> rr::kDirectSenderShift) & kSenderMask) != 0;
                ring_b = hole ? kStageB + (((lm.x >> rr::kDirectSenderShift) & kSenderMask) - 1) * kRingB : -1;
                slot0 = (r0 + lm.w) & 31;      // of local row 0
            };
            int32_t ring0, ring1, hs0, hs1;
            hole_of(ca, ring0, hs0); hole_of(ca + 1, ring1, hs1);
            const int32_t dummy_b = kDummyB + (TH + (wave - 4) * 64 + ln) * 8;      // (the rows-in waves' dummy slots follow the routing lanes')
            const bool wave_holes = __builtin_amdgcn_ballot_w64(ring0 >= 0 || ring1 >= 0) != 0;
            // a 16-byte load may reach past the tile's last column (the next tile's, or -- past the row's end -- zeros): never used
            const uint32_t va = ca < tm.nc ? (uint32_t)(tm.c0 + ca) * 8u : kDropAccess;
            // the wave's rows r0 + par, r0 + par + 2, ... of the caller's ring (a ring of one row: every row is that one)
            uint32_t rin = (uint32_t)(r0 + par) % a.in_rows;
            const uint32_t two = a.in_rows > 1 ? 2u : 0u;
            // AH of the wave's rows in flight = 2 AH rows ahead: a step cannot be shorter than the memory latency over AH (a loaded
            // latency of ~3.5 us; 16 rows ahead held the one-row tick at 0.22 us); the register ring is indexed statically
            constexpr int AH = PF;
            double2 Pa[AH];
            auto request = [&](int32_t arrival, double2 &pa) {      // row r0 + arrival, or nothing past the task's rows
                const __amdgpu_buffer_rsrc_t src = make_rsrc(reinterpret_cast<const char *>(a.in) + (uint64_t)rin * row_bytes, arrival < nrows ? row_bytes : 0u);      // (no records: the load is dropped)
                pa = load_f64x2_(src, va);
                rin += two;
                if (rin >= a.in_rows) rin -= a.in_rows;
                rin = __builtin_amdgcn_readfirstlane(rin);
            };
#pragma unroll
            for (int j = 0; j < AH; ++j) request(2 * j + par, Pa[j]);
            int32_t in_b = par * kRowB;      // the window slot of the next row this wave parks
            auto park = [&](const double2 &pa, int32_t arrival) {      // into the window, scaled (the ring of k_tile holds c4dt * lateral too)
                const double x0 = pa.x * c4a0, x1 = pa.y * c4a1;
                *reinterpret_cast<double2 *>(F + in_b + ca * 8) = make_double2(x0, x1);
                in_b += 2 * kRowB;
                if (in_b >= wrap) in_b -= wrap;
                if (wave_holes) {      // wave-uniform
                    const bool real = arrival < nrows;
                    *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring0 >= 0 ? ring0 + ((hs0 + arrival) & 31) * 8 : dummy_b)) = x0;
                    *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring1 >= 0 ? ring1 + ((hs1 + arrival) & 31) * 8 : dummy_b)) = x1;
                }
            };
            park(Pa[0], par);      // rows 0 and 1, before the first step
            request(2 * AH + par, Pa[0]);
            __syncthreads();
            for (int32_t k0 = 0; k0 < n_steps; k0 += PF) {      // a chunk is one revolution of the register ring
#pragma unroll
                for (int s = 0; s < PF; ++s) {      // step k0 + s: the wave's row 2 (k0 + s) + 2 + par arrives, the one 32 rows further on is requested
                    park(Pa[(s + 1) % AH], 2 * (k0 + s) + 2 + par);
                    request(2 * (k0 + s) + 2 + par + 2 * AH, Pa[(s + 1) % AH]);
                    barrier_lds();
                }
            }
        } else if (role == 2) {
            // ---------------------------------------------------------------- waves 8, 9: rows out.  Wave 8 + p stores the rows r with r % 2 == p; lane -> columns
            // 2 ln, 2 ln + 1 and 128 + 2 ln, 128 + 2 ln + 1.  The descriptor ends behind the tile's last column: a 16-byte piece past
            // it is dropped by the range check, and so is the second half of the piece that holds the last column of a tile with an
            // odd number of them (the check is made per dword).
            const int32_t par = wave - 8;
            const uint32_t va = (uint32_t)(tm.c0 + 2 * ln) * 8u, vb = va + 128u * 8u;
            const uint32_t tile_end = __builtin_amdgcn_readfirstlane((uint32_t)(tm.c0 + tm.nc) * 8u);      // (scalar: the descriptor must not be built per lane)
            uint32_t rout = (uint32_t)(r0 + par) % a.out_rows;
            const uint32_t two = a.out_rows > 1 ? 2u : 0u;
            int32_t out_b = par * kRowB;
            __syncthreads();
            for (int32_t k0 = 0; k0 < n_steps; k0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {      // step j: the pair routed last at step j - 1 leaves: rows 2 (j - 1) - 2 span (+ 1)
                    const int32_t leaving = 2 * (k0 + s - 1 - span) + par;
                    if (leaving >= 0 && leaving < nrows) {      // wave-uniform
                        const double2 xa = *reinterpret_cast<const double2 *>(F + out_b + 2 * ln * 8), xb = *reinterpret_cast<const double2 *>(F + out_b + 2 * ln * 8 + 128 * 8);
                        const __amdgpu_buffer_rsrc_t dst = make_rsrc(reinterpret_cast<char *>(a.out) + (uint64_t)rout * row_bytes, tile_end);
                        store_f64x2(dst, va, make_double2(clip0(xa.x), clip0(xa.y)));
                        store_f64x2(dst, vb, make_double2(clip0(xb.x), clip0(xb.y)));
                        out_b += 2 * kRowB;
                        if (out_b >= wrap) out_b -= wrap;
                        rout += two;
                        if (rout >= a.out_rows) rout -= a.out_rows;
                        rout = __builtin_amdgcn_readfirstlane(rout);
                    }
                    barrier_lds();
                }
            }
        } else {
            // ---------------------------------------------------------------- waves 10, 11: what the skeleton needs
            // The senders' rings fill by themselves (above); this wave writes every record that is complete -- 16 ticks, slot = tick % 16
            // with tick = row + lag: k_tile's layout -- into the skeleton's record ring, eight lanes per 128-byte record.  The lanes
            // of a wave form eight groups of eight, a group serves four senders, one of them a step, so a sender has a turn every four
            // steps (eight rows) and completes a record every eight; a ring holds two records, so the one being written out is never the
            // one being filled.  A task's first and last record are partly the neighbouring tasks': only the slots this task made are written.
            const int32_t s0 = a.send_ptr[t], ns = a.send_ptr[t + 1] - s0;
            const int32_t piece = ln & 7, member = ln >> 3;
            auto write_piece = [&](int64_t off, int32_t lo, int32_t hi, const double2 &v) {      // slots [lo, hi) of the record
                double *dst = a.rec + off + 2 * piece;
                if (lo <= 2 * piece && 2 * piece + 2 <= hi) *reinterpret_cast<double2 *>(dst) = v;
                else {
                    if (lo <= 2 * piece && 2 * piece < hi) dst[0] = v.x;
                    if (lo <= 2 * piece + 1 && 2 * piece + 1 < hi) dst[1] = v.y;
                }
            };
            constexpr int NS = kDirectSenders / 16;   // senders per group of eight lanes and wave
            const int32_t sw = wave - 10;
            int32_t ring_b[NS];                       // its ring in LDS, or -1
            int32_t done[NS], end[NS], avail0[NS];    // ticks (row + lag) written out so far / of the task's last row + 1 / visible at step 0 (may be negative)
            uint32_t chk[NS];                         // ring chunk of the record `done` lies in
            int64_t roff[NS];                         // ... and its offset in the record ring, in doubles
            const int64_t chunk_step = (int64_t)a.np * kRec, ring = (int64_t)a.rec_chunks * a.np * kRec;
#pragma unroll
            for (int f = 0; f < NS; ++f) {
                const int32_t i = f + NS * sw + 2 * NS * member;
                const bool have = i < ns;
                const int32_t sl = have ? a.send_lane[s0 + i] : 0;
                const int4 lm = a.lane[tm.c0 + (sl & 0x3FF)];
                const bool hole = (sl & kDirectHoleBit) != 0;
                ring_b[f] = have ? kStageB + i * kRingB : -1;
                done[f] = r0 + lm.w;
                end[f] = done[f] + nrows;
                // what wave 10 sees at step j: an outlet's rows below 2 j - 2 level (it routed two more at every step since it began),
                // a hole's rows up to 2 j + 1 (rows 2 j + 2, 2 j + 3 are parked during step j)
                avail0[f] = hole ? r0 + lm.w + 2 : r0 + lm.w - 2 * (lm.x & rr::kDirectDelayMask);
                chk[f] = ((uint32_t)done[f] >> 4) % a.rec_chunks;
                roff[f] = ((int64_t)chk[f] * a.np + (have ? lm.z : 0)) * kRec;
            }
            auto turn = [&](int f, int32_t avail, bool last) {      // writes sender f's record if it is complete (last: whatever this task made of it)
                if (ring_b[f] < 0) return;
                const int32_t upto = last ? min(end[f], (done[f] | 15) + 1) : (done[f] | 15) + 1;
                if (done[f] < upto && upto <= min(avail, end[f])) {
                    const double2 v = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(lds) + ring_b[f] + ((done[f] >> 4) & 1) * (kRec * 8) + piece * 16);
                    write_piece(roff[f], done[f] & 15, ((upto - 1) & 15) + 1, v);
                    done[f] = upto;
                    if ((upto & 15) == 0) {
                        roff[f] += chunk_step;
                        if (++chk[f] == a.rec_chunks) { chk[f] = 0; roff[f] -= ring; }
                    }
                }
            };
            __syncthreads();
            for (int32_t k0 = 0; k0 < n_steps; k0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    turn(s % NS, avail0[s % NS] + 2 * (k0 + s), false);
                    barrier_lds();
                }
            }
            // the records completed since their sender's last turn, then the task's last (partial) ones: everything is staged by now
            wave_lds_fence();
#pragma unroll
            for (int f = 0; f < NS; ++f) { turn(f, 0x7FFFFFFF, false); turn(f, 0x7FFFFFFF, false); turn(f, 0x7FFFFFFF, true); }
        }
    }
}

}  // namespace
