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
// What a tick (0.195 us per row and CU) answers to, measured build against build on one box (profiles/r04_direct_tick_ab.txt):
// not the instruction streams within reason (rows-in waves of 25 or 38 instructions, four of them taking turns, a lighter rows-out
// wave or wave 7; 63 instructions do slow it down), not the barriers (bare ones cost 9 ns), not an LDS read less per lane, not the
// rows in flight beyond 32, their alignment, or the memory level they come from.  Roles switched off one at a time: the routing
// waves alone need 1.6 ms per 512 rows (0.139 us a tick: their LDS round trips), each I/O role alone 1.0-1.2, the three I/O roles
// together 1.96, everything 2.06 -- no role's removal saves more than 8 %.  TWO rows per barrier (a variant, parity-green, kept
// as a patch) bring the routing waves to 1.16 ms and the whole to the same 2.1: there the I/O roles together are the time.  A
// kernel that only loads and stores the same row segments (profiles/microbench/row_stream_probe.hip) needs 0.73 + 0.68 = 1.42 ms:
// the CU's vector memory path serves loads and stores one after the other, ~87 cycles per 64 x 16-byte instruction (22-25 GB/s
// per CU, a plain copy's share of the chip).
//
// The skeleton (reaches with large or tall subtrees: 5 %) keeps records and k_tile.  Its columns lie between the subtrees'
// columns (HOLES).  A hole's scaled lateral inflow (waves 4, 5) and the discharge of an outlet lane (a small subtree's last reach)
// go into small LDS rings, from which wave 7 writes whole records into the skeleton's record ring: the hole's own position, the
// ghost that mirrors the outlet there; k_rec_out, given the holes' columns, patches the output rows from the skeleton's records afterwards.
struct DirectTile { int32_t c0, nc, lag_lo, span; };
struct DirectArgs {
    const DirectTile *tiles;
    int32_t n_tiles;
    const int4 *lane;           // per column {delay | kDirectHole, upstream lanes (3 x 10 bits), xinfo, lag}
    const double *coef;         // per column {c1row, c2, c3, c4dt}
    const int32_t *send_ptr;    // per tile: its senders in send_lane
    const int32_t *send_lane;   // lane | kDirectHole
    double *q;                  // per column: carried discharge
    double *qch;                // UNIT: per column: carried channel discharge (q_ch; zero on headwaters)
    const double *in;           // lateral rows (in_rows x n), read cyclically
    double *out;                // discharge rows (out_rows x n), written cyclically
    int64_t n;
    uint32_t in_rows, out_rows;
    double *rec;                // the skeleton's record ring [chunks][np][16] (k_tile's layout)
    uint32_t rec_chunks;
    int32_t np;
    int32_t m, K, total;        // this launch: rows [m K, min((m + 1) K, total))
    double *exports;            // boundary series another GPU reads (a partitioned network's exports that lanes route): (total, n_export)
    int32_t n_export;
    const float *in32;          // IN32: float32 lateral rows instead of `in` (as qlateral files store them; exact in float64)
    float *out32;               // OUT32: float32 discharge rows, each the mean of `factor` routed rows (TransformMuskingum.py:128-142), (total / factor, n), not cyclic
    int32_t factor;
    int32_t nsub;               // SUB: routing sub-steps per row (2 ... kDirectMaxSub); the row's output is their mean
    double inv_nsub;
    uint32_t in32_sel, out32_sel;   // byte selectors of the float32 rows (kSelNative / kSelSwap, rr_plan_set_row_format)
};
constexpr int kDirectMaxSub = 4;      // a hole's lateral value goes into `nsub` ring slots at once: with more the send wave's turn (every eight ticks) could come too late
constexpr int32_t kDirectHoleBit = rr::kDirectHole;
constexpr int kDirectLanes = 256, kDirectAhead = 16;
constexpr int kDirectThreads = kDirectLanes + 4 * 64;      // four routing waves + in, in, out, send: two waves per SIMD
constexpr int kDirectMaxWindow = 64;      // rows of the LDS window, span + 3
constexpr int kDirectSenders = rr::kDirectSenders;      // per tile
// LDS in doubles: X[2][258] | S[senders][32]: what the skeleton needs, a ring of two records (32 ticks) per sender, filled by the
// lanes that make the values -- the outlet's routing lane, the hole's column in waves 4 / 5 -- at slot = tick % 32 (tick = row + lag,
// k_tile's record slot) | D[2][256]: a slot per lane that takes the store of a lane with nothing to send (no branch, no bank
// conflict) | F[window rows][256]: 4.1 + 16.4 + 4.1 + 131 KB = 156 KB of the CU's 160 with the largest window
constexpr int kDirectStage = kDirectSenders * 2 * kRec + 2 * kDirectLanes;      // doubles
constexpr size_t direct_lds_bytes(int window_rows)
{
    return (size_t)(2 * (kDirectLanes + kTilePad) + kDirectStage + (int64_t)window_rows * kDirectLanes) * sizeof(double);
}

// the reference's clip at zero (_numba_kernels.py:84): x > 0 ? x : 0 in one instruction (the compiler's own max quiets its operand first)
__device__ __forceinline__ double clip0(double x)
{
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 load_f32x2_(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x2 bits = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    f32x2 v;
    __builtin_memcpy(&v, &bits, sizeof v);
    return v;
}
__device__ __forceinline__ void store_f32x2_nt(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x2 v)
{
    u32x2 bits;
    __builtin_memcpy(&bits, &v, sizeof bits);
    __builtin_amdgcn_raw_buffer_store_b64(bits, r, (int)byte_off, 0, 2);      // aux: nt
}
__device__ __forceinline__ double2 load_f64x2_(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    double2 v;
    __builtin_memcpy(&v, &bits, sizeof v);
    return v;
}

// Row segments leave with the non-temporal hint: a tile's first and last 128-byte line are shared with its neighbours, which run on
// other XCDs, so those lines reach memory in two parts either way; streamed past the L2 they cost less (a store-only probe of the same
// segments: 790 against 1,027 us; the year at 1M reaches 199.9 against 204.6 ms, profiles/r05_direct_ab_nt.txt).  Non-temporal LOADS of the
// rows were slower (212.9 ms), tiles dealt to the XCDs in column order changed nothing (204.0).
__device__ __forceinline__ void store_f64x2_nt(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, double2 v)
{
    u32x4 bits;
    __builtin_memcpy(&bits, &v, sizeof bits);
    __builtin_amdgcn_raw_buffer_store_b128(bits, r, (int)byte_off, 0, 2);      // aux: nt
}

// IN32: the lateral rows are float32 (the in-waves load 8 bytes per lane and convert; float32 -> float64 is exact, so the results are those
// of the float64 copy bit for bit).  OUT32: the routers' post-processing in the rows-out wave -- the mean over `factor` consecutive rows
// (sequential sum, one division, as numpy reduces a strided axis) and the float32 cast; K is a multiple of factor (choose_schedule), so
// every output row lies inside one task.
// SUB: routing sub-steps (_numba_kernels.py:66-84: the row's lateral value is held over its sub-steps, the output is their mean).  A tick
// is a sub-step: a lane reads F[its row] at every sub-step of the row and puts the mean there at the last one; rows arrive and leave
// every nsub ticks; the senders' rings and the skeleton's records are in sub-step space, as k_tile<SUB>'s.  Channel-only routing
// (Muskingum.py:262-290: no lateral rows) needs no flag: without a lateral array every rows-in load is dropped by its descriptor's range
// check and returns zero.
// UNIT: UnitMuskingum's recurrence (_numba_kernels.py:88-171, in k_tile<UNIT>'s gather form, operation for operation): the rows are the
// convolved lateral inflow, unscaled; a headwater lane publishes its row's value as it is (zero coefficients), an inner lane keeps its
// channel discharge q_ch beside the published q_full = q_ch + lateral and adds its headwater tributaries (the first `nh` of its upstream
// lanes) of the SAME row to both the c1 and the c2 term; the rows-out wave leaves headwater columns unclipped (line 122-123).
template <int PF, bool IN32 = false, bool OUT32 = false, bool SUB = false, int UNIT = 0>
__global__ __launch_bounds__(kDirectThreads, 1) void k_direct(const DirectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int TH = kDirectLanes, THP = TH + kTilePad;
    static_assert(PF % 2 == 0 && PF == kRec, "the parity of a tick is the parity of its place in a chunk; wave 6 writes a record per sender and chunk");
    char *const X = reinterpret_cast<char *>(lds);                  // [2][THP] discharges of the last two ticks, each followed by a slot that holds 0.0
    double *const S = lds + 2 * THP;                                // the senders' rings, then the dummy slots
    char *const F = reinterpret_cast<char *>(lds + 2 * THP + kDirectStage);        // [span + 3][TH] the row window
    const int tid = threadIdx.x, wave = tid >> 6, role = wave < 4 ? 0 : (wave < 6 ? 1 : wave - 4), ln = tid & 63;      // 0 route, 1 in (two waves), 2 out, 3 send
    if (tid < 2) lds[tid * THP + TH] = 0.0;
    const int32_t r0 = a.m * a.K, nrows = min(a.K, a.total - r0);
    const uint32_t row_bytes = (uint32_t)a.n * 8u;                  // n < 2^29 (choose_schedule)
    constexpr int kRowB = TH * 8;                                    // bytes of a window row
    constexpr int kStageB = 2 * THP * 8, kDummyB = kStageB + kDirectSenders * 2 * kRec * 8;      // byte offsets in LDS of S and of the dummy slots
    constexpr int kSenderMask = 0x7F;

    for (int32_t t = (int32_t)blockIdx.x; t < a.n_tiles; t += (int32_t)gridDim.x) {
        const DirectTile tm = a.tiles[t];
        const int32_t span = tm.span, wrap = (span + 3) * kRowB;     // bytes of the window in use
        // local tick k: row k + 1 arrives (waves 4, 5), lane with delay d routes row k - d (waves 0-3), row k - 1 - span leaves (wave 6),
        // the values of tick k - 1 are forwarded (wave 7); n_ticks of them, in chunks of PF
        const int32_t nsub = SUB ? a.nsub : 1, total_sub = nrows * nsub;      // sub-steps of the task
        const int32_t n_ticks = total_sub + span + 2;
        const int32_t tick0 = r0 * nsub + tm.lag_lo;      // the tile's tick (sub-step + lag) at local tick 0
        __syncthreads();      // every wave has left the previous tile

        if (role == 0) {
            // ---------------------------------------------------------------- routing lanes: lane = column
            const bool live = tid < tm.nc;
            const int32_t col = tm.c0 + (live ? tid : 0);
            const int4 lm = a.lane[col];
            const bool idle = !live || (lm.x & kDirectHoleBit) != 0;      // a hole's column only passes through the window
            const int32_t delta = idle ? 0x40000000 : (lm.x & rr::kDirectDelayMask);
            // an outlet that feeds the skeleton also puts its discharge into its sender ring; every lane of a tile is at the same
            // tick (row + lag) at the same moment, so the slot is wave-uniform
            const int32_t sender = idle ? 0 : (lm.x >> rr::kDirectSenderShift) & kSenderMask;
            const int32_t stage_b = sender ? kStageB + (sender - 1) * (2 * kRec * 8) : kDummyB + tid * 8;      // dummy: the lane's own slot (32 of them: slot 0 ... see below)
            const bool wave_sends = __builtin_amdgcn_ballot_w64(sender != 0) != 0;
            // a boundary export of a partitioned network that a lane routes: its discharge after every step, unclamped, into its column of the export series
            const int32_t exp_slot = (!idle && (lm.x & rr::kDirectExport)) ? lm.z : -1;
            const bool wave_exports = __builtin_amdgcn_ballot_w64(exp_slot >= 0) != 0;
            const int32_t u0 = lm.y & 0x3FF, u1 = (lm.y >> 10) & 0x3FF, u2 = (lm.y >> 20) & 0x3FF;
            const bool no_coef = idle || (UNIT && u0 == 0x3FF);      // UNIT: a headwater computes nothing (zero coefficients republish its row's value)
            const double c1 = no_coef ? 0.0 : a.coef[4 * (int64_t)col], c2 = no_coef ? 0.0 : a.coef[4 * (int64_t)col + 1], c3 = no_coef ? 0.0 : a.coef[4 * (int64_t)col + 2];
            const double q0 = idle ? 0.0 : a.q[col];
            // UNIT: the first nh upstream lanes are headwaters; up*_b are then the inner ones
            const int32_t nh = UNIT ? (int32_t)((uint32_t)lm.y >> 30) : 0;
            const int32_t i0 = nh == 0 ? u0 : (nh == 1 ? u1 : (nh == 2 ? u2 : 0x3FF)), i1 = nh == 0 ? u1 : (nh == 1 ? u2 : 0x3FF), i2 = nh == 0 ? u2 : 0x3FF;
            const int32_t up0_b = (i0 == 0x3FF || idle ? TH : i0) * 8, up1_b = (i1 == 0x3FF || idle ? TH : i1) * 8, up2_b = (i2 == 0x3FF || idle ? TH : i2) * 8;
            const int32_t hw0_b = (nh >= 1 && !idle ? u0 : TH) * 8, hw1_b = (nh >= 2 && !idle ? u1 : TH) * 8, hw2_b = (nh >= 3 && !idle ? u2 : TH) * 8;
            double qc = (UNIT && !idle) ? a.qch[col] : 0.0;      // UNIT: the channel discharge
            *reinterpret_cast<double *>(X + tid * 8) = q0;
            *reinterpret_cast<double *>(X + THP * 8 + tid * 8) = q0;
            int32_t own_b = idle || delta == 0 ? 0 : wrap - delta * kRowB;      // window slot of the row this lane routes this tick: (k - delta) mod (span + 3)
            if (!idle && delta > 0 && own_b < 0) own_b += wrap;                   // (delta <= span < span + 3)
            double s_prev = 0.0, q_last = q0;      // the lane's own discharge one tick back stays in a register
            __syncthreads();      // the discharges carried in, and row 0 in the window (waves 4, 5)
            auto ticks = [&](auto tested, int32_t k0) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const int prev = ((s + 1) & 1) * (THP * 8), cur = (s & 1) * (THP * 8);
                    // _numba_kernels.py:63-84 in gather form, the arithmetic of k_tile's short tick operation for operation
                    const double q_old = q_last;
                    const double s_cur = (*reinterpret_cast<const double *>(X + prev + up0_b) + *reinterpret_cast<const double *>(X + prev + up1_b)) +
                                         *reinterpret_cast<const double *>(X + prev + up2_b);
                    double *mine = reinterpret_cast<double *>(F + own_b + tid * 8);
                    const double lat = *mine;
                    double qk;
                    if constexpr (UNIT != 0) {      // _numba_kernels.py:142-167 in gather form, k_tile<UNIT>'s short tick operation for operation
                        const double s_hw = (*reinterpret_cast<const double *>(X + prev + hw0_b) + *reinterpret_cast<const double *>(X + prev + hw1_b)) +
                                            *reinterpret_cast<const double *>(X + prev + hw2_b);
                        const double r = __builtin_fma(c1, s_hw + s_cur, __builtin_fma(c2, s_hw + s_prev, c3 * qc));
                        qk = r + lat;
                        if (decltype(tested)::value) {
                            const bool active = (uint32_t)(k0 + s - delta) < (uint32_t)nrows;
                            qc = active ? r : qc;
                            qk = active ? qk : q_old;
                            *mine = active ? qk : lat;
                        } else {
                            qc = r;
                            *mine = qk;
                        }
                    } else {
                    qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, q_old, lat)));
                    if (decltype(tested)::value) {      // an idle lane keeps its discharge and hands its window slot back as it found it (nobody else writes it this tick)
                        const bool active = (uint32_t)(k0 + s - delta) < (uint32_t)nrows;
                        qk = active ? qk : q_old;
                        *mine = active ? qk : lat;
                    } else {
                        *mine = qk;
                    }
                    }
                    s_prev = s_cur; q_last = qk;
                    *reinterpret_cast<double *>(X + cur + tid * 8) = qk;
                    if (wave_sends) {      // wave-uniform
                        const int32_t slot_b = ((r0 + tm.lag_lo + k0 + s) & 31) * 8;
                        bool put = sender != 0;
                        if (decltype(tested)::value) put = put && (uint32_t)(k0 + s - delta) < (uint32_t)nrows;
                        *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (put ? stage_b + slot_b : kDummyB + tid * 8)) = qk;
                    }
                    if (wave_exports) {      // wave-uniform; a few lanes of a few tiles
                        const int32_t row = k0 + s - delta;
                        if (exp_slot >= 0 && (uint32_t)row < (uint32_t)nrows) a.exports[(int64_t)(r0 + row) * a.n_export + exp_slot] = qk;
                    }
                    own_b = own_b + kRowB == wrap ? 0 : own_b + kRowB;
                    barrier_lds();
                }
            };
            if constexpr (SUB) {
                // sub-steps: the lane stays on its row's window slot for nsub ticks, keeps the running sum and writes the mean at the last one
                int32_t own = 0, phase = 0;      // byte offset of the row's slot; sub-step of the row
                double isum = 0.0;
                for (int32_t k0 = 0; k0 < n_ticks; k0 += PF) {
#pragma unroll
                    for (int s = 0; s < PF; ++s) {
                        const int prev = ((s + 1) & 1) * (THP * 8), cur = (s & 1) * (THP * 8);
                        const double q_old = q_last;
                        const double s_cur = (*reinterpret_cast<const double *>(X + prev + up0_b) + *reinterpret_cast<const double *>(X + prev + up1_b)) +
                                             *reinterpret_cast<const double *>(X + prev + up2_b);
                        double *mine = reinterpret_cast<double *>(F + own + tid * 8);
                        const double lat = *mine;
                        const bool active = (uint32_t)(k0 + s - delta) < (uint32_t)total_sub;
                        double qk;
                        if constexpr (UNIT != 0) {      // (the row's lateral value is held over its sub-steps, a headwater republishes it every one of them)
                            const double s_hw = (*reinterpret_cast<const double *>(X + prev + hw0_b) + *reinterpret_cast<const double *>(X + prev + hw1_b)) +
                                                *reinterpret_cast<const double *>(X + prev + hw2_b);
                            const double r = __builtin_fma(c1, s_hw + s_cur, __builtin_fma(c2, s_hw + s_prev, c3 * qc));
                            qk = active ? r + lat : q_old;
                            qc = active ? r : qc;
                        } else {
                            qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, q_old, lat)));
                            qk = active ? qk : q_old;
                        }
                        const double acc = (phase == 0 ? 0.0 : isum) + qk;      // (k_tile<SUB>'s order: 0 + first, + second, ...)
                        const bool last = active && phase + 1 == nsub;
                        if (last && !(UNIT != 0 && no_coef)) *mine = acc * a.inv_nsub;      // the row's mean, unclamped: the rows-out wave clips (UNIT: a headwater's row value stays as it arrived, _numba_kernels.py:122-123)
                        isum = active ? acc : isum;
                        own = last ? (own + kRowB == wrap ? 0 : own + kRowB) : own;
                        phase = active ? (last ? 0 : phase + 1) : phase;
                        s_prev = s_cur; q_last = qk;
                        *reinterpret_cast<double *>(X + cur + tid * 8) = qk;
                        if (wave_sends) {      // wave-uniform
                            const int32_t slot_b = ((tick0 + k0 + s) & 31) * 8;
                            *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (sender != 0 && active ? stage_b + slot_b : kDummyB + tid * 8)) = qk;
                        }
                        if (wave_exports) {
                            if (exp_slot >= 0 && active) a.exports[(int64_t)(r0 * nsub + k0 + s - delta) * a.n_export + exp_slot] = qk;
                        }
                        barrier_lds();
                    }
                }
            } else {
                for (int32_t k0 = 0; k0 < n_ticks; k0 += PF) {
                    if (k0 >= span && k0 + PF <= nrows) ticks(std::false_type(), k0);      // every lane busy on every tick of the chunk
                    else ticks(std::true_type(), k0);
                }
            }
            if (!idle) a.q[col] = *reinterpret_cast<const double *>(X + THP * 8 + tid * 8);      // PF is even: the last tick wrote buffer 1
            if (UNIT && !idle) a.qch[col] = qc;
        } else if (role == 1) {
            // ---------------------------------------------------------------- waves 4, 5: rows in.  Wave 4 + h, lane -> columns 128 h + 2 ln, 128 h + 2 ln + 1
            const int32_t ca = (wave - 4) * (TH / 2) + 2 * ln;
            auto c4_of = [&](int32_t c) { return UNIT ? 1.0 : (c < tm.nc ? a.coef[4 * (int64_t)(tm.c0 + c) + 3] : 0.0); };      // (UNIT: the rows are discharges already)
            const double c4a0 = c4_of(ca), c4a1 = c4_of(ca + 1);
            // a hole's scaled lateral inflow also goes into its sender ring, at slot (row + lag) % 32
            auto hole_of = [&](int32_t c, int32_t &ring_b, int32_t &slot0) {
                const int4 lm = a.lane[tm.c0 + (c < tm.nc ? c : 0)];
                const bool hole = c < tm.nc && (lm.x & kDirectHoleBit) != 0 && ((lm.x >> rr::kDirectSenderShift) & kSenderMask) != 0;
                ring_b = hole ? kStageB + (((lm.x >> rr::kDirectSenderShift) & kSenderMask) - 1) * (2 * kRec * 8) : -1;
                slot0 = (r0 * nsub + lm.w) & 31;      // of local row 0 (its first sub-step)
            };
            int32_t ring0, ring1, hs0, hs1;
            hole_of(ca, ring0, hs0); hole_of(ca + 1, ring1, hs1);
            const int32_t dummy_b = kDummyB + (TH + (wave - 4) * 64 + ln) * 8;      // (the in-waves' dummy slots follow the routing lanes')
            const bool wave_holes = __builtin_amdgcn_ballot_w64(ring0 >= 0 || ring1 >= 0) != 0;
            // a 16-byte load may reach past the tile's last column (the next tile's, or -- past the row's end -- zeros): never used
            const bool has_lat = IN32 ? a.in32 != nullptr : a.in != nullptr;      // channel-only routing: every load is dropped and returns zero
            const uint32_t va = ca < tm.nc && has_lat ? (uint32_t)(tm.c0 + ca) * (IN32 ? 4u : 8u) : kDropAccess;
            uint32_t rin = (uint32_t)r0 % a.in_rows;
            const char *row = IN32 ? reinterpret_cast<const char *>(a.in32 + (int64_t)rin * a.n) : reinterpret_cast<const char *>(a.in + (int64_t)rin * a.n);
            const uint32_t in_row_bytes = IN32 ? row_bytes / 2u : row_bytes;
            // AH rows in flight: a tick cannot be shorter than the memory latency over AH (16 rows ahead hold the tick at 0.22 us:
            // 32 KB per CU in flight against ~3.5 us under load); the register ring is indexed statically: two chunk bodies alternate
            constexpr int AH = 2 * PF;
            typedef typename std::conditional<IN32, f32x2, double2>::type Pt;
            constexpr int AR = SUB ? 8 : AH;      // rows in flight (with sub-steps a row lasts nsub >= 2 ticks)
            Pt Pa[AR];
            auto request = [&](int32_t arrival, Pt &pa) {      // row r0 + arrival, or nothing past the task's rows
                const __amdgpu_buffer_rsrc_t src = make_rsrc(row, in_row_bytes);
                if constexpr (IN32) pa = load_f32x2_(src, arrival < nrows ? va : kDropAccess);
                else pa = load_f64x2_(src, arrival < nrows ? va : kDropAccess);
                ++rin; row += in_row_bytes;
                if (rin == a.in_rows) { rin = 0; row = IN32 ? reinterpret_cast<const char *>(a.in32) : reinterpret_cast<const char *>(a.in); }
            };
#pragma unroll
            for (int j = 0; j < AR; ++j) request(j, Pa[j]);
            int32_t in_b = 0;
            auto park = [&](const Pt &pa, int32_t arrival) {      // into the window, scaled (the ring of k_tile holds c4dt * lateral too)
                double x0, x1;
                if constexpr (IN32) { x0 = (double)f32_from_file(pa.x, a.in32_sel) * c4a0; x1 = (double)f32_from_file(pa.y, a.in32_sel) * c4a1; }
                else { x0 = pa.x * c4a0; x1 = pa.y * c4a1; }
                *reinterpret_cast<double2 *>(F + in_b + ca * 8) = make_double2(x0, x1);
                in_b = in_b + kRowB == wrap ? 0 : in_b + kRowB;
                if (wave_holes) {      // wave-uniform
                    const bool real = arrival < nrows;
                    if constexpr (SUB) {      // the row's value in each of its sub-steps' slots (k_rec_in<SUB> does the same for the record path)
                        for (int32_t p = 0; p < nsub; ++p) {
                            *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring0 >= 0 ? ring0 + ((hs0 + arrival * nsub + p) & 31) * 8 : dummy_b)) = x0;
                            *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring1 >= 0 ? ring1 + ((hs1 + arrival * nsub + p) & 31) * 8 : dummy_b)) = x1;
                        }
                    } else {
                        *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring0 >= 0 ? ring0 + ((hs0 + arrival) & 31) * 8 : dummy_b)) = x0;
                        *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + (real && ring1 >= 0 ? ring1 + ((hs1 + arrival) & 31) * 8 : dummy_b)) = x1;
                    }
                }
            };
            if constexpr (SUB) {
                // Row j is parked during tick j nsub - 1 (one tick before the first lane starts it).  Eight rows in flight cover at least
                // sixteen ticks; the ring is indexed statically: eight bodies a revolution.
                constexpr int AS = AR;
                const int32_t total_b = (n_ticks + PF - 1) / PF * PF;      // barriers every wave of the workgroup goes through
                park(Pa[0], 0);
                request(AS, Pa[0]);
                __syncthreads();
                int32_t k = 0, j = 1;
                while (k < total_b) {
#pragma unroll
                    for (int jj = 0; jj < AS; ++jj) {      // row j + jj sits in Pa[(1 + jj) % AS]
                        for (int32_t w = 0; w < nsub - 1 && k < total_b; ++w, ++k) barrier_lds();
                        if (k >= total_b) break;
                        park(Pa[(1 + jj) % AS], j + jj);
                        request(j + jj + AS, Pa[(1 + jj) % AS]);
                        barrier_lds();
                        ++k;
                    }
                    j += AS;
                }
                continue;      // next tile
            }
            park(Pa[0], 0);      // row 0, before the first tick
            request(AH, Pa[0]);
            __syncthreads();
            auto chunk = [&](auto half, int32_t k0) {      // ticks k0 ... k0 + PF - 1; k0 = PF (2 c + half): row k0 + s + 1 sits in P[(k0 + s + 1) % AH]
#pragma unroll
                for (int s = 0; s < PF; ++s) {      // tick k0 + s: row k0 + s + 1 arrives, row k0 + s + 1 + AH is requested
                    constexpr int base = decltype(half)::value * PF;
                    park(Pa[(base + s + 1) % AH], k0 + s + 1);
                    request(k0 + s + 1 + AH, Pa[(base + s + 1) % AH]);
                    barrier_lds();
                }
            };
            // (an odd last chunk stands outside the loop: a branch around the second half inside it would leave the compiler's count of
            // the loads in flight -- s_waitcnt vmcnt -- at the smaller of the two paths', 15 instead of 31, in every trip)
            int32_t k0 = 0;
            for (; k0 + PF < n_ticks; k0 += 2 * PF) {
                chunk(std::integral_constant<int, 0>(), k0);
                chunk(std::integral_constant<int, 1>(), k0 + PF);
            }
            if (k0 < n_ticks) chunk(std::integral_constant<int, 0>(), k0);
        } else if (role == 2) {
            // ---------------------------------------------------------------- wave 6: rows out.  Lane -> columns 2 ln, 2 ln + 1 and 128 + 2 ln, 128 + 2 ln + 1
            // The descriptor ends behind the tile's last column: a 16-byte piece past it is dropped by the range check, and so is the
            // second half of the piece that holds the last column of a tile with an odd number of them (the check is made per dword).
            constexpr uint32_t kOutB = OUT32 ? 4u : 8u;
            const uint32_t va = (uint32_t)(tm.c0 + 2 * ln) * kOutB, vb = va + 128u * kOutB, tile_end = (uint32_t)(tm.c0 + tm.nc) * kOutB;
            uint32_t rout = OUT32 ? 0u : (uint32_t)r0 % a.out_rows;
            double *row = OUT32 ? nullptr : a.out + (int64_t)rout * a.n;
            float *row32 = OUT32 ? a.out32 + (int64_t)(r0 / a.factor) * a.n : nullptr;      // r0 = m K is a multiple of factor
            double2 sa = make_double2(0.0, 0.0), sb = sa;      // OUT32: the sums of the output row being formed
            int32_t cnt = 0;
            // the mean's division: by a power of two it is the multiplication by the reciprocal, bit for bit, and a float64 division is ~40
            // instructions where the rows-out wave has a tick's worth of time (wave-uniform choice)
            const double fdiv = (double)a.factor, finv = 1.0 / fdiv;
            const bool pow2 = (a.factor & (a.factor - 1)) == 0;
            auto mean_of = [&](double sum) { return pow2 ? sum * finv : sum / fdiv; };
            int32_t out_b = 0;
            // UNIT: a headwater's column leaves as it is (_numba_kernels.py:122-123: no clip)
            auto raw_col = [&](int32_t c) { return UNIT != 0 && c < tm.nc && (a.lane[tm.c0 + c].y & 0x3FF) == 0x3FF; };
            const bool raw0 = raw_col(2 * ln), raw1 = raw_col(2 * ln + 1), raw2 = raw_col(128 + 2 * ln), raw3 = raw_col(128 + 2 * ln + 1);
            auto fin = [](double x, bool raw) { return (UNIT != 0 && raw) ? x : clip0(x); };
            int32_t u_sub = -1 - span, u_ph = 0, u_row = 0;      // SUB: the sub-step the slowest lane did one tick ago, its place in its row, the row
            __syncthreads();
            for (int32_t k0 = 0; k0 < n_ticks; k0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {      // tick k: row k - 1 - span leaves (sub-steps: the row whose last sub-step the slowest lane did at tick k - 1)
                    int32_t leaving = k0 + s - 1 - span;
                    if constexpr (SUB) {      // (no integer division: counted along)
                        leaving = -1;
                        if (u_sub >= 0) {
                            if (u_ph == nsub - 1) leaving = u_row;
                            if (++u_ph == nsub) { u_ph = 0; ++u_row; }
                        }
                        ++u_sub;
                    }
                    if (leaving >= 0 && leaving < nrows) {      // wave-uniform
                        const double2 xa = *reinterpret_cast<const double2 *>(F + out_b + 2 * ln * 8), xb = *reinterpret_cast<const double2 *>(F + out_b + 2 * ln * 8 + 128 * 8);
                        if constexpr (OUT32) {
                            const double2 ca2 = make_double2(fin(xa.x, raw0), fin(xa.y, raw1)), cb2 = make_double2(fin(xb.x, raw2), fin(xb.y, raw3));
                            sa = cnt ? make_double2(sa.x + ca2.x, sa.y + ca2.y) : ca2;
                            sb = cnt ? make_double2(sb.x + cb2.x, sb.y + cb2.y) : cb2;
                            if (++cnt == a.factor) {      // wave-uniform
                                cnt = 0;
                                f32x2 fa, fb;
                                fa.x = f32_to_file((float)mean_of(sa.x), a.out32_sel); fa.y = f32_to_file((float)mean_of(sa.y), a.out32_sel);
                                fb.x = f32_to_file((float)mean_of(sb.x), a.out32_sel); fb.y = f32_to_file((float)mean_of(sb.y), a.out32_sel);
                                const __amdgpu_buffer_rsrc_t dst = make_rsrc(row32, tile_end);
                                store_f32x2_nt(dst, va, fa);
                                store_f32x2_nt(dst, vb, fb);
                                row32 += a.n;
                            }
                        } else {
                            const __amdgpu_buffer_rsrc_t dst = make_rsrc(row, tile_end);
                            store_f64x2_nt(dst, va, make_double2(fin(xa.x, raw0), fin(xa.y, raw1)));
                            store_f64x2_nt(dst, vb, make_double2(fin(xb.x, raw2), fin(xb.y, raw3)));
                            ++rout; row += a.n;
                            if (rout == a.out_rows) { rout = 0; row = a.out; }
                        }
                        out_b = out_b + kRowB == wrap ? 0 : out_b + kRowB;
                    }
                    barrier_lds();
                }
            }
        } else {
            // ---------------------------------------------------------------- wave 7: what the skeleton needs
            // The senders' rings fill by themselves (above); this wave writes every record that is complete -- 16 ticks, slot = tick % 16
            // with tick = row + lag: k_tile's layout -- into the skeleton's record ring, eight lanes per 128-byte record.  The lanes
            // form eight groups of eight, a group serves kDirectSenders / 8 senders in turn, one a tick (eight records a tick at most:
            // one store instruction), so a sender has a turn every eight ticks and completes a record every 16; a ring holds two
            // records, so the one being written out is never the one being filled.  (Copying
            // the values itself, this wave was busy 90 % of a tick and every other wave waited for it: profiles/r04_direct_wave_stamps_before.txt.)
            // A task's first and last record are partly the neighbouring tasks': only the slots this task made are written.
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
            // per turn (= local tick % NS): the sender this lane serves then
            constexpr int NS = kDirectSenders / 8;    // senders per group of eight lanes
            static_assert(PF % NS == 0, "a sender's turns are the ticks with tick % NS == its place in the group");
            int32_t ring_b[NS];                       // its ring in LDS, or -1
            bool holef[NS];                           // the sender is a hole (its values arrive row by row, nsub ticks at a time)
            uint32_t done[NS], end[NS], avail0[NS];   // ticks (sub-step + lag) written out so far / of the task's last sub-step + 1 / visible at local tick 0
            uint32_t chk[NS];                         // ring chunk of the record `done` lies in
            int64_t roff[NS];                         // ... and its offset in the record ring, in doubles
            const int64_t chunk_step = (int64_t)a.np * kRec, ring = (int64_t)a.rec_chunks * a.np * kRec;
#pragma unroll
            for (int f = 0; f < NS; ++f) {
                const int32_t i = member + 8 * f;      // the tile's senders fill the turns one after the other, eight to a turn: a turn is one
                                                       // store instruction whether it carries one record or eight, and the CU issues one vector
                                                       // memory instruction per ~87 cycles (profiles/r04_direct_tick_ab.txt, block 9)
                const bool have = i < ns;
                const int32_t sl = have ? a.send_lane[s0 + i] : 0;
                const int4 lm = a.lane[tm.c0 + (sl & 0x3FF)];
                const bool hole = (sl & kDirectHoleBit) != 0;
                ring_b[f] = have ? kStageB + i * (2 * kRec * 8) : -1;
                holef[f] = hole;
                done[f] = (uint32_t)(r0 * nsub + lm.w);
                end[f] = done[f] + (uint32_t)total_sub;
                // what wave 7 sees at local tick k: an outlet's values of the ticks before k (every lane of the tile is at tick tick0 + k),
                // a hole's rows up to k / nsub (row j is parked during tick j nsub - 1), every sub-step of them
                avail0[f] = hole ? (uint32_t)((r0 + 1) * nsub + lm.w) : (uint32_t)tick0;
                chk[f] = (done[f] >> 4) % a.rec_chunks;
                roff[f] = ((int64_t)chk[f] * a.np + (have ? lm.z : 0)) * kRec;
            }
            auto turn = [&](int f, uint32_t avail, bool last) {      // writes sender f's record if it is complete (last: whatever this task made of it)
                if (ring_b[f] < 0) return;
                const uint32_t upto = last ? min(end[f], (done[f] | 15u) + 1u) : (done[f] | 15u) + 1u;
                if (done[f] < upto && upto <= min(avail, end[f])) {
                    const double2 v = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(lds) + ring_b[f] + ((done[f] >> 4) & 1u) * (kRec * 8) + piece * 16);
                    write_piece(roff[f], (int32_t)(done[f] & 15u), (int32_t)((upto - 1u) & 15u) + 1, v);
                    done[f] = upto;
                    if ((upto & 15u) == 0) {
                        roff[f] += chunk_step;
                        if (++chk[f] == a.rec_chunks) { chk[f] = 0; roff[f] -= ring; }
                    }
                }
            };
            int32_t k_rows = 0, k_ph = 0;      // SUB: (local tick / nsub) * nsub, counted along
            __syncthreads();
            for (int32_t k0 = 0; k0 < n_ticks; k0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const uint32_t seen = SUB ? (uint32_t)(holef[s % NS] ? k_rows : k0 + s) : (uint32_t)(k0 + s);
                    turn(s % NS, avail0[s % NS] + seen, false);
                    if constexpr (SUB) { if (++k_ph == nsub) { k_ph = 0; k_rows += nsub; } }
                    barrier_lds();
                }
            }
            // the records completed since their sender's last turn, then the task's last (partial) ones: everything is staged by now
            wave_lds_fence();
#pragma unroll
            for (int f = 0; f < NS; ++f) { turn(f, 0xFFFFFFFFu, false); turn(f, 0xFFFFFFFFu, false); turn(f, 0xFFFFFFFFu, true); }
        }
    }
}

}  // namespace
