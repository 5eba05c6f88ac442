// rr_kernels_tile.hpp -- the time-tiled routing kernel over subtree tiles (k_tile) and its state kernels.
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

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

// Per-tile and per-position constants travel packed, one pointer each: a kernel argument costs two scalar registers for the
// whole kernel, and with the record buffers in 64 of a wave's 128 vector registers every scalar spilled is a vector one lost.
struct TileMeta {
    int32_t b0, b1;          // positions [b0, b1)
    int32_t level;           // launch d runs macro-chunk d - level
    int32_t lag_lo, lag_hi;  // smallest / largest lag of the tile's positions
    int32_t flags;           // kTileWide | kTileExports
    int32_t pad0, pad1;
};
constexpr int32_t kTileWide = 1, kTileExports = 2, kTaskTested = 4;      // TileMeta::flags: a position with more than three upstream positions / a boundary export in the tile
struct TileArgs {
    const TileMeta *tiles;
    const int4 *pos;                      // per position {lag | flags, first upstream position, xpos, upstream count | headwaters among them << 16};
                                          // xpos: kTileExportBit: position of the ghost that mirrors this reach; kExportBit: its slot in the export series (never both)
    const double *coef;                   // per position {c1row, c2, c3}: c1row is the (uniform) weight of a reach's upstream terms
    double *sq, *ss, *si, *sqch;          // carried state: discharge, sum of upstream discharges one tick back, interval sum, channel discharge
    double *exports;                      // boundary series another GPU reads (multi-GPU)
    int32_t n_export;
    double *rec;                          // record ring [rec_chunks][np][16]
    Div32 rec_chunks;
    int32_t np, t_first, t_last, KC, diag, n_macro, total, has_lat;
    int32_t tile_filter;                  // 0: every tile of the launch; 1: all but the kTileWide ones (the LEAN kernel); 2: only those (its companion launch of the general kernel)
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
// Ring layout: rec[chunk][position][16] -- a chunk is a plane, a tile's records of one chunk are contiguous.  (Pairs of chunks side
// by side, so that a record pass moves 256-byte pieces, changed nothing: profiles/r03_rec_pairs_ab.txt.)
constexpr uint32_t kPosBytes = 128u;      // from one position's record to the next position's in the same chunk
__device__ __forceinline__ int64_t rec_elem(uint32_t ring_chunk, int64_t np, int64_t position)      // offset in doubles of a record
{
    return ((int64_t)ring_chunk * np + position) * kRec;
}
// Records move between HBM and their owning lanes through a per-wave LDS transpose: a lane owns a position (its record
// lives in registers), but a memory instruction in which every lane touches 16 bytes of a different record costs L2 one
// request per lane.  Through the transpose EIGHT neighbouring lanes load or store the 128 contiguous bytes of one record: a
// memory instruction moves eight whole lines.  (Rounds 1-3 moved 64-byte sectors, four lanes each, the first as soon as its
// eight ticks were done; a line that reaches the L2 in two parts is often written back half filled, and the launch then took
// 428 us instead of 360: DESIGN.md section 3c, profiles/r03_tile_store_whole.txt, r03_tile_store_line.txt, r03_tile_load_line.txt.)
constexpr int kStageLanes = 32;    // positions transposed at a time: half a wave (4.5 KiB of staging per wave)
constexpr int kStageStrideOut = 18;   // doubles per parked record: 128 bytes + 16 of padding and flags
// Lanes of one wave exchange data through its staging area without a workgroup barrier: a wave's LDS instructions
// execute in order.  The compiler still has to be told that other lanes wrote (it would reuse earlier reads).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS in doubles: X[2][TH + 2] | stage[waves][kStageLanes * kStageStrideOut] | aux[5][TH].  X: the tile's discharges of the
// last two ticks, each buffer followed by slots that always hold 0.0 (the "upstream position" of a reach that has none: the
// short tick reads three upstream values unconditionally); stage: the per-wave transpose areas; aux: UnitMuskingum's short tick
// keeps the channel discharge (aux[3]) and the previous tick's upstream sum (aux[4]) of every position here instead of in
// registers -- the record buffers leave none to spare, and a spill reload waits for every record load in flight -- and reads
// them with the upstream values at the top of a tick (the coefficients there too was slower: profiles/r03_unit_lean.txt).
constexpr int kTilePad = 2;
constexpr int kTileAux = 5;
constexpr size_t tile_lds_bytes(int threads)
{
    return (size_t)(2 * ((int64_t)threads + kTilePad) + (threads / 64) * kStageLanes * kStageStrideOut + kTileAux * threads) * sizeof(double);
}

// One task: KC record chunks of one tile, one position per thread.  R[16] is the record the ticks work on, in place
// (lateral in, discharge out); N[16] receives the NEXT chunk's record while the 16 ticks of this one run, so inside a task
// HBM traffic and tick arithmetic overlap and only the first chunk's load is exposed.  Whole 128-byte records are
// requested at once (a half record would cost the fabric a full line: measured, FETCH_SIZE 1.8x).
// LEAN: one sub-step per row, RapidMuskingum (the headline's case) or UnitMuskingum: the short tick below instead of the general
// one (the two do not fit one kernel: with the record buffers in 64 of 128 registers the allocator spills the records in flight).
// NOLAT (short tick only): channel-only routing (Muskingum.py:262-290) -- no lateral rows were turned into records, so a record slot
// holds whatever the ring held; only a ghost's slot means something (what its reach published).
template <int TH, bool UNIT, bool SUB, bool LEAN = false, bool NOLAT = false>
__global__ __launch_bounds__(TH, 4) void k_tile(const TileArgs a)      // 16 waves per CU: 1,024 / TH workgroups of 128 VGPRs
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int32_t total = a.total, K = a.KC * kRec;
    constexpr int THP = TH + kTilePad;      // doubles per discharge buffer
    // this wave's transpose area and the lane number, rebuilt from the thread index wherever they are used: kept in registers
    // across a task the allocator spills them, and the reload waits for every record load in flight
    auto stage_of = [&](int32_t t) { return lds + 2 * THP + (size_t)(t >> 6) * (kStageLanes * kStageStrideOut); };
    if (tid < 2) lds[tid * THP + TH] = 0.0;      // the zero slots; nothing else ever writes them (first barrier: before the first tick)
    auto ring = [&](int32_t chunk) {      // the records of one chunk as a buffer: position p at byte p * kPosBytes (np * kPosBytes < 2^32: choose_schedule)
        const uint32_t c = a.rec_chunks.mod((uint32_t)chunk);
        return make_rsrc(a.rec + rec_elem(c, a.np, 0), (uint32_t)a.np * kPosBytes);
    };

    // A workgroup takes the tiles t_last - blockIdx.x - g * gridDim.x, g = 0, 1, ... of this launch (highest level first:
    // the few tiles with ghosts start in the first round), so that the first record and the state of its NEXT tile are
    // requested while the current one still ticks.  A tile none of whose positions is active during its task has nothing
    // to do (pipeline fill and drain; a ghost has the lag of the reach it mirrors, so it is idle exactly when its owner
    // did not write its record) and is skipped.
    struct Task { int32_t tile, m, b0, b1, plain; };
    auto select = [&](int32_t from, Task &t) {
        for (int32_t c = from; c >= a.t_first; c -= (int32_t)gridDim.x) {
            const TileMeta tm = a.tiles[c];
            if (a.tile_filter && ((tm.flags & kTileWide) != 0) != (a.tile_filter == 2)) continue;
            const int32_t m = a.diag - tm.level;
            if (m < 0 || m >= a.n_macro) continue;
            if (m * K >= tm.lag_hi + total || (m + 1) * K <= tm.lag_lo) continue;
            t.tile = c; t.m = m; t.b0 = tm.b0; t.b1 = tm.b1;
            // LEAN: 1 = every position active on every tick of the task, nothing special in the tile: the tick without any test;
            // else the tile's flags | kTaskTested
            t.plain = (tm.flags == 0 && m * K >= tm.lag_hi && (m + 1) * K <= tm.lag_lo + total) ? 1 : (tm.flags | kTaskTested);
            return true;
        }
        return false;
    };
    Task cur;
    if (!select(a.t_last - (int32_t)blockIdx.x, cur)) return;

    // Eight lanes fetch (store) the eight 16-byte pieces of one record: in
    // flight a lane's N[] holds OTHER positions' pieces; receive() hands them to their owners through the wave's staging area.
    double R[kRec], N[kRec];
    // load j of a record: positions 8 j ... 8 j + 7 of the wave, eight lanes per 128-byte record
    auto issue_load = [&](__amdgpu_buffer_rsrc_t src, int32_t b0, int32_t b1, int j, bool real) {
        const int32_t t = fresh(tid), ln = t & 63;      // addresses are rebuilt at every use, not kept in registers across the task
        const int32_t pos = min(b0 + (t - ln) + 8 * j + (ln >> 3), b1 - 1);
        load_f64x2(src, real ? (uint32_t)pos * kPosBytes + (uint32_t)((ln & 7) * 16) : kDropAccess, N[2 * j], N[2 * j + 1]);
    };
    auto receive = [&]() {
        const int32_t tl = fresh(tid), lane = tl & 63;
        double *stage = stage_of(tl);
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // the records of half a wave at a time: loads 4 h ... 4 h + 3
#pragma unroll
            for (int g = 0; g < 4; ++g)
                reinterpret_cast<double2 *>(stage + (8 * g + (lane >> 3)) * kStageStrideOut)[lane & 7] =
                    make_double2(N[2 * (4 * h + g)], N[2 * (4 * h + g) + 1]);
            wave_lds_fence();
            if (lane / kStageLanes == h) {
                const double2 *src = reinterpret_cast<const double2 *>(stage + (lane % kStageLanes) * kStageStrideOut);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const double2 v = src[j]; R[2 * j] = v.x; R[2 * j + 1] = v.y; }
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
        const int32_t p = t.b0 + fresh(tid);      // formed here, not when the next task is chosen: kept across the chunk loop the address is spilled
        if (p < t.b1) {
            const int4 pm = a.pos[p];
            const uint32_t cc = (uint32_t)pm.w;
            const int32_t first_up = pm.y - t.b0;
            s.lg = pm.x; s.up = first_up | (int32_t)((cc & 0xFFFFu) << 16);
            s.xp = pm.z;
            if (UNIT) { s.uh = first_up + (int32_t)(cc >> 16); s.qch = a.sqch[p]; }
            if (SUB) s.isum = a.si[p];
            s.s_prev = a.ss[p];
            s.q = a.sq[p]; s.c1 = a.coef[3 * (int64_t)p]; s.c2 = a.coef[3 * (int64_t)p + 1]; s.c3 = a.coef[3 * (int64_t)p + 2];
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
    receive();          // the first tile's first record: the only record load nothing overlaps
    const bool has_lat = a.has_lat != 0;   // channel-only routing: the records only carry discharge

    for (;;) {
        int32_t lg = LEAN && UNIT ? 0 : st.lg, up = LEAN && UNIT ? 0 : st.up, xp = LEAN && UNIT ? 0 : st.xp, uh = LEAN && UNIT ? 0 : st.uh, sub = 0;
        double c1 = st.c1, c2 = st.c2, c3 = st.c3, s_prev = st.s_prev, qch = st.qch, isum = st.isum;
        const int32_t b0 = cur.b0, tau_begin = cur.m * K;
        lds[(size_t)((tau_begin + 1) & 1) * THP + tid] = st.q;       // tick tau_begin reads the buffer of tick tau_begin - 1
        // UnitMuskingum's short tick has six upstream slots where RapidMuskingum's has three, and no register to spare: left in
        // registers, the position's four constant words were spilled to scratch and reloaded every chunk behind a wait for every load
        // in flight (11 spills; profiles/r04_unit_spills.txt).  They wait in LDS instead (aux[0], aux[1]: this thread's own slots).
        constexpr bool kParked = LEAN && UNIT;
        int32_t *const parked = reinterpret_cast<int32_t *>(lds + 2 * THP + (TH / 64) * (kStageLanes * kStageStrideOut));
        if (kParked) {      // only this thread reads and writes these slots: no barrier
            double *aux = lds + 2 * THP + (TH / 64) * (kStageLanes * kStageStrideOut);
            reinterpret_cast<int2 *>(parked)[tid] = make_int2(st.up, st.uh);
            reinterpret_cast<int2 *>(parked + 2 * TH)[tid] = make_int2(st.lg, st.xp);
            aux[3 * TH + tid] = st.qch; aux[4 * TH + tid] = st.s_prev;
        }
        auto the_up = [&]() { return kParked ? parked[2 * tid] : fresh(up); };
        auto the_uh = [&]() { return kParked ? parked[2 * tid + 1] : fresh(uh); };
        auto the_lg = [&]() { return kParked ? parked[2 * TH + 2 * tid] : fresh(lg); };
        auto the_xp = [&]() { return kParked ? parked[2 * TH + 2 * tid + 1] : fresh(xp); };
        if (SUB && lg >= 0) {      // phase of the position's sub-step counter at the first tick of the task
            const int32_t ts0 = tau_begin - (lg & kLagMask);
            const uint32_t r = a.nsub.mod((uint32_t)(ts0 < 0 ? -ts0 : ts0));
            sub = ts0 >= 0 ? (int32_t)r : (r ? (int32_t)(a.nsub.d - r) : 0);
        }
        Task nxt;
        const bool has_next = select(cur.tile - (int32_t)gridDim.x, nxt);

        // The whole record at once: half a wave parks its 32 records, then all 64 lanes store them, eight lanes per 128-byte
        // line -- both sectors of a line in one instruction, so no line is ever written back half filled.
        auto store_record = [&](__amdgpu_buffer_rsrc_t dst) {
            const int32_t tl = fresh(tid), lane = tl & 63;
            double *stage = stage_of(tl);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (lane / kStageLanes == h) {
                    double2 *mine = reinterpret_cast<double2 *>(stage + (lane % kStageLanes) * kStageStrideOut);
#pragma unroll
                    for (int j = 0; j < 8; ++j) mine[j] = make_double2(R[2 * j], R[2 * j + 1]);
                    int32_t *word = reinterpret_cast<int32_t *>(mine + 8);
                    const int32_t lgw = LEAN ? the_lg() : lg;
                    word[0] = (lgw < 0 || (lgw & (kGhostBit | kTileGhostBit))) ? 1 : 0;   // not this tile's to write
                    if (!SUB) word[1] = (lgw >= 0 && (lgw & kTileExportBit)) ? the_xp() : -1;      // position of the ghost that mirrors this reach
                }
                wave_lds_fence();
                const int32_t t = fresh(tid), ln = t & 63;
                const uint32_t first = (uint32_t)(b0 + (t - ln) + h * kStageLanes) * kPosBytes;
#pragma unroll
                for (int g = 0; g < 4; ++g) {       // lanes 8i .. 8i+7: the eight 16-byte pieces of position 8 g + i
                    const int pm = 8 * g + (ln >> 3), piece = ln & 7;
                    const double2 *theirs = reinterpret_cast<const double2 *>(stage + pm * kStageStrideOut);
                    const double2 v = theirs[piece];
                    const bool skip = reinterpret_cast<const int32_t *>(theirs + 8)[0] != 0;
                    store_f64x2(dst, skip ? kDropAccess : first + (uint32_t)pm * kPosBytes + (uint32_t)piece * 16u, v);
                    if (!SUB) {     // the same line into the record of the ghost that mirrors the reach (always issued, see store_f64)
                        const int32_t gx = reinterpret_cast<const int32_t *>(theirs + 8)[1];
                        store_f64x2(dst, gx < 0 ? kDropAccess : (uint32_t)gx * kPosBytes + (uint32_t)piece * 16u, v);
                    }
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
                const double *rd = lds + (size_t)((tau + 1) & 1) * THP;
                double *wr = lds + (size_t)(tau & 1) * THP;
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
                        if (lgk & kExportBit) a.exports[(int64_t)ts * a.n_export + fresh(xp)] = qk;      // xp: the export slot (see TileArgs::xpos)
                    } else {
                        // explicit fma: every copy of this tick must round identically (split run == joint run)
                        qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, qk, lat)));
                        outv = qk; routed = true;
                        if (lgk & kExportBit) a.exports[(int64_t)ts * a.n_export + fresh(xp)] = qk;      // xp: the export slot (see TileArgs::xpos)
                    }
                    if (routed) {
                        if (SUB) {      // mean over the sub-steps of a row, written to the slot of the row's last sub-step
                            const double acc = (sub == 0 ? 0.0 : isum) + outv;
                            isum = acc;
                            if (sub + 1 == (int32_t)a.nsub.d) { const double v = acc * a.inv_nsub; R[s] = v > 0.0 ? v : 0.0; }
                        } else {
                            R[s] = outv;      // unclamped: the ghost that mirrors this reach gets a copy of the record; k_rec_out clamps
                        }
                    }
                }
                if (SUB) sub = sub + 1 == (int32_t)a.nsub.d ? 0 : sub + 1;
                s_prev = s_cur;
                wr[t] = qk;
                // With sub-steps a record slot holds a row mean, not the tick's discharge: a reach mirrored by a ghost of another
                // tile sends 8 bytes per tick into the ghost's record, always issued (see store_f64).  Without, the ghost gets a
                // copy of the whole record in store_record.
                if (SUB) store_f64(rec_cur, (lgk >= 0 && (lgk & kTileExportBit)) ? (uint32_t)fresh(xp) * kPosBytes + (uint32_t)s * 8u : kDropAccess, qk);
                barrier_lds();
            }
        };

        // The short tick (LEAN).  A tick is: own value and three upstream values from LDS (a reach with fewer reads the zero slot
        // instead), three fused multiply-adds, one LDS write, the barrier -- no flag test, no loop.  During pipeline fill and drain
        // (a task in which some position is not yet, or no longer, active on some tick) an activity test and four selects keep an
        // idle position's value and record slot as they are; every other task -- 99 % of a year's -- runs without.  1M reaches,
        // K = 64: k_tile 241 -> us per launch (profiles/r03_tile_diet.txt).  The sum is the general tick's in the same order
        // (first + second + third; the general tick's leading 0.0 + and the zero slot's + 0.0 can only change the sign of a zero).
        // A ghost position needs nothing special: its coefficients are zero (rr_plan_set_coeffs) and it has no upstream position,
        // so the three multiply-adds hand back its record slot, which is what it publishes.  A tile with a reach of more than three
        // upstream reaches is left to a companion launch of the general kernel (TileArgs::tile_filter); boundary exports of a
        // partitioned network are stored from the record registers when a half record is complete (store_exports).
        const int32_t kind = LEAN ? cur.plain : 0;
        int32_t own_b = 0, up0_b = 0, up1_b = 0, up2_b = 0, lagm = 0;      // LDS byte offsets inside a discharge buffer: own slot, the three upstream slots (or the zero slot)
        int32_t hw0_b = 0, hw1_b = 0, hw2_b = 0;      // UnitMuskingum: the headwater tributaries' slots (they come first in the upstream range); up*_b then hold the inner ones
        bool ghostm = false;                          // NOLAT: this position is a ghost (its record slot is its forcing)
        auto lds_at = [&](int parity, int32_t byte) -> double & { return *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + parity * (THP * 8) + byte); };
        // record loads of the short path: one address register per chunk, the piece of load j is an immediate / scalar offset
        auto issue_load_plain = [&](__amdgpu_buffer_rsrc_t src, uint32_t voff, int j) {
            // the piece's offset is part of the per-lane offset (a compile-time constant per unrolled load, folded into the
            // instruction's immediate where it fits): the range check of a raw buffer covers that sum, not a scalar offset
            const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(src, (int)(voff + (uint32_t)j * 8u * kPosBytes), 0, 0);
            double2 v;
            __builtin_memcpy(&v, &bits, sizeof v);
            N[2 * j] = v.x; N[2 * j + 1] = v.y;
        };
        auto ticks_plain = [&](auto tested, int32_t tau0, int half, __amdgpu_buffer_rsrc_t rec_next, uint32_t nvoff) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int s = 8 * half + s8;      // tau0 is a multiple of 16: the parity of the tick is the parity of s
                if (half == 0) issue_load_plain(rec_next, nvoff, s8);
                const double q_old = lds_at((s + 1) & 1, own_b);
                const double s_cur = (lds_at((s + 1) & 1, up0_b) + lds_at((s + 1) & 1, up1_b)) + lds_at((s + 1) & 1, up2_b);
                double qk;
                if (UNIT) {      // _numba_kernels.py:142-167 in gather form; a headwater and a ghost have zero coefficients: r = 0, the slot is republished
                    constexpr int kAux = (2 * THP + (TH / 64) * (kStageLanes * kStageStrideOut)) * 8;      // byte offset of aux[0][0]
                    auto aux_at = [&](int k) -> double & { return *reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + kAux + k * (TH * 8) + own_b); };
                    const double s_hw = (lds_at((s + 1) & 1, hw0_b) + lds_at((s + 1) & 1, hw1_b)) + lds_at((s + 1) & 1, hw2_b);
                    const double qc = aux_at(3);
                    const double r = __builtin_fma(c1, s_hw + s_cur, __builtin_fma(c2, s_hw + aux_at(4), c3 * qc));
                    qk = r + R[s];
                    if (decltype(tested)::value) {
                        const bool active = (uint32_t)(tau0 + s - lagm) < (uint32_t)total;
                        R[s] = active ? qk : R[s];
                        aux_at(3) = active ? r : qc;
                        qk = active ? qk : q_old;
                    } else {
                        R[s] = qk;
                        aux_at(3) = r;
                    }
                    aux_at(4) = s_cur;
                } else {
                    qk = __builtin_fma(c1, s_cur, __builtin_fma(c2, s_prev, __builtin_fma(c3, q_old, (NOLAT && !ghostm) ? 0.0 : R[s])));
                    if (decltype(tested)::value) {
                        const bool active = (uint32_t)(tau0 + s - lagm) < (uint32_t)total;
                        R[s] = active ? qk : R[s];
                        qk = active ? qk : q_old;
                    } else {
                        R[s] = qk;      // unclamped: k_rec_out clamps
                    }
                    s_prev = s_cur;
                }
                lds_at(s & 1, own_b) = qk;
                barrier_lds();
            }
        };
        // boundary exports of a partitioned network: with one sub-step per row the record slot of a tick IS the discharge of that
        // sub-step, so the export series is written from the half record that has just been completed
        auto store_exports = [&](int32_t tau0, int half) {
            const int32_t lgk = the_lg();
            if (lgk >= 0 && (lgk & kExportBit)) {
                const int32_t ts0 = tau0 + 8 * half - (lgk & kLagMask), slot = the_xp();      // xp: the export slot (see TileArgs::pos)
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8)
                    if ((uint32_t)(ts0 + s8) < (uint32_t)total) a.exports[(int64_t)(ts0 + s8) * a.n_export + slot] = R[8 * half + s8];
            }
        };

        barrier_lds();      // the buffer of tick tau_begin - 1 is in place (and every wave has left the previous tile)
        for (int32_t cc = 0; cc < a.KC; ++cc) {
            const int32_t chunk = cur.m * a.KC + cc, tau0 = chunk * kRec;
            const bool last = cc + 1 == a.KC;
            // what arrives during this chunk: the tile's next chunk, or -- in its last one -- the first chunk of the next tile
            const __amdgpu_buffer_rsrc_t rec_next = ring(last ? nxt.m * a.KC : chunk + 1);
            const int32_t nb0 = last ? nxt.b0 : b0, nb1 = last ? nxt.b1 : cur.b1;
            if constexpr (LEAN) {
                // a position past the end of the next tile reads whatever follows it in the ring (or zeros past its end): never used
                const int32_t t = fresh(tid), ln = t & 63, upk = the_up();
                const int32_t cnt = (int32_t)((uint32_t)upk >> 16), u0s = upk & 0xFFFF;
                own_b = t * 8;
                if (NOLAT) { const int32_t lgs = the_lg(); ghostm = lgs >= 0 && (lgs & (kGhostBit | kTileGhostBit)) != 0; }
                if (UNIT) {
                    const int32_t nh = the_uh() - u0s, ni = cnt - nh, i0 = u0s + nh;      // headwater tributaries [u0s, uh), inner ones [uh, u0s + cnt)
                    hw0_b = (nh >= 1 ? u0s : TH) * 8; hw1_b = (nh >= 2 ? u0s + 1 : TH) * 8; hw2_b = (nh >= 3 ? u0s + 2 : TH) * 8;
                    up0_b = (ni >= 1 ? i0 : TH) * 8; up1_b = (ni >= 2 ? i0 + 1 : TH) * 8; up2_b = (ni >= 3 ? i0 + 2 : TH) * 8;
                } else {
                    up0_b = (cnt >= 1 ? u0s : TH) * 8; up1_b = (cnt >= 2 ? u0s + 1 : TH) * 8; up2_b = (cnt >= 3 ? u0s + 2 : TH) * 8;
                }
                const uint32_t nvoff = (uint32_t)(nb0 + (t - ln) + (ln >> 3)) * kPosBytes + (uint32_t)(ln & 7) * 16u;      // unsigned: 2^25 positions x 128 B
                const __amdgpu_buffer_rsrc_t src = (!last || has_next) ? rec_next : make_rsrc(a.rec, 0u);
                if (kind == 1) {
                    ticks_plain(std::false_type(), tau0, 0, src, nvoff);
                    ticks_plain(std::false_type(), tau0, 1, src, nvoff);
                } else {
                    { const int32_t lgs = the_lg(); lagm = lgs < 0 ? 0x40000000 : (lgs & kLagMask); }      // a slot past the end of the tile is never active
                    ticks_plain(std::true_type(), tau0, 0, src, nvoff);
                    if (kind & kTileExports) store_exports(tau0, 0);
                    ticks_plain(std::true_type(), tau0, 1, src, nvoff);
                    if (kind & kTileExports) store_exports(tau0, 1);
                }
            } else {
                ticks(tau0, 0, rec_next, nb0, nb1, !last || has_next);
                ticks(tau0, 1, rec_next, nb0, nb1, false);
            }
            store_record(rec_cur);      // after the sixteenth tick: whole lines, never a half-filled write-back
            // the next tile's state: small, and only the wait for it is exposed between two tiles.  LEAN asks for it after the
            // chunk loop: requested inside, its thirteen registers are live across every chunk beside both record buffers, and
            // what the allocator spills it reloads behind a wait for every load in flight
            if (!LEAN && last && has_next) load_state(nxt, st);
            receive();      // the record that has had 16 ticks to arrive (zeros after the last chunk of the last tile)
            rec_cur = rec_next;
        }
        if ((LEAN ? the_lg() : lg) >= 0) {
            const int32_t p = b0 + tid;
            const double *aux = lds + 2 * THP + (TH / 64) * (kStageLanes * kStageStrideOut);
            a.sq[p] = lds[(size_t)((tau_begin + K - 1) & 1) * THP + tid]; a.ss[p] = (LEAN && UNIT) ? aux[4 * TH + tid] : s_prev;
            if (UNIT) a.sqch[p] = (LEAN && UNIT) ? aux[3 * TH + tid] : qch;
            if (SUB) a.si[p] = isum;
        }
        if (LEAN && has_next) load_state(nxt, st);
        if (!has_next) break;
        cur = nxt;
    }
}

// sq = q0 at every position (a ghost starts from the state of the reach it mirrors), ss = sum of the upstream q0
__global__ __launch_bounds__(kBlock) void k_tile_state_in(double *sq, double *ss, double *si, const double *q_t, const int32_t *perm,
                                                          const int4 *pos, int32_t np)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= np) return;
    double s = 0.0;
    const int4 pm = pos[p];
    const int32_t u0 = pm.y, u1 = u0 + (int32_t)((uint32_t)pm.w & 0xFFFFu);
    for (int32_t u = u0; u < u1; ++u) s += q_t[perm[u]];
    sq[p] = q_t[perm[p]]; ss[p] = s; si[p] = 0.0;
}

// UnitMuskingum state for the time-tiled kernel: published discharge = q_full on inner reaches (0 on headwaters until
// their first tick), ss = sum over the INNER tributaries only (the headwater ones come first), qch = channel discharge.
// full[i] / chan[i]: q_full / q_ch scattered to params order, zeros on headwaters (k_unit_scatter).
__global__ __launch_bounds__(kBlock) void k_tile_unit_state_in(double *sq, double *ss, double *si, double *sqch, const double *full,
                                                               const double *chan, const int32_t *perm, const int4 *pos, int32_t np)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p >= np) return;
    double s = 0.0;
    const int4 pm = pos[p];
    const uint32_t cc = (uint32_t)pm.w;
    const int32_t u0 = pm.y + (int32_t)(cc >> 16), u1 = pm.y + (int32_t)(cc & 0xFFFFu);
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

// the direct row path keeps UnitMuskingum's state in params order: the inner reaches' pairs leave from there
__global__ __launch_bounds__(kBlock) void k_unit_gather(double *q_ch, double *q_full, const double *chan, const double *full, const int32_t *inner_idx, int32_t n_inner)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_inner) return;
    q_full[k] = full[inner_idx[k]]; q_ch[k] = chan[inner_idx[k]];
}

// the skeleton of the direct row path: every position that is not a ghost hands its reach's discharge back
__global__ __launch_bounds__(kBlock) void k_skel_state_out(double *q_t, const double *sq, const int32_t *perm, const int4 *pos, int32_t np)
{
    const int32_t p = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (p < np && !(pos[p].x & (kGhostBit | kTileGhostBit))) q_t[perm[p]] = sq[p];
}

__global__ __launch_bounds__(kBlock) void k_tile_state_out(double *q_t, const double *sq, const int32_t *inv, int32_t n)
{
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) q_t[i] = sq[inv[i]];
}

}  // namespace
