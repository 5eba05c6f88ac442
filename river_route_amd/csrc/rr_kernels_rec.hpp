// rr_kernels_rec.hpp -- the record passes: params-order rows <-> tick-indexed records, with the fused float32 / convolution forms, and the stand-alone resample-cast.
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- record permutation (one pass each way), see k_tile ----
// Column i of the params-order rows is position p = inv[i] with lag L = 16 * sh + o.  Tick-row r of that column (tick-row =
// routing sub-step: runoff row r / nsub, sub-step r % nsub) is slot (r + L) % 16 of record (r + L) / 16, so with B = 8 records
// per batch the 128 tick-rows [128 j - o, 128 j + 128 - o) are exactly the records B j + sh .. B j + sh + B - 1.  k_rec_in
// reads the runoff rows behind the 143 tick-rows [128 j - 15, 128 j + 128) of a 32-column tile coalesced into LDS (all loads
// in flight before the first LDS write) and writes B whole 128-byte records per column (8 lanes x 16 B per record), every
// sub-step slot of a row holding the row's lateral value; k_rec_out reads B + 1 records per column the same way and writes
// the tile's rows of the batch coalesced: the slot of a row's LAST sub-step holds the row's mean discharge.  B = 16 halves
// the share of rows / records read twice at a batch's edge but needs 72 KB of LDS per workgroup: 1 % slower when every row
// comes from HBM (380 vs 384 ms per year), faster only while a short cyclic forcing array is found in a cache
// (profiles/r02_rec_batch.txt).
// Tile shapes, each the winner of an A/B on the GPU box: 256 threads (128 to 1,024: within 0.5 %, profiles/r03_rec_threads_ab.txt),
// 32 columns (16 to 64: the same, r03_rec_tile_width_ab.txt), one tile per workgroup (persistent workgroups with the next tile's
// loads in flight: 427 / 469 us per 128 rows against 419 / 434, r03_rec_persistent_ab.txt), 8 records per batch (r02_rec_batch.txt).
constexpr int kRecCols = 32, kRecBatch = 8, kRecThreads = 256;
constexpr int kRecInThreads = kRecThreads, kRecOutThreads = kRecThreads;      // workgroup sizes of k_rec_in / k_rec_out
constexpr int kRecInCols = kRecCols, kRecOutCols = kRecCols;                  // columns of a tile of k_rec_in / k_rec_out
constexpr int kRecRows = 16 * kRecBatch;    // tick-rows of one batch

struct RecPermArgs {
    double *rec;
    Div32 rec_chunks;
    int64_t n, np, T, total, batch;   // T runoff rows, total = T * nsub tick-rows
    Div32 nsub;
    const int2 *colmeta;      // per params column: {position, lag}
    const double *scale;      // c4dt in PARAMS order (RapidMuskingum: the ring holds c4dt * lateral) or NULL
    RowView rows;             // params-order rows (source of k_rec_in, destination of k_rec_out)
    const float *rows_in32;   // k_rec_in<..., IN32>: the source rows are float32 (a qlateral file stored that way: 4 B read instead of 8, exact in float64); same shape as `rows`
    float *rows32;            // k_rec_out: float32 destination with `factor` rows averaged (router post-processing), or NULL
    Div32 factor;
    const int32_t *cols;      // k_rec_out: column of the destination rows each of the n records' columns goes to (the holes of the direct row path), or NULL: column i
    int32_t swizzle;          // column tiles in XCD-contiguous order (rr_common.hpp: xcd_swizzle)
    int32_t clamp;            // k_rec_out: 0 records hold final values (sub-steps), 1 clamp at zero, 2 clamp all but headwater columns (UnitMuskingum)
    uint32_t in32_sel, out32_sel;   // byte selectors of the float32 rows (rr_plan_set_row_format): kSelNative, or kSelSwap for a big-endian file's rows
};

constexpr int32_t kColHeadwater = 1 << 30;      // colmeta[].y: lag | this flag

constexpr int kRecTileRows = 16 * kRecBatch + 15;      // tick-rows behind one batch of records
constexpr int kRecTileLd = kRecCols + 1;

// Ring chunk `first + ahead`, first < chunks and ahead = lag / 16 + a record of the batch: the ring is longer than the deepest
// lag plus two batches (rr_exec.hpp: choose_schedule), so one subtraction wraps it -- the division costs 12 VALU instructions,
// and the fused convolution is bound by those (profiles/r03_uh_diet.txt).
__device__ __forceinline__ uint32_t ring_chunk(const Div32 &chunks, uint32_t first, uint32_t ahead)
{
    const uint32_t c = first + ahead;
    return c >= chunks.d ? c - chunks.d : c;
}

// Second half of the in-pass: the LDS tile (row = runoff row - row_first, kRecTileLd doubles per row) becomes records.
// BATCH records per column from tick-row kRecRows * a.batch - 15 on (the fused convolution takes two batches at a time).
// smeta / sscale: the tile's column metadata and scale in LDS (k_rec_in keeps them there: loaded with the rows, no registers
// held across the stores), or NULL: read from the plan's arrays.
template <bool SUB, int THREADS = kRecThreads, int BATCH = kRecBatch, bool LDSMETA = false, int COLS = kRecCols>
__device__ __forceinline__ void write_records(const RecPermArgs &a, const double *tile, int64_t col0, int64_t tick_first, int64_t row_first,
                                              const int2 *smeta = nullptr, const double *sscale = nullptr)
{
    constexpr int R = 16 * BATCH + 15;
    const int tid = threadIdx.x;
    static_assert(COLS * BATCH * 8 % THREADS == 0, "records of a tile must divide among the threads");
    constexpr int IT = COLS * BATCH * 8 / THREADS;
    int2 meta[LDSMETA ? 1 : IT];
    double f[LDSMETA ? 1 : IT];
    if constexpr (!LDSMETA) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {     // all metadata loads first: they are independent
            const int64_t i = col0 + (it * THREADS + tid) / (8 * BATCH);
            meta[it] = i < a.n ? a.colmeta[i] : make_int2(-1, 0);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int64_t i = col0 + (it * THREADS + tid) / (8 * BATCH);
            f[it] = (a.scale && i < a.n) ? a.scale[i] : 1.0;
        }
    }
    const uint32_t chunk_first = a.rec_chunks.mod((uint32_t)kRecBatch * (uint32_t)a.batch);
#pragma unroll(LDSMETA ? (SUB ? 2 : 4) : IT)
    for (int it = 0; it < IT; ++it) {
        const int piece = it * THREADS + tid;       // (column, record, 16-byte part): 8 consecutive lanes = one record
        const int c = piece / (8 * BATCH), k = (piece >> 3) % BATCH, part = piece & 7;
        const int2 m = LDSMETA ? smeta[c] : meta[LDSMETA ? 0 : it];
        const double scale = LDSMETA ? sscale[c] : f[LDSMETA ? 0 : it];
        const int32_t p = m.x;
        if (p < 0) continue;
        const int32_t lag = m.y & kLagMask;
        const int o = lag & 15;
        const uint32_t chunk = ring_chunk(a.rec_chunks, chunk_first, (uint32_t)(lag >> 4) + k);
        const int r = 15 - o + 16 * k + 2 * part;       // tick-row tick_first + r
        double v0, v1;
        if (SUB) {
            const int64_t t0 = tick_first + r, t1 = t0 + 1;
            uint32_t s;
            const int r0 = t0 < 0 ? 0 : (int)((int64_t)a.nsub.div((uint32_t)t0, s) - row_first);
            const int r1 = t1 < 0 ? 0 : (int)((int64_t)a.nsub.div((uint32_t)t1, s) - row_first);
            v0 = tile[min(r0, R - 1) * (COLS + 1) + c] * scale; v1 = tile[min(r1, R - 1) * (COLS + 1) + c] * scale;
        } else {
            v0 = tile[r * (COLS + 1) + c] * scale; v1 = tile[(r + 1) * (COLS + 1) + c] * scale;
        }
        double2 *dst = reinterpret_cast<double2 *>(a.rec + rec_elem(chunk, a.np, p)) + part;
        typedef double d2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<d2 *>(dst) = d2{v0, v1};
    }
}

// One column tile per workgroup: all of its row loads in flight before the first LDS write, the tile's column metadata and
// scale travelling with the rows into LDS (no registers held across the stores).  The rows alone read in 186 us per 128 rows at 1M
// reaches, the records alone store in 199 us, the pass takes 415: within 7 % of the two one after the other
// (profiles/r03_rec_probe.txt, r03_alias_kernel_times.txt).
template <bool SUB, bool IN32 = false>
__global__ __launch_bounds__(kRecInThreads) void k_rec_in(const RecPermArgs a)
{
    constexpr int R = kRecTileRows;
    __shared__ double tile[R * (kRecInCols + 1)];
    __shared__ int2 smeta[kRecInCols];
    __shared__ double sscale[kRecInCols];
    const int tid = threadIdx.x;
    const int64_t tick_first = kRecRows * a.batch - 15;                 // may be negative in the first batch
    uint32_t sub_unused;
    const int64_t row_first = SUB ? (int64_t)a.nsub.div((uint32_t)(tick_first < 0 ? 0 : tick_first), sub_unused) : tick_first;
    constexpr int RPT = (R + kRecInThreads / kRecInCols - 1) / (kRecInThreads / kRecInCols);
    const int c = tid % kRecInCols, r0 = tid / kRecInCols;
    const int need = SUB ? (int)((uint32_t)(R - 1) / a.nsub.d) + 2 : R;     // runoff rows behind the batch's tick-rows
    const uint32_t n_tiles = (uint32_t)((a.n + kRecInCols - 1) / kRecInCols);
    if (blockIdx.x >= n_tiles) return;
    const int64_t col0 = (int64_t)(a.swizzle ? xcd_swizzle(blockIdx.x, n_tiles) : blockIdx.x) * kRecInCols;
    double v[RPT];
    int2 cm = make_int2(-1, 0);
    double cs = 1.0;
    {   // branch-free: out-of-range rows / columns are clamped here and zeroed on the way into LDS
        const int64_t i = min(col0 + c, a.n - 1);
        if (tid < kRecInCols) {
            cm = col0 + c < a.n ? a.colmeta[i] : make_int2(-1, 0);
            cs = a.scale ? a.scale[i] : 1.0;
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int64_t t = row_first + min(r0 + q * (kRecInThreads / kRecInCols), need - 1);
            const int64_t off = a.rows.offset(t < 0 ? 0 : (t >= a.T ? a.T - 1 : t)) + i;
            v[q] = IN32 ? (double)f32_from_file(a.rows_in32[off], a.in32_sel) : a.rows.base[off];
        }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = r0 + q * (kRecInThreads / kRecInCols);
        const int64_t row = row_first + r;
        if (r < R) tile[r * (kRecInCols + 1) + c] = (row >= 0 && row < a.T && col0 + c < a.n) ? v[q] : 0.0;
    }
    if (tid < kRecInCols) { smeta[c] = cm; sscale[c] = cs; }
    __syncthreads();
    write_records<SUB, kRecInThreads, kRecBatch, true, kRecInCols>(a, tile, col0, tick_first, row_first, smeta, sscale);
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
// Columns of a tile of the fused convolution: 16 (128-byte row pieces) against the 32 of the plain passes: 52.6 against 55.2 ms
// for BASELINE config 4 (profiles/r03_small_networks_and_uh16.txt).
// BATCH = batches of kRecBatch records (128 tick-rows) per launch, 1 or 2: the n_ks - 1 depth rows before a tile and the taps
// are read once per tile, so the pair moves 19.4 B per value where the single batch moves 22.9 (n_ks = 48); a workgroup has
// 32 threads per record: 16 columns x (2 x records) groups of 9 rows.
constexpr int kUhCols = 16, kUhTileLd = kUhCols + 1;
constexpr int kUhMaxBatches = 2;
constexpr int uh_threads(int batches) { return 32 * kRecBatch * batches; }
constexpr int uh_tile_rows(int batches) { return 16 * kRecBatch * batches + 15; }
constexpr int uh_groups(int batches) { return uh_threads(batches) / kUhCols; }
constexpr int uh_rows_per_thread(int batches) { return (uh_tile_rows(batches) + uh_groups(batches) - 1) / uh_groups(batches); }
constexpr int uh_win_rows(int batches) { return uh_groups(batches) * uh_rows_per_thread(batches); }      // rows the windows start in: one more than the tile has
// depth rows [win_rows + nk - 1] (the last ones are padding: the windows of the threads past the tile's end reach them) | taps [nk]
constexpr size_t rec_in_uh_lds_bytes(int nk, int batches) { return (size_t)((uh_win_rows(batches) + nk - 1) + nk) * kUhTileLd * sizeof(double); }

// The kernel issues 432 multiply-adds per thread at NK = 48 and used to issue 670 other vector instructions around them: a
// division per cyclic row and record, clamps and 64-bit compares per LDS access (SQ_INSTS_VALU, profiles/r03_b_sq_counters_*);
// at 16 lanes per cycle that was 450 of its 770 us.  Now: LDS accesses at compile-time offsets from one address per thread
// (the tile is padded instead of the indices clamped), the cyclic rows stepped from one division, 32-bit range tests, the
// carried-in state behind a branch the whole launch takes the same way.
template <bool SUB, int NK, int BATCHES, bool IN32 = false>      // IN32: the depth rows are float32 (a.rows_in32, the shape of a.rows)
__global__ __launch_bounds__(uh_threads(BATCHES)) void k_rec_in_uh(const RecPermArgs a, const UhArgs u)
{
    constexpr int R = uh_tile_rows(BATCHES), G = uh_groups(BATCHES), RP = uh_rows_per_thread(BATCHES);
    constexpr int ROWS = uh_win_rows(BATCHES) + NK - 1, DATA = R + NK - 1;      // depth rows in LDS / those that hold data
    extern __shared__ __attribute__((aligned(16))) double uh_lds[];
    double *dt = uh_lds;                          // [ROWS][kUhTileLd] depth rows row_first - (NK - 1) ...
    double *tp = uh_lds + ROWS * kUhTileLd;      // [NK][kUhTileLd] taps
    const int tid = threadIdx.x, c = tid % kUhCols, g = tid / kUhCols;
    const int64_t col0 = (int64_t)(a.swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * kUhCols;
    const int64_t tick_first = kRecRows * a.batch - 15;
    uint32_t sub_unused;
    const int64_t row_first = SUB ? (int64_t)a.nsub.div((uint32_t)(tick_first < 0 ? 0 : tick_first), sub_unused) : tick_first;
    const int need = SUB ? (int)((uint32_t)(R - 1) / a.nsub.d) + 2 : R;
    const int64_t i = min(col0 + c, a.n - 1);
    const bool live = col0 + c < a.n;
    const int32_t tb = (int32_t)row_first - (NK - 1) + g;      // depth row of this thread's first load (negative in a call's first launch); T < 2^31
    {   // all loads in flight first (branch-free: rows outside the call and columns past the end load something valid and are zeroed afterwards)
        constexpr int DPT = (ROWS + G - 1) / G, TPT = (NK + G - 1) / G;
        double dv[DPT], tv[TPT];
        if constexpr (SUB) {
#pragma unroll
            for (int q = 0; q < DPT; ++q) {
                const int64_t t = row_first - (NK - 1) + min(g + q * G, need + NK - 2);
                const int64_t off = a.rows.offset(t < 0 ? 0 : (t >= a.T ? a.T - 1 : t)) + i;
                dv[q] = IN32 ? (double)f32_from_file(a.rows_in32[off], a.in32_sel) : a.rows.base[off];
            }
        } else {
            // load q reads the cyclic row (tb + G q - t0) mod rows: one division, for the first q whose row cannot be negative,
            // and steps of G mod rows either way from there (a row before the call's first steps to some row of the array: loaded, not used)
            constexpr int QM = (NK - 1 + 15 + G - 1) / G;
            static_assert(QM < DPT, "the tile is longer than the rows before it");
            const uint32_t d = a.rows.rows.d, step = a.rows.rows.mod((uint32_t)G);
            uint32_t m[DPT];
            m[QM] = a.rows.rows.mod((uint32_t)(tb + G * QM - (int32_t)a.rows.t0));
#pragma unroll
            for (int q = QM + 1; q < DPT; ++q) { const uint32_t x = m[q - 1] + step; m[q] = x >= d ? x - d : x; }
#pragma unroll
            for (int q = QM - 1; q >= 0; --q) m[q] = m[q + 1] >= step ? m[q + 1] - step : m[q + 1] + d - step;
#pragma unroll
            for (int q = 0; q < DPT; ++q) {
                const int64_t off = (int64_t)m[q] * a.rows.ld + i;
                dv[q] = IN32 ? (double)f32_from_file(a.rows_in32[off], a.in32_sel) : a.rows.base[off];
            }
        }
#pragma unroll
        for (int q = 0; q < TPT; ++q) tv[q] = u.kernel[(int64_t)min(g + q * G, u.n_ks - 1) * a.n + i];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int r = g + q * G;
            const bool inside = (uint32_t)(tb + q * G) < (uint32_t)a.T;      // 0 <= row < T in one test
            if (r < ROWS) dt[r * kUhTileLd + c] = (live && inside && r < DATA) ? dv[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            const int k = g + q * G;
            if (k < NK) tp[k * kUhTileLd + c] = (live && k < u.n_ks) ? tv[q] : 0.0;
        }
    }
    __syncthreads();
    const int rb = g * RP;      // first output row of this thread
    const bool mine = !SUB || rb < need;
    double acc[RP];
    if (mine) {
        const double *wp = dt + rb * kUhTileLd + c;       // depth row (row_first + rb + q - (NK - 1)) at wp[q * kUhTileLd]: rb + W - 1 < ROWS
        if (row_first < (int64_t)u.n_ks) {      // the carried-in state reaches the first n_ks rows of a call: a few launches, all of whose workgroups come here
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int64_t t = row_first + rb + j;
                acc[j] = (live && t >= 0 && t < u.n_ks && t < a.T) ? u.state[t * a.n + i] : 0.0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RP; ++j) acc[j] = 0.0;
        }
        // Half of the taps in registers at a time and the window streamed past them (each depth value meets the taps it is
        // multiplied with while it is in one register): 24 taps + 9 sums instead of a 56-value window, 90 registers instead of
        // 142, five workgroups per CU instead of three.  Every sum still adds its products in the order k = 0, 1, ...
        const double *kp = tp + c;
        constexpr int KH = NK / 2;
        static_assert(NK % 2 == 0, "taps are padded to an even count");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double tap[KH];
#pragma unroll
            for (int k = 0; k < KH; ++k) tap[k] = kp[(h * KH + k) * kUhTileLd];
#pragma unroll
            for (int w = RP - 1 + NK - 1 - h * KH; w >= NK - 1 - h * KH - (KH - 1); --w) {      // window index j + NK - 1 - k, k ascending for each j
                const double x = wp[w * kUhTileLd];
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const int k = j + NK - 1 - w - h * KH;
                    if (k >= 0 && k < KH) acc[j] = __builtin_fma(tap[k], x, acc[j]);
                }
            }
        }
    }
    __syncthreads();      // every window is in registers: the depth tile's space now takes the outputs
    if (mine) {
        double *op = dt + rb * kUhTileLd + c;
#pragma unroll
        for (int j = 0; j < RP; ++j) op[j * kUhTileLd] = acc[j];      // rb + j < win_rows: the rows past the tile's end are written and never read
    }
    __syncthreads();
    write_records<SUB, uh_threads(BATCHES), kRecBatch * BATCHES, false, kUhCols>(a, dt, col0, tick_first, row_first);
}

// OUT32: the router's post-processing fused in (TransformMuskingum.py:128-142): mean over `factor` consecutive rows
// (sequential sum, one division, as numpy reduces a strided axis) and the float32 cast; 128 % (factor * nsub) == 0.
// One column tile per workgroup, every record read of it in flight at once.
template <bool SUB, bool OUT32>
__global__ __launch_bounds__(kRecOutThreads) void k_rec_out(const RecPermArgs a)
{
    constexpr int S = 16 * (kRecBatch + 1);
    __shared__ double recs[kRecOutCols][S + 1];
    const int tid = threadIdx.x;
    static_assert(kRecOutCols * (kRecBatch + 1) * 8 % kRecOutThreads == 0, "record pieces of a tile must divide among the threads");
    constexpr int IT = kRecOutCols * (kRecBatch + 1) * 8 / kRecOutThreads;
    const uint32_t n_tiles = (uint32_t)((a.n + kRecOutCols - 1) / kRecOutCols);
    if (blockIdx.x >= n_tiles) return;
    const int64_t col0 = (int64_t)(a.swizzle ? xcd_swizzle(blockIdx.x, n_tiles) : blockIdx.x) * kRecOutCols;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v[IT];
    const uint32_t chunk_first = a.rec_chunks.mod((uint32_t)kRecBatch * (uint32_t)a.batch);
    {   // a column past the end reads position 0 and is not written
        int2 meta[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {      // every metadata load first: the record loads depend on them, one wait for all
            const int64_t i = col0 + (it * kRecOutThreads + tid) / ((kRecBatch + 1) * 8);
            meta[it] = i < a.n ? a.colmeta[i] : make_int2(0, 0);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int piece = it * kRecOutThreads + tid;
            const int k = (piece >> 3) % (kRecBatch + 1), part = piece & 7;
            const uint32_t chunk = ring_chunk(a.rec_chunks, chunk_first, (uint32_t)((meta[it].y & kLagMask) >> 4) + k);
            v[it] = reinterpret_cast<const d2 *>(a.rec + rec_elem(chunk, a.np, meta[it].x))[part];
        }
    }
    const int64_t tick0 = kRecRows * a.batch;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int piece = it * kRecOutThreads + tid;
        const int c = piece / ((kRecBatch + 1) * 8), k = (piece >> 3) % (kRecBatch + 1), part = piece & 7;
        recs[c][16 * k + 2 * part] = v[it].x;
        recs[c][16 * k + 2 * part + 1] = v[it].y;
    }
    __syncthreads();
    const int c = tid % kRecOutCols;
    const int64_t i = col0 + c;
    if (i >= a.n) return;
    const int32_t my = a.colmeta[i].y;
    const int o = my & 15;
    const int64_t io = a.cols ? a.cols[i] : i;      // destination column; the float32 rows' pitch is the float64 rows' (rows.ld)
    // single sub-step: the records hold the unclamped discharge (a ghost's record is a copy of its reach's); the reference's
    // clip at zero (_numba_kernels.py:80) happens here.  UnitMuskingum leaves its headwaters' lateral inflow as it is (:122-123).
    const bool clamp = !SUB && (a.clamp == 1 || (a.clamp == 2 && !(my & kColHeadwater)));
    auto out = [&](double x) { return clamp && !(x > 0.0) ? 0.0 : x; };
    if (OUT32) {
        // output row q averages runoff rows [q * factor, (q + 1) * factor), each the slot of its last sub-step
        const int step = (int)(a.factor.d * (SUB ? a.nsub.d : 1u));          // tick-rows per output row, divides 128
        const int64_t q0 = tick0 / step;
        for (int q = tid / kRecOutCols; q < kRecRows / step; q += kRecOutThreads / kRecOutCols) {
            if ((q0 + q + 1) * step > a.total) break;
            const int nsub = SUB ? (int)a.nsub.d : 1;
            double acc = out(recs[c][o + q * step + nsub - 1]);
            for (int j = 1; j < (int)a.factor.d; ++j) acc += out(recs[c][o + q * step + j * nsub + nsub - 1]);
            a.rows32[(q0 + q) * a.rows.ld + io] = f32_to_file((float)(a.factor.d > 1 ? acc / (double)a.factor.d : acc), a.out32_sel);
        }
    } else {
        for (int r = tid / kRecOutCols; r < kRecRows; r += kRecOutThreads / kRecOutCols) {
            const int64_t tick = tick0 + r;
            if (tick >= a.total) break;
            if (SUB) {
                uint32_t sub;
                const uint32_t row = a.nsub.div((uint32_t)tick, sub);
                if (sub + 1 == a.nsub.d) a.rows.row(row)[io] = recs[c][o + r];
            } else {
                a.rows.row(tick)[io] = out(recs[c][o + r]);
            }
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

}  // namespace
