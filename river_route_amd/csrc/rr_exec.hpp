// rr_exec.hpp -- the plan object and the executor: schedule choice, record ring, streaming session, host PCIe pipeline,
// the route cores behind the C ABI.  Part of the one translation unit rr_engine.hip builds.
#pragma once

// ------------------------------------------------------------------------------------------------
// plan object
// ------------------------------------------------------------------------------------------------

enum class Mode { Rapid, Muskingum, Unit };

// Where the (time, reach) rows in params order come from / go to.
struct Rows {
    const double *dev_in = nullptr;   // device array, rows_in rows
    const float *dev_in32 = nullptr;  // instead of dev_in: float32 lateral rows (RapidMuskingum; exact in float64)
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
    // direct row path (rr_kernels_direct.hpp): K rows per task, the skeleton behind it on records
    bool rows_direct = false;
    int64_t n_tasks = 0;          // direct launches: rows [d K, (d + 1) K) in launch d
    int64_t d_done = 0;           // direct launches made
    int64_t KS = 1;               // record chunks per task of the skeleton's launches (KS divides KC; shorter where the part feeds another GPU)
    bool export_lanes = false, export_skel = false;    // a boundary export is routed by a lane of a direct tile (final as soon as its rows are routed) / by the skeleton
    DirectArgs da{};
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
    int copy_threads = kCopyThreads;      // per direction: PCIe carries 26 GB/s each way whatever the count (profiles/r03_host_path_threads.txt)
    int64_t chunk_mib = 512;              // staging chunk: long enough for the DMA engines to reach their rate, short enough to pipeline
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
    double *d_a2 = nullptr, *d_c1own = nullptr, *d_z = nullptr;   // UnitMuskingum with general edge data (rr_plan_set_unit_weights): streaming kernel only
    bool unit_general = false;
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
    int wave_threads = 512;
    int64_t wave_K = 0;          // ticks per task (multiple of 16); 0 = chosen per call
    int64_t next_KC = 1, next_KS = 1, next_chunks = 0;   // prepare_call: task length(s) and record ring of the call about to start
    int64_t kc_cap = int64_t{1} << 20;      // longest task (record chunks) the device had room for; 0: no record ring fits, the plan streams
    TileMeta *d_tmeta = nullptr;     // per tile (rr_kernels_tile.hpp)
    int4 *d_pmeta = nullptr;         // per position {lag | flags, first upstream position, xpos, upstream counts}
    int32_t *d_tperm = nullptr, *d_tinv = nullptr;
    int32_t *d_inner_idx = nullptr;
    bool export_inside = false;      // an export reach that another tile mirrors (it has a downstream reach in this plan): streaming kernel only
    double *d_coef = nullptr;        // per position {c1row, c2, c3}
    double *d_coef_unit = nullptr;   // the same with zeros at the positions without upstream positions (UnitMuskingum's short tick)
    std::vector<double> h_coef;      // the same on the host: boundary ghosts of a partitioned network get zeros (upload_tile_coef)
    int32_t n_wide_tiles = 0;        // tiles with a reach of more than three upstream reaches: general kernel beside the LEAN one
    double *d_sq = nullptr, *d_ss = nullptr, *d_si = nullptr, *d_sqch = nullptr;
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
    // direct row path: column-range tiles where the params order numbers small subtrees contiguously (rr_plan.hpp: DirectPlan)
    rr::DirectPlan dp;
    bool direct_enabled = true, direct_now = false;      // RR_DIRECT=0 (tests): records for every call
    bool uh_rows_ok = true;      // false once the device refused the convolved rows of rr_unit_route_uh*_dev on the direct row path: records from then on
    int direct_window = 1;               // rows of the LDS window: largest span + 1 of the plan's tiles
    DirectTile *d_dtiles = nullptr;
    int4 *d_dlane = nullptr;
    int32_t *d_dsend_ptr = nullptr, *d_dsend_lane = nullptr;
    double *d_dcoef = nullptr, *d_dq = nullptr;
    TileMeta *d_ktmeta = nullptr;        // the skeleton's tiles (TileArgs of its k_tile launches)
    int4 *d_kpmeta = nullptr;
    int32_t *d_kperm = nullptr, *d_kholecol = nullptr;
    double *d_kcoef = nullptr, *d_ksq = nullptr, *d_kss = nullptr, *d_ksi = nullptr, *d_ksqch = nullptr;
    int2 *d_kholemeta = nullptr;         // per hole {position in the skeleton, lag}: the out-pass that patches the holes
    int2 *d_kghostmeta = nullptr;        // per boundary ghost {position of the skeleton's ghost that mirrors it, lag}: the in-pass of the ghost series
    std::vector<double> h_dcoef;         // {c1row, c2, c3, c4dt} per column as rr_plan_set_coeffs got them
    int32_t direct_block = 0;            // tile capacity the plan was built with (rr_plan_set_boundary lays the direct plan out again)
    int64_t n_kholes = 0;
    int32_t n_kwide = 0;
    bool in32_big_endian = false, out32_big_endian = false;      // rr_plan_set_row_format: float32 rows as a big-endian file stores them
    bool lean_enabled = true;           // RR_TILE_LEAN=0 (tests): the general tick for every call
    bool uh_pairs = true;               // the fused convolution takes two record batches per launch where it can (RR_UH_PAIRS=0: tests)
    bool perm_ready = false;            // the streaming kernel's tiled permutations are on the device

    // profile of the last route call: the routing kernel (ev, rr_plan_profile) and, sampled the same way, the kernels around it
    // (aux: 0 in-pass, 1 out-pass, 2 the skeleton's k_tile launches of the direct row path, 3 its out-pass over the holes)
    static constexpr int kAuxKinds = 4, kAuxSamples = 256;
    std::vector<hipEvent_t> aux_ev;      // 2 per sample, made by rr_plan_reserve
    std::vector<int> aux_kind;
    int64_t aux_launches[kAuxKinds] = {0, 0, 0, 0};
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

// Tile and position constants of the time-tiled kernel, packed (TileArgs); boundary reaches of a partitioned network change
// the lag flags, the xpos word and the tile flags (rr_plan_set_boundary).
int upload_tile_meta(rr_plan *P, const std::vector<int32_t> &lag, const std::vector<int32_t> &xpos, const std::vector<int32_t> &flags)
{
    const rr::TilePlan &TP = P->tp;
    std::vector<TileMeta> tm((size_t)TP.n_tiles);
    for (int32_t t = 0; t < TP.n_tiles; ++t)
        tm[t] = TileMeta{TP.tile_ptr[t], TP.tile_ptr[t + 1], TP.tile_level[t], TP.tile_lag_lo[t], TP.tile_lag_hi[t], flags[t], 0, 0};
    std::vector<int4> pm((size_t)TP.np);
    for (int64_t p = 0; p < TP.np; ++p) pm[p] = make_int4(lag[p], TP.cfirst[p], xpos[p], (int32_t)TP.ccnt[p]);
    int rc = RR_OK;
    if (!P->d_tmeta) rc = dev_alloc(&P->d_tmeta, TP.n_tiles);
    if (!rc && !P->d_pmeta) rc = dev_alloc(&P->d_pmeta, TP.np);
    if (!rc) rc = dev_upload(P->d_tmeta, tm);
    if (!rc) rc = dev_upload(P->d_pmeta, pm);
    return rc;
}

// Coefficients in tile order.  A ghost computes nothing: a tile ghost (mirror of a reach another tile owns) and a boundary
// ghost of a partitioned network (a reach another GPU owns) both publish what arrives in their record, which is what the
// short tick's three multiply-adds do with zero coefficients.
int upload_tile_coef(rr_plan *P)
{
    if (P->h_coef.empty()) return RR_OK;
    std::vector<double> coef(P->h_coef);
    for (int32_t i : P->ghost_reach) { const int64_t p = P->tp.inv[i]; coef[3 * p] = coef[3 * p + 1] = coef[3 * p + 2] = 0.0; }
    int rc = dev_upload(P->d_coef, coef);
    if (rc) return rc;
    // UnitMuskingum's short tick: a headwater publishes its lateral inflow (_numba_kernels.py:122-123), which zero coefficients do too
    for (int64_t p = 0; p < P->tp.np; ++p)
        if ((P->tp.ccnt[p] & 0xFFFFu) == 0) coef[3 * p] = coef[3 * p + 1] = coef[3 * p + 2] = 0.0;
    return dev_upload(P->d_coef_unit, coef);
}

typedef void (*direct_kernel_t)(const DirectArgs);
direct_kernel_t direct_kernel(bool in32, bool out32, bool sub = false, int unit = 0)
{
    if (unit == 1)      // UnitMuskingum: float64 rows
        return sub ? (out32 ? (direct_kernel_t)k_direct<kDirectAhead, false, true, true, 1> : (direct_kernel_t)k_direct<kDirectAhead, false, false, true, 1>)
                   : (out32 ? (direct_kernel_t)k_direct<kDirectAhead, false, true, false, 1> : (direct_kernel_t)k_direct<kDirectAhead, false, false, false, 1>);
#define RR_DK(I_, O_) (sub ? (direct_kernel_t)k_direct<kDirectAhead, I_, O_, true> : (direct_kernel_t)k_direct<kDirectAhead, I_, O_, false>)
    return in32 ? (out32 ? RR_DK(true, true) : RR_DK(true, false)) : (out32 ? RR_DK(false, true) : RR_DK(false, false));
#undef RR_DK
}

// The direct row path's device arrays (rr::DirectPlan): per-column constants, the skeleton's tile arrays, the columns of the holes' out-pass.
// Called by rr_plan_create and again by rr_plan_set_boundary, which lays the plan out anew around the boundary reaches.
int upload_direct_plan(rr_plan *P)
{
    void *old[] = {P->d_dtiles, P->d_dlane, P->d_dsend_ptr, P->d_dsend_lane, P->d_dcoef, P->d_dq, P->d_ktmeta, P->d_kpmeta, P->d_kperm, P->d_kholecol, P->d_kcoef,
                   P->d_ksq, P->d_kss, P->d_ksi, P->d_ksqch, P->d_kholemeta, P->d_kghostmeta};
    for (void *p : old) if (p) (void)hipFree(p);
    P->d_dtiles = nullptr; P->d_dlane = nullptr; P->d_dsend_ptr = P->d_dsend_lane = nullptr; P->d_dcoef = P->d_dq = nullptr; P->d_ktmeta = nullptr; P->d_kpmeta = nullptr;
    P->d_kperm = P->d_kholecol = nullptr; P->d_kcoef = P->d_ksq = P->d_kss = P->d_ksi = P->d_ksqch = nullptr; P->d_kholemeta = P->d_kghostmeta = nullptr;
    P->n_kholes = 0; P->n_kwide = 0;
    P->direct_window = 3;
    for (int32_t sp : P->dp.tile_span) P->direct_window = std::max(P->direct_window, sp + 3);
    if (!P->dp.ok || P->device < 0) return RR_OK;
    const rr::HostPlan &H = P->h;
    const rr::DirectPlan &D = P->dp;
    const rr::TilePlan &K = D.skel;
    const int64_t n = H.n;
    for (int v = 0; v < 12; ++v)      // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (v = 8 ... 11: UnitMuskingum, float64 rows in)
        if (hipFuncSetAttribute((const void *)direct_kernel(v < 8 && (v & 1) != 0, (v & 2) != 0, v < 8 ? (v & 4) != 0 : (v & 1) != 0, v >= 8 ? 1 : 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)direct_lds_bytes(kDirectMaxWindow)) != hipSuccess) {
            (void)hipGetLastError();
            P->direct_enabled = false;
        }
    std::vector<DirectTile> dt((size_t)D.n_tiles);
    for (int32_t t = 0; t < D.n_tiles; ++t) dt[t] = DirectTile{D.tile_c0[t], D.tile_nc[t], D.tile_lag_lo[t], D.tile_span[t]};
    std::vector<int4> dl((size_t)n);
    for (int64_t i = 0; i < n; ++i) dl[i] = make_int4(D.delay[i], D.up3[i], D.xinfo[i], H.lag[H.inv[i]]);
    int rc = dev_alloc(&P->d_dtiles, D.n_tiles);
    if (!rc) rc = dev_upload(P->d_dtiles, dt);
    if (!rc) rc = dev_alloc(&P->d_dlane, n);
    if (!rc) rc = dev_upload(P->d_dlane, dl);
    if (!rc) rc = dev_alloc(&P->d_dsend_ptr, D.n_tiles + 1);
    if (!rc) rc = dev_upload(P->d_dsend_ptr, D.send_ptr);
    if (!rc) rc = dev_alloc(&P->d_dsend_lane, (int64_t)D.send_lane.size());
    if (!rc) rc = dev_upload(P->d_dsend_lane, D.send_lane);
    if (!rc) rc = dev_alloc(&P->d_dcoef, 4 * n);
    if (!rc) rc = dev_alloc(&P->d_dq, n);
    std::vector<TileMeta> tm((size_t)K.n_tiles);
    std::vector<int32_t> klag(K.lag), kxpos(K.xpos), kflags(K.tile_flags);
    // boundary exports among the skeleton's reaches: flagged as in the plan's own tiles (rr_plan_set_boundary); the slot in the export
    // series travels in xpos[] (an export is an outlet of its part: no ghost of another tile mirrors it -- checked by build_direct_plan)
    for (size_t e = 0; e < P->export_reach.size(); ++e) {
        const int32_t i = P->export_reach[e];
        if (!D.big[i]) continue;
        const int32_t p = K.inv[i];
        klag[p] |= kExportBit; kxpos[p] = (int32_t)e; kflags[K.tile_of[p]] |= kTileExports;
    }
    for (int32_t t = 0; t < K.n_tiles; ++t) {
        tm[t] = TileMeta{K.tile_ptr[t], K.tile_ptr[t + 1], K.tile_level[t], K.tile_lag_lo[t], K.tile_lag_hi[t], kflags[t], 0, 0};
        P->n_kwide += (kflags[t] & kTileWide) ? 1 : 0;
    }
    std::vector<int4> pm((size_t)K.np);
    for (int64_t p = 0; p < K.np; ++p) pm[p] = make_int4(klag[p], K.cfirst[p], kxpos[p], (int32_t)K.ccnt[p]);
    std::vector<int2> hm, gm;
    std::vector<int32_t> hc;
    for (int64_t i = 0; i < n; ++i)
        if (D.big[i]) { hm.push_back(make_int2(K.inv[i], K.lag[K.inv[i]] & kLagMask)); hc.push_back((int32_t)i); }
    for (int32_t i : P->ghost_reach) { const int32_t g = K.ext_ghost[i]; gm.push_back(make_int2(g, K.lag[g] & kLagMask)); }
    P->n_kholes = (int64_t)hm.size();
    if (!rc) rc = dev_alloc(&P->d_ktmeta, K.n_tiles);
    if (!rc) rc = dev_upload(P->d_ktmeta, tm);
    if (!rc) rc = dev_alloc(&P->d_kpmeta, K.np);
    if (!rc) rc = dev_upload(P->d_kpmeta, pm);
    if (!rc) rc = dev_alloc(&P->d_kperm, K.np);
    if (!rc) rc = dev_upload(P->d_kperm, K.perm);
    if (!rc) rc = dev_alloc(&P->d_kcoef, 3 * K.np);
    if (!rc) rc = dev_alloc(&P->d_ksq, K.np);
    if (!rc) rc = dev_alloc(&P->d_kss, K.np);
    if (!rc) rc = dev_alloc(&P->d_ksi, K.np);
    if (!rc) rc = dev_alloc(&P->d_ksqch, K.np);
    if (!rc) rc = dev_alloc(&P->d_kholemeta, P->n_kholes);
    if (!rc) rc = dev_upload(P->d_kholemeta, hm);
    if (!rc) rc = dev_alloc(&P->d_kholecol, P->n_kholes);
    if (!rc) rc = dev_upload(P->d_kholecol, hc);
    if (!rc) rc = dev_alloc(&P->d_kghostmeta, (int64_t)gm.size());
    if (!rc) rc = dev_upload(P->d_kghostmeta, gm);
    return rc;
}

// Coefficients of the direct row path from the host copy rr_plan_set_coeffs keeps: per column for the lanes, per position for the skeleton
// (a ghost position computes nothing: zeros, as in upload_tile_coef).
int upload_direct_coef(rr_plan *P)
{
    if (!P->dp.ok || P->device < 0 || P->h_dcoef.empty()) return RR_OK;
    int rc = dev_upload(P->d_dcoef, P->h_dcoef);
    const rr::TilePlan &K = P->dp.skel;
    std::vector<double> kc(3 * (size_t)K.np, 0.0);
    for (int64_t p = 0; p < K.np; ++p) {
        if (K.lag[p] & kTileGhostBit) continue;
        const int64_t i = K.perm[p];
        kc[3 * p] = P->h_dcoef[4 * i]; kc[3 * p + 1] = P->h_dcoef[4 * i + 1]; kc[3 * p + 2] = P->h_dcoef[4 * i + 2];
    }
    if (!rc) rc = dev_upload(P->d_kcoef, kc);
    return rc;
}

// The two-phase tiled permutation params order <-> lag order of the streaming kernel (k_perm_a / k_perm_b).  Built when a
// call first streams: the time-tiled kernel, which takes almost every call, has its own record passes.
int upload_tiled_permutations(rr_plan *P)
{
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n;
    const int32_t *pis[2] = {H.perm.data(), H.inv.data()};
    int rc = RR_OK;
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
    if (rc) {      // all or nothing: a later call must not find half of the tables
        for (int w = 0; w < 2; ++w) {
            if (P->d_slot_a[w]) (void)hipFree(P->d_slot_a[w]);
            if (P->d_slot_b[w]) (void)hipFree(P->d_slot_b[w]);
            if (P->d_m_index[w]) (void)hipFree(P->d_m_index[w]);
            P->d_slot_a[w] = P->d_slot_b[w] = nullptr; P->d_m_index[w] = nullptr;
        }
        return rc;
    }
    P->perm_ready = true;
    return RR_OK;
}

// Every fourth launch of a kind is bracketed by HIP events while the plan samples (rr_plan_set_options): aux_begin returns the
// sample's number or -1.
int aux_begin(rr_plan *P, int kind, hipStream_t stream)
{
    const int64_t seq = P->aux_launches[kind]++;
    if (P->sample_every < kSampleGroup || (seq & 3) != 0 || 2 * (P->aux_kind.size() + 1) > P->aux_ev.size()) return -1;
    if (hipEventRecord(P->aux_ev[2 * P->aux_kind.size()], stream) != hipSuccess) return -1;
    P->aux_kind.push_back(kind);
    return (int)P->aux_kind.size() - 1;
}
void aux_end(rr_plan *P, int sample, hipStream_t stream)
{
    if (sample >= 0) (void)hipEventRecord(P->aux_ev[2 * sample + 1], stream);
}

// ---- session -------------------------------------------------------------------------------------

// Record chunks per task.  A longer task amortises the load of the tile's state and of its first half chunk, which
// nothing overlaps; every tile level adds one task of skew to the pipeline and to the record ring.  Tasks of 128 and 256
// ticks are for long calls on networks whose ring then stays under a fifth of the card (choose_schedule): they are worth
// 11 % at 100k reaches, 7 % at 250k, 3 % at 500k (profiles/microbench/k_sweep_small.sh) and 2-3 % at 1M, where 128 ticks
// make the ring 50 GB (366.5 / 367.3 ms per year against 374.3 / 382.3 at 64 ticks, profiles/r03_k_sweep.txt).
int64_t pick_KC(const rr_plan *P, int64_t total_ticks)
{
    if (P->wave_K > 0) return std::max<int64_t>(1, std::min<int64_t>(P->wave_K, 256) / kRec);      // (RR_WAVE_K above 256 -- the longest task the schedule itself picks here -- is for the direct row path's lane tasks: the in-pass runs at most four batches of 128 tick-rows ahead of the out-pass)
    return total_ticks >= 32768 ? 16 : (total_ticks >= 16384 ? 8 : (total_ticks >= 4096 ? 4 : (total_ticks >= 512 ? 2 : 1)));
}

// Which routing kernel a call uses.  The time-tiled schedule needs one upstream weight per reach, a network that tiles
// (rr::TilePlan) and room for its record ring; its fill and drain cost (levels x K) ticks more than the streaming kernel's, a
// few launches, so only calls of a handful of sub-steps stream.  RR_WAVE=1 forces it where it applies, RR_WAVE=0 forbids it.
//
// Records are indexed by tick = tick-row + lag, modulo the ring, per position: a position's slots never hold another
// position's data, so what the ring must cover is one position's tick-rows in flight.  Rows enter for all columns at once
// (ahead of the level-0 tiles) and leave for all columns at once (after the last level has passed their tick + depth), so
// every position keeps depth + levels * K tick-rows plus the batching of the two permutation passes; that its window sits
// lag ticks later than a headwater's does not widen it.  The ring may take five eighths of the card; a deep network that
// does not fit gets shorter tasks, then the streaming kernel.
//
// The choice is a pure function of the plan, the call's shape and kc_cap (lowered only when the device refuses an
// allocation), so rr_plan_reserve and the call it prepares for agree on it.
struct Schedule {
    bool direct = false;              // direct row path: KC = rows per task / 16, chunks / ring = the skeleton's record ring
    int64_t KS = 1;                   // direct row path: record chunks per task of the skeleton's launches
    bool tiled = false;
    int64_t KC = 1, chunks = 0;       // time-tiled: record chunks per task, chunks of the record ring
    int64_t ring = 0;                 // doubles of P->d_ring: record ring, or the work rows of the streaming kernel
    int64_t mrows = 0, stage = 0;     // streaming kernel: doubles of the permutation's intermediate rows / of the host staging rows
};

// Rows per task of the direct path: a task costs its tile `span` ticks of fill and drain (lanes start one after the other: 10 % at
// 512 rows, 5 % at 1,024).  The skeleton's tasks are cut separately (Schedule::KS, direct_KS below): every tile level of the skeleton costs
// one of ITS tasks of pipeline and of record ring, so long lane tasks over short skeleton tasks have both -- the year at 1M reaches
// 196.5 ms with 512 / 512, 186.4 with 1,024 / 512, 185.1 with 2,048 / 512; a 3,504-row call 24.4 ms with 256 / 256, 21.6 with 512 / 128; a
// 744-row call 7.4 ms with 128 / 128, 6.6 with 256 / 128 (profiles/r05_direct_ks_ab.txt; with the skeleton's tasks as long as the lanes',
// 1,024 rows lost to 512: 213 against 207 ms, profiles/r04_direct_k_and_fill.txt).
int64_t pick_direct_K(const rr_plan *P, int64_t T)
{
    if (P->wave_K > 0) return P->wave_K;
    return T >= 16384 ? 1024 : (T >= 2048 ? 512 : (T >= 512 ? 256 : (T >= 128 ? 128 : 64)));
}
int64_t direct_KS(int64_t task_chunks, int64_t total_ticks)      // record chunks per skeleton task: 512 ticks in long calls, 128 in short ones; divides the lanes' task
{
    int64_t ks = std::min<int64_t>(task_chunks, (total_ticks >= 16384 ? 512 : 128) / kRec);
    while (task_chunks % ks) --ks;
    return ks;
}

// ring_in / ring_out (streaming calls, rr_stream_begin): rows of the caller's cyclic lateral / discharge arrays where those are shorter
// than the call (0: they hold every row).  A caller may refill such a ring between two rr_stream_advance calls, so a direct task must
// not span more rows than the ring holds: K is capped (a ring of fewer than 32 rows keeps to records, which take rows in batches of 128
// tick-rows the caller announces one by one).
// out32 (float32 output fused in, rr_*_f32*_dev): the direct task is a multiple of 128 rows, so that every output row -- the mean of
// `factor` routed rows, factor divides 128 -- lies inside one task.
// uh (rr_unit_route_uh*_dev: runoff depths + unit-hydrograph kernel): on the direct row path the convolution runs first, as a pass of its own
// into T rows of work memory (mrows) that the lanes then read -- the reference's own two steps (UnitMuskingum.py:72-98) -- where those rows
// fit a third of the card; otherwise the call keeps to records with the convolution fused into the in-pass.
Schedule choose_schedule(const rr_plan *P, Mode mode, int64_t T, int64_t nsub, bool force_streaming, bool host_io, bool plain_rows = false,
                         int64_t ring_in = 0, int64_t ring_out = 0, bool out32 = false, bool uh = false)
{
    Schedule sch;
    const int64_t total = T * nsub, dmax = P->h.depth - 1, n = P->h.n;
    // The direct row path: RapidMuskingum, one sub-step per row, float64 rows in device arrays, one weight per reach -- the
    // headline's call -- on a params order that numbers small subtrees contiguously (boundary reaches of a partitioned network
    // included: rr_plan_set_boundary lays the direct plan out around them).
    // Sub-steps (up to kDirectMaxSub a row) and channel-only routing take it too; with sub-steps only without boundary ghosts.  UnitMuskingum: below.
    // UnitMuskingum (float64 rows of convolved lateral inflow, no boundary reaches) takes it as well.
    const bool unit_direct = mode == Mode::Unit && nsub <= kDirectMaxSub && P->n_ghost == 0 && P->n_export == 0 && !P->unit_general && P->tp.ok &&
                             (!uh || (P->uh_rows_ok && (P->dev_total_bytes == 0 || T * n * 8 <= (int64_t)(P->dev_total_bytes / 3))));
    if (plain_rows && P->direct_enabled && P->dp.ok && (mode == Mode::Rapid || mode == Mode::Muskingum || unit_direct) && nsub <= kDirectMaxSub && (nsub == 1 || P->n_ghost == 0) &&
        P->weights_uniform && !force_streaming && !host_io && P->wave_enabled && total >= 8 && n < (int64_t{1} << 29)) {
        int64_t K = pick_direct_K(P, T);
        const int64_t levels = P->dp.skel.n_levels, np = P->dp.skel.np;
        if (ring_in > 0 && ring_in < T) K = std::min(K, ring_in / kRec * kRec);
        if (ring_out > 0 && ring_out < T) K = std::min(K, ring_out / kRec * kRec);
        if (out32) K = std::max<int64_t>(kRecRows, K / kRecRows * kRecRows);
        sch.direct = true; sch.KC = K / kRec;
        // The skeleton's own tasks are shorter than the lanes' (several of the skeleton's launches per direct launch; a direct task is K rows =
        // K nsub ticks): see pick_direct_K -- and 64 ticks in a part that feeds another GPU, whose boundary series every level delays by one
        // task (as kc_long below).  KS divides KC nsub.
        sch.KS = direct_KS(sch.KC * nsub, total);
        if (P->n_export > 0 && P->wave_K <= 0) for (sch.KS = std::min<int64_t>(sch.KC * nsub, 4); (sch.KC * nsub) % sch.KS; --sch.KS) {}
        if (np > 0) {      // a record lives from the launch that forwards its first row to the out-pass behind the skeleton's last level
            // a ring cut to the call's length: with boundary ghosts it has to hold their in-pass's batches whole -- k_rec_in writes nine records per position and
            // batch, zeros past the call's end, and in a ring shorter than that a batch wrapped onto its own first records (a 40- or 130-row call of a shallow part:
            // wrong boundary inflow; a 200-row call: the last batch waited for its own slots for ever -- both found by profiles/microbench/parts_fuzz.py)
            int64_t whole_call = (total + dmax) / kRec + 2;
            if (P->n_ghost > 0) whole_call = std::max<int64_t>(whole_call, kRecBatch * ((total + 14) / kRecRows + 1) + (dmax >> 4) + 2);
            sch.chunks = std::min<int64_t>((levels * sch.KS * kRec + 2 * K * nsub + 2 * dmax + kRecRows + 2 * kRec) / kRec + 2, whole_call);
            sch.ring = sch.chunks * kRec * np;
        }
        if (uh) sch.mrows = T * n;      // the convolved rows
        if (K >= 2 * kRec && np < (int64_t{1} << 25) && (P->dev_total_bytes == 0 || sch.ring * 8 <= (int64_t)(P->dev_total_bytes / 2))) return sch;
        sch = Schedule();
    }
    bool ok = P->wave_enabled && P->tp.ok && P->weights_uniform && n > 0 && !force_streaming && P->tp.np < (int64_t{1} << 25) &&
              !P->export_inside && P->kc_cap >= 1 && !(mode == Mode::Unit && P->unit_general);
    if (ok && !P->wave_forced) ok = total >= 32;
    if (ok) {
        const int64_t np = P->tp.np, levels = P->tp.n_levels;
        const int64_t all_chunks = kRecBatch * ((total + 14) / kRecRows + 2) + (dmax >> 4) + 2;
        const int64_t slack = 4;      // room for the in-pass to run ahead of the routing: four batches of 128 tick-rows
        ok = false;
        // a part of a cut network that feeds another GPU keeps to 64 ticks: every tile level delays its boundary series by one
        // task, and the GPU downstream waits for it (10M reaches on 8 GPUs: whole job 7.2 against 7.1 x 10^11 reach-steps/s in the
        // time-weighted simulation, profiles/r03_pipeline_sim_time.txt)
        const int64_t kc_long = P->n_export > 0 && P->wave_K <= 0 ? 4 : (int64_t{1} << 20);
        for (int64_t KC = std::min(std::min(pick_KC(P, total + dmax), P->kc_cap), kc_long); KC >= 1; KC /= 2) {
            const int64_t chunks = std::min<int64_t>(all_chunks, (dmax + levels * KC * kRec) / kRec + slack * kRecBatch);
            const int64_t bytes = chunks * kRec * np * (int64_t)sizeof(double);
            // five eighths of the card; the shortest tasks may take thirteen sixteenths: the streaming kernel, the only alternative,
            // keeps depth x n work rows itself (1M reaches 24k deep: 204 GB of records at K = 16 against 192 GB of rows)
            if (P->dev_total_bytes > 0 && bytes > (int64_t)(P->dev_total_bytes / 16 * (KC == 1 ? 13 : 10))) continue;
            if (KC > 4 && P->wave_K <= 0 && P->dev_total_bytes > 0 && bytes > (int64_t)(P->dev_total_bytes / 5)) continue;      // long tasks only while the ring stays under a fifth of the card
            sch.tiled = true; sch.KC = KC; sch.chunks = chunks; sch.ring = chunks * kRec * np;
            ok = true;
            break;
        }
    }
    if (!ok) {      // streaming kernel: lateral rows come in, discharge rows overwrite them in place and stay until the outlet-most reaches have passed them
        const int64_t C = std::max<int64_t>(1, P->chunk_rows), lag_rows = (dmax + nsub - 1) / std::max<int64_t>(1, nsub);
        const bool direct = P->h.identity && !host_io;
        sch.ring = direct ? 0 : std::min<int64_t>(T, lag_rows + 2 * C + 2) * n;
        sch.mrows = direct ? 0 : C * n;
        sch.stage = host_io ? C * n : 0;
    }
    return sch;
}

int host_pipe_prepare(rr_plan *P);

// Sizes and allocates what a call of this shape works in.  The only place on a route call's path that allocates: the
// host-pointer entry points come here by themselves, the *_dev ones expect rr_plan_reserve to have been here.
int reserve_core(rr_plan *P, Mode mode, int64_t T, int64_t nsub, bool force_streaming, bool host_io, Schedule *out, bool plain_rows = false, bool out32 = false, bool uh = false)
{
    if (P->h.n == 0 || T <= 0) { if (out) *out = Schedule(); return RR_OK; }
    Schedule sch;
    for (;;) {
        sch = choose_schedule(P, mode, T, nsub, force_streaming, host_io, plain_rows, 0, 0, out32, uh);
        if (sch.direct && uh && ensure_cap(&P->d_mrows, &P->mrows_cap, sch.mrows) != RR_OK) {      // no room for the convolved rows: the fused form on records
            (void)hipGetLastError();
            P->uh_rows_ok = false;
            continue;
        }
        if (ensure_cap(&P->d_ring, &P->ring_cap, sch.ring) == RR_OK) break;
        (void)hipGetLastError();
        if (sch.direct) return fail(RR_E_ALLOC, "route: the skeleton's record ring does not fit on the device");
        if (!sch.tiled) return fail(RR_E_ALLOC, "route: the work rows of the streaming kernel do not fit on the device");
        P->kc_cap = sch.KC / 2;      // shorter tasks, a smaller ring; 0: this plan streams
    }
    int rc = ensure_cap(&P->d_mrows, &P->mrows_cap, sch.mrows);
    if (!rc) rc = ensure_cap(&P->d_stage, &P->stage_cap, sch.stage);
    if (!rc && !sch.tiled && !sch.direct && !P->perm_ready) rc = upload_tiled_permutations(P);
    if (!rc && sch.tiled && host_io) rc = host_pipe_prepare(P);
    if (rc) return rc;
    // events: first / last of a call, the sampled launches (rr_plan_set_options), the second stream's
    if (!P->ev_first) { HIPCHK(hipEventCreate(&P->ev_first)); HIPCHK(hipEventCreate(&P->ev_last)); }
    const int64_t K = sch.KC * kRec, total_ticks = T * nsub + P->h.depth - 1;
    size_t samples = P->sample_every >= kSampleGroup ? (size_t)std::min<int64_t>(4096, total_ticks / P->sample_every + 1) : 0;
    if (sch.tiled && samples > 0) samples = (size_t)std::min<int64_t>(4096, ((total_ticks + K - 1) / K + P->tp.n_levels) / 4 + 1);
    if (sch.direct && samples > 0) samples = (size_t)std::min<int64_t>(4096, (T + K - 1) / K + 1);      // every direct launch
    while (P->ev.size() < 2 * samples) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); P->ev.push_back(e); }
    if (samples > 0) while (P->aux_ev.size() < 2 * rr_plan::kAuxSamples) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); P->aux_ev.push_back(e); }
    if (out) *out = sch;
    return RR_OK;
}

// The schedule of the call about to start.  strict (the *_dev entry points, which only enqueue): everything must have been
// reserved; otherwise it is reserved here.
int prepare_call(rr_plan *P, Mode mode, int64_t T, int64_t nsub, bool force_streaming, bool host_io, bool strict, bool plain_rows = false,
                 int64_t ring_in = 0, int64_t ring_out = 0, bool out32 = false, bool uh = false)
{
    Schedule sch;
    if (!strict) {
        int rc = reserve_core(P, mode, T, nsub, force_streaming, host_io, &sch, plain_rows, out32, uh);
        if (rc) return rc;
    } else {
        sch = choose_schedule(P, mode, T, nsub, force_streaming, host_io, plain_rows, ring_in, ring_out, out32, uh);
        const size_t samples = P->sample_every >= kSampleGroup ? 1 : 0;
        if (P->h.n > 0 && T > 0 && (sch.ring > P->ring_cap || sch.mrows > P->mrows_cap || sch.stage > P->stage_cap || !P->ev_first || P->ev.size() < 2 * samples ||
                                    (!sch.tiled && !sch.direct && !P->perm_ready && sch.mrows > 0)))
            return fail(RR_E_STATE, "this call needs " + std::to_string((sch.ring + sch.mrows + sch.stage) * 8) + " bytes of work memory on the device (" +
                                        std::to_string((P->ring_cap + P->mrows_cap + P->stage_cap) * 8) + " reserved): call rr_plan_reserve(plan, mode, " +
                                        std::to_string(T) + ", " + std::to_string(nsub) + ", ...) first; the *_dev entry points only enqueue work");
    }
    P->wave_now = sch.tiled; P->direct_now = sch.direct; P->next_KC = sch.KC; P->next_KS = sch.KS; P->next_chunks = sch.chunks;
    return RR_OK;
}

bool use_wave(const rr_plan *P, Mode) { return P->wave_now; }

// The record passes run on the caller's stream between the routing launches.  (On a second stream beside them they slow the
// routing launch down by what they gain: 395 against 388 ms per year, slower on small networks too --
// profiles/r03_nt_and_rec_stream_ab.txt, r03_rec_stream_small_networks.txt.)
hipStream_t rec_stream(const rr_plan *P) { return P->ses.stream; }

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
    S.rows_direct = P->direct_now;
    S.wave = use_wave(P, mode) && !S.rows_direct;
    S.direct = H.identity && !host_io && !S.wave && !S.rows_direct;   // engine order == params order: the streaming kernel reads the caller's arrays
    const int64_t C = std::max<int64_t>(1, P->chunk_rows);

    P->prof_launches = P->prof_samples = P->prof_brackets = 0;
    P->aux_kind.clear();
    for (int k = 0; k < rr_plan::kAuxKinds; ++k) P->aux_launches[k] = 0;
    P->prof_reach_steps = n * S.total;
    P->ev_reaches.clear();
    P->last_stream = stream;
    S.open = true;
    if (n == 0 || S.total == 0) return RR_OK;
    if (P->n_ghost > 0 && !ghost_series) { S.open = false; return fail(RR_E_INVALID, "plan has ghost reaches but no ghost series was given"); }
    if (P->n_export > 0 && !export_series) { S.open = false; return fail(RR_E_INVALID, "plan has export reaches but no export series was given"); }
    if (io.dev_out32 && !S.wave && !S.rows_direct) { S.open = false; return fail(RR_E_UNSUPPORTED, "float32 output needs the time-tiled kernel or the direct row path"); }

    if (S.wave || S.rows_direct) { S.KC = P->next_KC; S.rec_chunks = P->next_chunks; }     // ring sized by choose_schedule, allocated by rr_plan_reserve
    if (S.rows_direct) {
        if (io.uh_kernel || io.runoff || (S.has_in && !io.dev_in && !io.dev_in32) || (!io.dev_out && !io.dev_out32) || (mode == Mode::Unit && (io.dev_in32 || P->n_ghost > 0 || P->n_export > 0)) || nsub > kDirectMaxSub ||
            (io.dev_out32 && (io.out_factor < 1 || (S.KC * kRec) % io.out_factor != 0 || T % io.out_factor != 0))) {
            S.open = false;
            return fail(RR_E_STATE, "route: the direct row path was chosen for a call it does not take");      // (choose_schedule's plain_rows / out32)
        }
        const rr::TilePlan &TP = P->dp.skel;
        const int64_t K = S.KC * kRec;
        S.KS = std::max<int64_t>(1, P->next_KS);
        S.n_tasks = (S.T + K - 1) / K;
        S.n_macro = (S.total_ticks + S.KS * kRec - 1) / (S.KS * kRec);      // of the skeleton's launches
        S.n_diags = TP.n_tiles > 0 ? S.n_macro + TP.n_levels - 1 : 0;
        S.n_in_batches = (S.total + 14) / kRecRows + 1;      // of the boundary series (if any)
        S.n_out_batches = TP.np > 0 ? (S.total + kRecRows - 1) / kRecRows : 0;
        S.export_skew = 0; S.export_lanes = S.export_skel = false;
        for (int32_t i : P->export_reach) {
            if (!P->dp.big[i]) { S.export_lanes = true; continue; }
            S.export_skel = true;
            const int32_t p = TP.inv[i];
            S.export_skew = std::max<int64_t>(S.export_skew, (int64_t)TP.tile_level[TP.tile_of[p]] * S.KS * kRec + (TP.lag[p] & kLagMask));
        }
        DirectArgs &da = S.da;
        da.tiles = P->d_dtiles; da.n_tiles = P->dp.n_tiles; da.lane = P->d_dlane; da.coef = P->d_dcoef; da.q = P->d_dq; da.qch = nullptr;
        if (mode == Mode::Unit) { da.q = P->d_full; da.qch = P->d_chan; }      // UnitMuskingum: q_full and q_ch in params order (unit_state_in)
        da.send_ptr = P->d_dsend_ptr; da.send_lane = P->d_dsend_lane;
        da.in = S.has_in ? io.dev_in : nullptr; da.out = io.dev_out; da.n = n; da.in_rows = (uint32_t)std::max<int64_t>(1, io.rows_in); da.out_rows = (uint32_t)std::max<int64_t>(1, io.rows_out);
        da.in32 = S.has_in ? io.dev_in32 : nullptr; da.out32 = io.dev_out32; da.factor = (int32_t)std::max<int64_t>(1, io.out_factor);
        da.nsub = (int32_t)nsub; da.inv_nsub = 1.0 / (double)nsub;
        da.in32_sel = P->in32_big_endian ? kSelSwap : kSelNative; da.out32_sel = P->out32_big_endian ? kSelSwap : kSelNative;
        da.rec = P->d_ring; da.rec_chunks = (uint32_t)std::max<int64_t>(1, S.rec_chunks); da.np = (int32_t)TP.np;
        da.K = (int32_t)K; da.total = (int32_t)S.T;
        da.exports = export_series; da.n_export = (int32_t)P->n_export;
        TileArgs &w = S.ta;      // the skeleton's k_tile launches
        w.tiles = P->d_ktmeta; w.pos = P->d_kpmeta; w.coef = P->d_kcoef;
        w.sq = P->d_ksq; w.ss = P->d_kss; w.si = P->d_ksi; w.sqch = P->d_ksqch;
        w.exports = export_series; w.n_export = (int32_t)P->n_export;
        w.rec = P->d_ring; w.rec_chunks = Div32((uint32_t)std::max<int64_t>(1, S.rec_chunks));
        w.np = (int32_t)TP.np; w.KC = (int32_t)S.KS; w.n_macro = (int32_t)S.n_macro; w.total = (int32_t)S.total;
        w.has_lat = S.has_in ? 1 : 0; w.nsub = Div32((uint32_t)nsub); w.inv_nsub = 1.0 / (double)nsub;
    }
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
            if (io.out_factor < 1 || kRecRows % step != 0 || T % io.out_factor != 0) { S.open = false; return fail(RR_E_UNSUPPORTED, "float32 output: factor * sub-steps must divide the rows of a record batch (128) and factor the number of rows"); }
        }
        TileArgs &w = S.ta;
        w.tiles = P->d_tmeta; w.pos = P->d_pmeta; w.coef = P->d_coef;
        w.sq = P->d_sq; w.ss = P->d_ss; w.si = P->d_si; w.sqch = P->d_sqch;
        w.exports = export_series; w.n_export = (int32_t)P->n_export;
        w.rec = P->d_ring; w.rec_chunks = Div32((uint32_t)S.rec_chunks);
        w.np = (int32_t)TP.np; w.KC = (int32_t)S.KC; w.n_macro = (int32_t)S.n_macro; w.total = (int32_t)S.total;
        w.has_lat = S.has_in ? 1 : 0; w.nsub = Div32((uint32_t)nsub); w.inv_nsub = 1.0 / (double)nsub;
    }
    if (getenv("RR_VERBOSE") && S.rows_direct)
        fprintf(stderr, "rr: n=%lld T=%lld direct rows: K=%lld tiles=%d window=%d holes=%lld outlets=%lld; skeleton: positions=%lld tiles=%d levels=%d ring_chunks=%lld (%.1f GB)\n",
                (long long)n, (long long)T, (long long)(S.KC * kRec), P->dp.n_tiles, P->direct_window, (long long)P->dp.n_holes, (long long)P->dp.n_exports,
                (long long)P->dp.skel.np, P->dp.skel.n_tiles, P->dp.skel.n_levels, (long long)S.rec_chunks, (double)S.rec_chunks * kRec * P->dp.skel.np * 8 / 1e9);
    else if (getenv("RR_VERBOSE"))
        fprintf(stderr, "rr: n=%lld T=%lld nsub=%lld tiled=%d K=%lld tiles=%d levels=%d block=%d ghosts=%lld ring_chunks=%lld (%.1f GB) lds=%zu\n",
                (long long)n, (long long)T, (long long)nsub, (int)S.wave, (long long)(S.KC * kRec), P->tp.n_tiles, P->tp.n_levels, P->tp.block,
                (long long)P->tp.n_ghost, (long long)S.rec_chunks, S.wave ? (double)S.rec_chunks * kRec * P->tp.np * 8 / 1e9 : 0.0,
                tile_lds_bytes(P->wave_threads));
    if (!S.wave && !S.rows_direct) {
        // work ring in engine order: lateral rows come in, discharge rows overwrite them in place; rows stay until the
        // outlet-most reaches have passed them
        const int64_t lag_rows = (dmax + nsub - 1) / nsub;
        S.ring_rows = S.direct ? 0 : std::min<int64_t>(T, lag_rows + 2 * C + 2);
        if (S.ring_rows > 0xFFFFFFFFLL || T > 0x7FFFFFFFLL) { S.open = false; return fail(RR_E_INVALID, "route: too many time rows"); }
        if (S.ring_rows * n > P->ring_cap || (!S.direct && C * n > P->mrows_cap) || (host_io && C * n > P->stage_cap)) {      // prepare_call sized them
            S.open = false;
            return fail(RR_E_STATE, "route: work rows of the streaming kernel were not reserved");
        }
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
    if (S.rows_direct && S.max_samples > 0) S.max_samples = (size_t)std::min<int64_t>(4096, S.n_tasks);            // every direct launch
    S.max_samples = std::min(S.max_samples, P->ev.size() / 2);      // events are made by rr_plan_reserve, never here
    if (!P->ev_first) { S.open = false; return fail(RR_E_STATE, "route: the plan's events were not reserved"); }
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
        ua.a2 = P->unit_general ? P->d_a2 : nullptr; ua.c1own = P->d_c1own;
        if (P->unit_general) { ua.t.c1row = nullptr; ua.zc = P->d_z + (tau % 3) * n; ua.za = P->d_z + ((tau + 2) % 3) * n; }
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

// Tile = one position per thread, 512 threads, two workgroups per CU: 16 waves per CU whose two record buffers fill the register
// file, and while the waves of one workgroup wait to issue their record loads the other one ticks (368 ms per year at 1M reaches
// against 391 with one 1,024-thread workgroup and 382 with four 256-thread ones; smaller tiles for smaller networks do not pay
// either: profiles/r03_tile_size_sweep.txt).
constexpr int kTileThreads = 512;
tile_kernel_t tile_kernel(bool unit, bool sub, bool lean = false, bool nolat = false)
{
    constexpr int T = kTileThreads;
    if (lean && nolat && !unit && !sub) return (tile_kernel_t)k_tile<T, false, false, true, true>;      // channel-only routing, one sub-step per row: the short tick without a lateral term
    return unit ? (sub ? (tile_kernel_t)k_tile<T, true, true> : (lean ? (tile_kernel_t)k_tile<T, true, false, true> : (tile_kernel_t)k_tile<T, true, false>))
                : (sub ? (tile_kernel_t)k_tile<T, false, true> : (lean ? (tile_kernel_t)k_tile<T, false, false, true> : (tile_kernel_t)k_tile<T, false, false>));
}

// Launch d of the time-tiled schedule: the tasks (tile, macro-chunk d - level) of every tile whose macro-chunk exists.
// Tiles are stored by level, so they are one contiguous range; a tile with no active position returns at once.
// TP / w / wide tiles: the plan's own tiles, or the skeleton of the direct row path (its levels start at 1).
int launch_tile_diag(rr_plan *P, const rr::TilePlan &TP, TileArgs &w, int32_t n_wide, const double *coef, const double *coef_unit, int64_t d, bool sampled)
{
    Session &S = P->ses;
    const int64_t l_lo = std::max<int64_t>(0, d - (S.n_macro - 1)), l_hi = std::min<int64_t>(TP.n_levels - 1, d);
    if (l_hi < l_lo) return RR_OK;
    int64_t t_lo = TP.level_start[l_lo], t_hi = TP.level_start[l_hi + 1];
    // tiles are sorted by their smallest lag inside a level; while the pipeline fills, the tiles of level 0 that
    // have not started yet are a suffix of it
    const int64_t K = (int64_t)w.KC * kRec;      // (the skeleton of the direct row path may run shorter tasks than its lanes)
    if (l_lo == 0) {
        const int64_t end0 = TP.level_start[1];
        int64_t hi = std::min<int64_t>(t_hi, end0);
        while (hi > t_lo && (d + 1) * K <= TP.tile_lag_lo[hi - 1]) --hi;
        if (t_hi <= end0) t_hi = hi;     // only level 0 in this launch: trim; otherwise the idle ones just return
    }
    if (t_hi <= t_lo) return RR_OK;
    w.diag = (int32_t)d; w.t_first = (int32_t)t_lo; w.t_last = (int32_t)t_hi - 1;
    // every fourth launch is bracketed by HIP events, full or not (fill and drain launches run fewer tiles), so the
    // sampled average is the average rocprofv3 reports for the kernel; the position-ticks of a sample are those of the
    // tiles that run a task in it
    const bool sample = sampled && S.max_samples > 0 && (P->prof_launches % 4) == 0 && (size_t)P->prof_brackets < S.max_samples;
    if (sample) HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets], S.stream));
    // one workgroup per resident slot (16 waves per CU); each walks its share of the launch's tiles
    const dim3 g((unsigned)std::min<int64_t>(t_hi - t_lo, (int64_t)P->cu_count * (1024 / P->wave_threads)));
    const size_t lds_bytes = tile_lds_bytes(P->wave_threads);
    // With one sub-step per row every mode runs the short tick (k_tile<..., LEAN>), and beside it the general kernel for
    // the tiles the short tick does not take (a reach with more than three upstream reaches); RR_TILE_LEAN=0: the general one
    const bool unit = S.mode == Mode::Unit;
    const bool lean = S.nsub == 1 && P->lean_enabled;      // every router's default (dt_routing = dt_runoff); sub-steps keep the general tick
    w.tile_filter = lean ? 1 : 0;
    w.coef = (lean && unit) ? coef_unit : coef;
    hipLaunchKernelGGL(tile_kernel(unit, S.nsub > 1, lean, S.mode == Mode::Muskingum), g, dim3((unsigned)P->wave_threads), lds_bytes, S.stream, w);
    if (lean && n_wide > 0) {
        w.tile_filter = 2; w.coef = coef;
        hipLaunchKernelGGL(tile_kernel(unit, false, false), g, dim3((unsigned)P->wave_threads), lds_bytes, S.stream, w);
    }
    if (sample) {
        HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets + 1], S.stream));
        // position-ticks of the sample: the tiles that run a task in this launch (k_tile's select(): the macro-chunk exists and
        // some position is active in it); a tile that returns at once moves nothing -- in a short call most of a launch's tiles
        int64_t moved = 0;
        for (int64_t c = t_lo; c < t_hi; ++c) {
            const int64_t m = d - TP.tile_level[c];
            if (m < 0 || m >= S.n_macro || m * K >= TP.tile_lag_hi[c] + S.total || (m + 1) * K <= TP.tile_lag_lo[c]) continue;
            moved += TP.tile_ptr[c + 1] - TP.tile_ptr[c];
        }
        P->ev_reaches.push_back(moved * K);
        P->prof_samples += K;
        ++P->prof_brackets;
    }
    return RR_OK;
}

// Whether launch d of a tile plan has any tile to run (launch_tile_diag returns early otherwise).
bool tile_diag_launches(const rr_plan *P, const rr::TilePlan &TP, int64_t d)
{
    const Session &S = P->ses;
    const int64_t l_lo = std::max<int64_t>(0, d - (S.n_macro - 1)), l_hi = std::min<int64_t>(TP.n_levels - 1, d);
    return l_hi >= l_lo && TP.level_start[l_hi + 1] > TP.level_start[l_lo];
}

int session_launch_diag(rr_plan *P, int64_t d)
{
    int rc = launch_tile_diag(P, P->tp, P->ses.ta, P->n_wide_tiles, P->d_coef, P->d_coef_unit, d, true);
    ++P->prof_launches;
    return rc;
}

// Boundary inflow of a partitioned network: the ghost series (total sub-steps x ghosts, row = sub-step) is a matrix of
// tick-rows like the lateral rows, and its columns become the records of the ghost positions by the same pass.
void launch_ghost_permute(rr_plan *P, int64_t batch)
{
    Session &S = P->ses;
    RecPermArgs ra{};
    ra.in32_sel = P->in32_big_endian ? kSelSwap : kSelNative; ra.out32_sel = P->out32_big_endian ? kSelSwap : kSelNative;
    // (direct row path: the ghosts' records are those of the skeleton's positions that mirror them)
    ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = P->n_ghost; ra.np = S.rows_direct ? P->dp.skel.np : P->tp.np; ra.T = S.total; ra.total = S.total;
    ra.batch = batch; ra.nsub = Div32(1u); ra.colmeta = S.rows_direct ? P->d_kghostmeta : P->d_ghostmeta; ra.scale = nullptr;
    ra.rows = RowView{const_cast<double *>(S.ghost_series), P->n_ghost, 0, (uint32_t)S.total};
    ra.factor = Div32(1u);
    hipLaunchKernelGGL(k_rec_in<false>, dim3((unsigned)((P->n_ghost + kRecInCols - 1) / kRecInCols)), dim3(kRecInThreads), 0, rec_stream(P), ra);
}

typedef void (*rec_in_uh_t)(const RecPermArgs, const UhArgs);
constexpr int kUhFusedMaxTaps = 64;
int uh_padded_taps(int64_t n_ks) { return n_ks <= 16 ? 16 : (n_ks <= 48 ? 48 : 64); }
rec_in_uh_t rec_in_uh_kernel(bool sub, int64_t n_ks, int batches, bool in32 = false)
{
    const int nk = uh_padded_taps(n_ks);
#define RR_UHIN_IN(NK_, B_, S_) (in32 ? (rec_in_uh_t)k_rec_in_uh<S_, NK_, B_, true> : (rec_in_uh_t)k_rec_in_uh<S_, NK_, B_, false>)
#define RR_UHIN_SUB(NK_, B_) (sub ? RR_UHIN_IN(NK_, B_, true) : RR_UHIN_IN(NK_, B_, false))
#define RR_UHIN_PICK(NK_) (batches == 2 ? RR_UHIN_SUB(NK_, 2) : RR_UHIN_SUB(NK_, 1))
    return nk == 16 ? RR_UHIN_PICK(16) : (nk == 48 ? RR_UHIN_PICK(48) : RR_UHIN_PICK(64));
#undef RR_UHIN_PICK
#undef RR_UHIN_SUB
#undef RR_UHIN_IN
}

// `count` batches from `batch` on: 1, or 2 for the fused convolution when its rows are there (session_advance_tile)
void launch_rec_permute(rr_plan *P, bool in, int64_t batch, int count = 1)
{
    Session &S = P->ses;
    const int64_t n = P->h.n;
    RecPermArgs ra{};
    ra.in32_sel = P->in32_big_endian ? kSelSwap : kSelNative; ra.out32_sel = P->out32_big_endian ? kSelSwap : kSelNative;
    ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = n; ra.np = P->tp.np; ra.T = S.T; ra.total = S.total; ra.batch = batch;
    ra.nsub = Div32((uint32_t)S.nsub);
    ra.colmeta = P->d_colmeta;
    ra.scale = (in && S.mode == Mode::Rapid) ? P->d_c4_params : nullptr;
    ra.rows = in ? RowView{const_cast<double *>(S.io.dev_in), n, 0, (uint32_t)S.io.rows_in}
                 : RowView{S.io.dev_out, n, 0, (uint32_t)std::max<int64_t>(1, S.io.rows_out)};
    ra.rows32 = in ? nullptr : S.io.dev_out32;
    ra.rows_in32 = in ? S.io.dev_in32 : nullptr;
    ra.factor = Div32((uint32_t)std::max<int64_t>(1, S.io.out_factor));
    ra.clamp = S.nsub > 1 ? 0 : (S.mode == Mode::Unit ? 2 : 1);
    // The out-pass walks its column tiles in XCD-contiguous order (xcd_swizzle): where the row pitch is not a whole number
    // of 128-byte lines neighbouring tiles write parts of the same lines, and on one XCD those meet in its L2 (1.25M-reach
    // part with an odd column count: 522 -> 473 ms per year; 1% - 2% with an aligned pitch too).  The in-pass gains
    // nothing from it (its shared lines are reads) and the two together were slower: profiles/r02_rec_swizzle.txt.
    ra.swizzle = in ? 0 : 1;
    // one column tile per workgroup, dispatched in address order: a persistent grid with the next tile's loads in flight was no
    // faster (427 / 469 us per 128 rows against 419 / 434-448 at 1M reaches; a plain copy shows the same, profiles/r03_hbm_probe_*.txt)
    const dim3 gp((unsigned)((n + (in ? kRecInCols : kRecOutCols) - 1) / (in ? kRecInCols : kRecOutCols)));
    const bool sub = S.nsub > 1;
    hipStream_t st = rec_stream(P);
    const int aux = aux_begin(P, in ? 0 : 1, st);
    if (in && S.io.runoff) {
        const dim3 gr((unsigned)((n + kRunoffInThreads - 1) / kRunoffInThreads), (unsigned)kRecBatch);
        if (S.io.runoff->is_f32) hipLaunchKernelGGL(k_rec_in_runoff<float>, gr, dim3(kRunoffInThreads), 0, st, ra, *S.io.runoff);
        else hipLaunchKernelGGL(k_rec_in_runoff<double>, gr, dim3(kRunoffInThreads), 0, st, ra, *S.io.runoff);
    } else if (in && S.io.uh_kernel) {
        UhArgs ua{S.io.uh_kernel, S.io.uh_state, (int32_t)S.io.uh_nks};
        hipLaunchKernelGGL(rec_in_uh_kernel(sub, S.io.uh_nks, count, ra.rows_in32 != nullptr), dim3((unsigned)((n + kUhCols - 1) / kUhCols)), dim3(uh_threads(count)),
                           rec_in_uh_lds_bytes(uh_padded_taps(S.io.uh_nks), count), st, ra, ua);
    } else if (in && ra.rows_in32) {
        if (sub) hipLaunchKernelGGL((k_rec_in<true, true>), gp, dim3(kRecInThreads), 0, st, ra);
        else hipLaunchKernelGGL((k_rec_in<false, true>), gp, dim3(kRecInThreads), 0, st, ra);
    } else if (in) {
        if (sub) hipLaunchKernelGGL(k_rec_in<true>, gp, dim3(kRecInThreads), 0, st, ra);
        else hipLaunchKernelGGL(k_rec_in<false>, gp, dim3(kRecInThreads), 0, st, ra);
    } else if (ra.rows32) {
        if (sub) hipLaunchKernelGGL((k_rec_out<true, true>), gp, dim3(kRecOutThreads), 0, st, ra);
        else hipLaunchKernelGGL((k_rec_out<false, true>), gp, dim3(kRecOutThreads), 0, st, ra);
    } else {
        if (sub) hipLaunchKernelGGL((k_rec_out<true, false>), gp, dim3(kRecOutThreads), 0, st, ra);
        else hipLaunchKernelGGL((k_rec_out<false, false>), gp, dim3(kRecOutThreads), 0, st, ra);
    }
    aux_end(P, aux, st);
}

// Time-tiled schedule: batches of kRecRows (128) tick-rows become records as soon as their rows are there and their ring slots
// are free, launches run while their input is present, finished batches leave.
int session_advance_tile(rr_plan *P, int64_t rows_ready, int64_t ghost_ready, int64_t *export_ready)
{
    Session &S = P->ses;
    const int64_t dmax = P->h.depth - 1, levels = P->tp.n_levels, K = S.KC * kRec;
    const int64_t ticks_ready = std::min(rows_ready, S.T) * S.nsub;
    for (;;) {
        bool progressed = false;
        // one batch of tick-rows -> records; a record slot is recycled only after every tick-row it can hold has left.
        // Lateral rows and boundary sub-steps (the ghost series of a partitioned network) advance separately: a ghost in a
        // tile of level l at lag L is first read (l K + L) ticks into the schedule, so the boundary may trail the rows.
        // Batch j writes, for a position of lag L, the records whose last tick-row lies in the batch: chunks up to
        // (R (j + 1) + L) / 16, R = kRecRows.  One ring revolution earlier that slot held the same position's tick-rows up to
        // R (j + 1) - 16 rec_chunks + 15, whatever L is: those must have left.
        auto slot_free = [&](int64_t j) {
            const int64_t must_have_left = kRecRows * (j + 1) + kRec - kRec * S.rec_chunks;
            return must_have_left <= 0 || S.ticks_stored >= std::min(S.total, must_have_left);
        };
        if (S.has_in && S.in_batches < S.n_in_batches && ticks_ready >= std::min(kRecRows * (S.in_batches + 1), S.total) && slot_free(S.in_batches)) {
            // the fused convolution takes the next batch along when that one's rows and ring slots are there too (less to read per value)
            const int count = (S.io.uh_kernel && !S.io.runoff && P->uh_pairs && S.in_batches + 2 <= S.n_in_batches &&
                               ticks_ready >= std::min(kRecRows * (S.in_batches + 2), S.total) && slot_free(S.in_batches + 1)) ? 2 : 1;
            launch_rec_permute(P, true, S.in_batches, count);
            S.in_batches += count;
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
            const int64_t top = std::min(std::min(S.diag + 1, S.n_macro) * S.KC - 1, (S.total_ticks - 1) / kRec);      // (a task longer than what is left of the call -- RR_WAVE_K -- ends with the call's last chunk)
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

// The direct row path (rr_kernels_direct.hpp).  Launch d: k_direct routes rows [d K, (d + 1) K) of every small subtree straight from
// and to the caller's rows, forwarding the skeleton's lateral inflow and the subtrees' outflow into the skeleton's records; then
// the skeleton's tiles of level l run macro-chunk d - l on those records (levels start at 1: what they read was written by an
// earlier launch); rows all of whose skeleton reaches are final get their holes patched from the records (k_rec_out over the
// holes' columns), 128 rows at a time, which also frees their ring slots.
// (The skeleton's launches and the out-pass on a second stream beside the direct launches, with or without CUs set aside for them,
// changed nothing -- 212 to 226 ms per year against 212: profiles/r04_direct_second_stream_ab.txt -- the out-pass's scattered
// 8-byte stores, 5 % of the columns, compete for the same memory system.)
int session_advance_direct(rr_plan *P, int64_t rows_ready, int64_t ghost_ready, int64_t *export_ready)
{
    Session &S = P->ses;
    const rr::TilePlan &TP = P->dp.skel;
    const int64_t dmax = P->h.depth - 1, levels = TP.n_levels, K = S.KC * kRec, Kt = K * S.nsub, KS = S.KS * kRec, per = Kt / KS, n = P->h.n;      // K rows = Kt ticks per direct task
    rows_ready = std::min(rows_ready, S.T);
    const bool skel = TP.n_tiles > 0, ghosts = skel && P->n_ghost > 0;
    // a record slot is recycled only after every tick-row it can hold has left (the same rule as session_advance_tile's)
    auto slot_free = [&](int64_t j) {      // (a batch of the boundary in-pass writes whole records up to tick-row 128 (j + 1) + lag, zeros past the call's end)
        const int64_t must_have_left = kRecRows * (j + 1) + kRec - kRec * S.rec_chunks;
        return must_have_left <= 0 || S.ticks_stored >= std::min(S.total, must_have_left);
    };
    for (;;) {
        bool progressed = false;
        while (S.d_done < S.n_tasks) {
            const int64_t d = S.d_done;
            if (rows_ready < std::min((d + 1) * K, S.T)) break;
            if (skel) {      // this launch writes record slots up to tick (d + 1) K + dmax: whatever they held one revolution earlier must have left
                const int64_t top = std::min((d + 1) * Kt + dmax, S.total_ticks) / kRec;
                if (top >= S.rec_chunks && S.ticks_stored < std::min(S.total, kRec * (top - S.rec_chunks + 1))) break;
            }
            const bool sample = S.max_samples > 0 && (size_t)P->prof_brackets < S.max_samples;
            if (sample) HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets], S.stream));
            S.da.m = (int32_t)d;
            const dim3 g((unsigned)std::min<int64_t>(P->dp.n_tiles, (int64_t)P->cu_count));
            hipLaunchKernelGGL(direct_kernel(S.io.dev_in32 != nullptr && S.has_in, S.da.out32 != nullptr, S.nsub > 1, S.mode == Mode::Unit ? 1 : 0), g, dim3(kDirectThreads), direct_lds_bytes(P->direct_window), S.stream, S.da);
            if (sample) {
                HIPCHK(hipEventRecord(P->ev[2 * P->prof_brackets + 1], S.stream));
                P->ev_reaches.push_back(n * (std::min((d + 1) * K, S.T) - d * K));
                P->prof_samples += K;
                ++P->prof_brackets;
            }
            ++P->prof_launches;
            ++S.d_done;
            progressed = true;
        }
        const int64_t rows_routed = S.d_done >= S.n_tasks ? S.T : S.d_done * K;      // by the lanes
        const int64_t ticks_routed = rows_routed * S.nsub;
        S.rows_loaded = rows_routed;
        // The boundary series of a partitioned network becomes the records of the skeleton's ghosts that mirror the boundary reaches,
        // 128 tick-rows at a time (k_rec_in's batches), never ahead of the rows the lanes have routed: the ring's slots up to there are free.
        while (ghosts && S.ghost_batches < S.n_in_batches) {
            const int64_t need = std::min(kRecRows * (S.ghost_batches + 1), S.total);
            if (ghost_ready < need || need > ticks_routed || !slot_free(S.ghost_batches)) break;
            launch_ghost_permute(P, S.ghost_batches);
            ++S.ghost_batches;
            progressed = true;
        }
        const int64_t ghost_loaded = !ghosts || S.ghost_batches >= S.n_in_batches ? S.total : std::max<int64_t>(0, kRecRows * S.ghost_batches - 15);
        // The skeleton's launch j runs its tiles of level l (from 1) on macro-chunk j - l, KS ticks each: what they read -- the records the
        // lanes forwarded and the boundary records, ticks below j KS -- is there once the lanes have routed that many rows.
        while (skel && S.diag < S.n_diags) {
            const int64_t j = S.diag;
            if (S.d_done < S.n_tasks && j > S.d_done * per - 1) break;
            if (ghost_loaded < std::min(S.total, j * KS)) break;
            if (tile_diag_launches(P, TP, j)) {      // (an empty bracket would be counted as a launch of the skeleton)
                const int aux = aux_begin(P, 2, S.stream);
                int rc = launch_tile_diag(P, TP, S.ta, P->n_kwide, P->d_kcoef, P->d_kcoef, j, false);
                aux_end(P, aux, S.stream);      // before rc is looked at: rr_plan_profile_aux reads both events of every sample
                if (rc) return rc;
            }
            ++S.diag;
            progressed = true;
        }
        // rows the direct tiles have written, and of those the rows whose skeleton reaches are final too
        int64_t done = ticks_routed;      // in sub-steps
        if (skel) {
            const int64_t m_done = S.diag - levels;      // the tiles of the last level have finished macro-chunk diag - levels
            int64_t sk = S.diag >= S.n_diags ? S.total : (m_done >= 0 ? std::max<int64_t>(0, (m_done + 1) * KS - dmax) : 0);
            done = std::min(done, std::min(sk, S.total));
            while (S.out_batches < S.n_out_batches && done >= std::min(kRecRows * (S.out_batches + 1), S.total)) {
                RecPermArgs ra{};
    ra.in32_sel = P->in32_big_endian ? kSelSwap : kSelNative; ra.out32_sel = P->out32_big_endian ? kSelSwap : kSelNative;
                ra.rec = P->d_ring; ra.rec_chunks = Div32((uint32_t)S.rec_chunks); ra.n = P->n_kholes; ra.np = TP.np; ra.T = S.T; ra.total = S.total;
                ra.batch = S.out_batches; ra.nsub = Div32((uint32_t)S.nsub); ra.colmeta = P->d_kholemeta; ra.cols = P->d_kholecol; ra.scale = nullptr;
                ra.rows = RowView{S.io.dev_out, n, 0, (uint32_t)std::max<int64_t>(1, S.io.rows_out)};
                ra.rows32 = S.io.dev_out32;
                ra.factor = Div32((uint32_t)std::max<int64_t>(1, S.io.out_factor)); ra.clamp = S.nsub > 1 ? 0 : 1; ra.swizzle = 0;
                const int aux = aux_begin(P, 3, S.stream);
                const dim3 gh((unsigned)((P->n_kholes + kRecOutCols - 1) / kRecOutCols));
                if (P->n_kholes > 0) {
                    if (S.nsub > 1) { if (ra.rows32) hipLaunchKernelGGL((k_rec_out<true, true>), gh, dim3(kRecOutThreads), 0, S.stream, ra); else hipLaunchKernelGGL((k_rec_out<true, false>), gh, dim3(kRecOutThreads), 0, S.stream, ra); }
                    else if (ra.rows32) hipLaunchKernelGGL((k_rec_out<false, true>), gh, dim3(kRecOutThreads), 0, S.stream, ra);
                    else hipLaunchKernelGGL((k_rec_out<false, false>), gh, dim3(kRecOutThreads), 0, S.stream, ra);
                }
                aux_end(P, aux, S.stream);
                ++S.out_batches;
                S.ticks_stored = std::min(S.total, kRecRows * S.out_batches);
                progressed = true;
            }
            S.rows_stored = S.ticks_stored / S.nsub;
        } else {
            S.ticks_stored = done; S.rows_stored = done / S.nsub;
        }
        if (!progressed) break;
    }
    if (S.d_done >= S.n_tasks && S.diag >= S.n_diags) S.tau = S.total_ticks;
    if (export_ready) {      // a lane's export is final with its row; a skeleton reach's once its tile's level has passed the tick
        int64_t e = S.d_done >= S.n_tasks ? S.total : S.d_done * Kt;
        if (S.export_skel) e = std::min(e, S.diag >= S.n_diags ? S.total : S.diag * KS - S.export_skew);
        *export_ready = P->n_export > 0 ? std::max<int64_t>(0, std::min(e, S.total)) : S.total;
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
    if (S.rows_direct) return session_advance_direct(P, rows_ready, std::min(ghost_ready, S.total), export_ready);
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
    if (!complete && getenv("RR_VERBOSE"))
        fprintf(stderr, "rr: incomplete call: T=%lld total=%lld tau=%lld/%lld rows_loaded=%lld rows_stored=%lld | direct=%d tasks %lld/%lld diag %lld/%lld in_batches %lld ghost_batches %lld/%lld out_batches %lld/%lld "
                        "ticks_stored=%lld K=%lld KS=%lld chunks=%lld macro=%lld\n", (long long)S.T, (long long)S.total, (long long)S.tau, (long long)S.total_ticks, (long long)S.rows_loaded, (long long)S.rows_stored,
                (int)S.rows_direct, (long long)S.d_done, (long long)S.n_tasks, (long long)S.diag, (long long)S.n_diags, (long long)S.in_batches, (long long)S.ghost_batches, (long long)S.n_in_batches,
                (long long)S.out_batches, (long long)S.n_out_batches, (long long)S.ticks_stored, (long long)(S.KC * kRec), (long long)(S.KS * kRec), (long long)S.rec_chunks, (long long)S.n_macro);
    if (!complete) return fail(RR_E_STATE, "routing call closed before all of its time steps were routed");
    if (P->h.n == 0 || S.total == 0) return RR_OK;
    if (S.bracket_open) P->prof_samples -= P->prof_samples % kSampleGroup;   // incomplete bracket: not counted
    HIPCHK(hipEventRecord(P->ev_last, S.stream));
    HIPCHK(hipGetLastError());
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
    if (P->direct_now) {      // the lanes carry their discharge in params order; the skeleton's positions and ghosts as k_tile wants them
        HIPCHK(hipMemcpyAsync(P->d_dq, d_q, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        const int64_t np = P->dp.skel.np;
        if (np > 0)
            hipLaunchKernelGGL(k_tile_state_in, grid1(np), dim3(kBlock), 0, stream, P->d_ksq, P->d_kss, P->d_ksi, d_q, P->d_kperm, P->d_kpmeta, (int32_t)np);
        return RR_OK;
    }
    if (use_wave(P, mode)) {
        const int64_t np = P->tp.np;
        hipLaunchKernelGGL(k_tile_state_in, grid1(np), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_si, d_q, P->d_tperm,
                           P->d_pmeta, (int32_t)np);
        return RR_OK;
    }
    hipLaunchKernelGGL(k_state_in, grid1(n), dim3(kBlock), 0, stream, P->d_x, P->d_x + n, P->d_x + 2 * n, d_q,
                       P->d_perm, (int32_t)n);
    return RR_OK;
}

void launch_state_out(rr_plan *P, Mode mode, double *d_q, int64_t total, hipStream_t stream)
{
    const int64_t n = P->h.n;
    if (P->direct_now) {
        (void)hipMemcpyAsync(d_q, P->d_dq, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream);
        const int64_t np = P->dp.skel.np;
        if (np > 0) hipLaunchKernelGGL(k_skel_state_out, grid1(np), dim3(kBlock), 0, stream, d_q, (const double *)P->d_ksq, P->d_kperm, P->d_kpmeta, (int32_t)np);
        return;
    }
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
    H.chunk_rows = std::max<int64_t>(16, std::min<int64_t>(4096, ((H.chunk_mib << 20) / (n * 8) + 15) / 16 * 16));
    if (n * 8 * 64 <= (int64_t{1} << 30)) H.chunk_rows = std::max<int64_t>(H.chunk_rows, 64);
    H.ring_chunks = std::max<int64_t>(8, (2 * kRecRows + 15) / H.chunk_rows + 6);      // a batch of rows + its 15-row overlap stays readable
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
    double t_join = 0.0, t_wait = 0.0;      // RR_VERBOSE: seconds this thread waited for the copy threads / for events
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (copied_out < nchunks) {
        pool.clear();
        bool progressed = false;
        const bool fill = host_in && filled < nchunks && filled < h2d_issued + kPinned;
        if (fill) {
            if (filled >= kPinned) { const double t0 = now(); RR_PIPE(hipEventSynchronize(H.ev_h2d[filled - kPinned])); t_wait += now() - t0; }      // the buffer's previous chunk has left
            parallel_copy(H.pin_in[filled % kPinned], host_in + filled * C * n, (size_t)(rows_of(filled) * n), H.copy_threads, pool);
        }
        bool empty = false;
        if (copied_out < d2h_issued) {      // only a download that HAS arrived: waiting for one here would stall the uploads behind it
            const hipError_t q = hipEventQuery(H.ev_d2h[copied_out]);
            if (q == hipSuccess) empty = true;
            else if (q != hipErrorNotReady) return bail(RR_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
        if (empty) {
            parallel_copy(host_out + copied_out * C * n, H.pin_out[copied_out % kPinned], (size_t)(rows_of(copied_out) * n), H.copy_threads, pool);
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
        { const double t0 = now(); for (auto &t : pool) t.join(); t_join += now() - t0; }
        if (fill) ++filled;
        if (empty) ++copied_out;
        if (!progressed && !fill && !empty) {
            if (copied_out < d2h_issued) { const double t0 = now(); RR_PIPE(hipEventSynchronize(H.ev_d2h[copied_out])); t_wait += now() - t0; }      // nothing else to do but wait for it
            else return bail(RR_E_STATE, "host pipeline: no stage can make progress");
        }
    }
#undef RR_PIPE
    if (getenv("RR_VERBOSE"))
        fprintf(stderr, "rr: host pipeline: %lld chunks of %lld rows, %d copy threads per direction; waited %.1f ms for the copy threads, %.1f ms for transfers\n",
                (long long)nchunks, (long long)C, H.copy_threads, t_join * 1e3, t_wait * 1e3);
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
    {   // host rows reach the time-tiled kernel through the PCIe pipeline's device rings (they are "device rows" to the schedule)
        const bool plain = !host_rows && (mode == Mode::Muskingum || io.dev_in || io.dev_in32) && (io.dev_out || io.dev_out32) && !io.uh_kernel && !io.runoff;      // rows in device arrays, float64 or float32: the direct row path applies
        int rc = prepare_call(P, mode, T, nsub, false, false, !host_rows, plain, 0, 0, io.dev_out32 != nullptr);
        if (rc == RR_OK && host_rows && !P->wave_now) rc = prepare_call(P, mode, T, nsub, true, true, false);
        if (rc == RR_OK && host_rows && P->wave_now) rc = host_pipe_prepare(P);
        if (rc) return rc;
    }
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
    if (P->direct_now) {      // the lanes carry q_full and q_ch in params order (zeros on headwaters); the skeleton's positions as k_tile<UNIT> wants them
        e0 = hipMemsetAsync(P->d_full, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(P->d_chan, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess && ni > 0)
            hipLaunchKernelGGL(k_unit_scatter, grid1(ni), dim3(kBlock), 0, stream, P->d_full, P->d_chan, d_qfull, d_qch, P->d_inner_idx, (int32_t)ni);
        const int64_t np = P->dp.skel.np;
        if (e0 == hipSuccess && np > 0)
            hipLaunchKernelGGL(k_tile_unit_state_in, grid1(np), dim3(kBlock), 0, stream, P->d_ksq, P->d_kss, P->d_ksi, P->d_ksqch,
                               (const double *)P->d_full, (const double *)P->d_chan, P->d_kperm, P->d_kpmeta, (int32_t)np);
    } else if (use_wave(P, Mode::Unit)) {   // q_full / q_ch scattered to params order (zeros on headwaters), then gathered position by position
        e0 = hipMemsetAsync(P->d_full, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(P->d_chan, 0, n * sizeof(double), stream);
        if (e0 == hipSuccess && ni > 0)
            hipLaunchKernelGGL(k_unit_scatter, grid1(ni), dim3(kBlock), 0, stream, P->d_full, P->d_chan, d_qfull, d_qch, P->d_inner_idx, (int32_t)ni);
        if (e0 == hipSuccess)
            hipLaunchKernelGGL(k_tile_unit_state_in, grid1(P->tp.np), dim3(kBlock), 0, stream, P->d_sq, P->d_ss, P->d_si, P->d_sqch,
                               (const double *)P->d_full, (const double *)P->d_chan, P->d_tperm, P->d_pmeta, (int32_t)P->tp.np);
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
    if (P->direct_now) {      // the skeleton's reaches back into the params-order arrays the lanes kept, then the inner reaches' pairs
        const int64_t np = P->dp.skel.np;
        if (np > 0) {
            hipLaunchKernelGGL(k_skel_state_out, grid1(np), dim3(kBlock), 0, stream, P->d_full, (const double *)P->d_ksq, P->d_kperm, P->d_kpmeta, (int32_t)np);
            hipLaunchKernelGGL(k_skel_state_out, grid1(np), dim3(kBlock), 0, stream, P->d_chan, (const double *)P->d_ksqch, P->d_kperm, P->d_kpmeta, (int32_t)np);
        }
        hipLaunchKernelGGL(k_unit_gather, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull, (const double *)P->d_chan, (const double *)P->d_full, P->d_inner_idx, (int32_t)ni);
    } else if (use_wave(P, Mode::Unit))
        hipLaunchKernelGGL(k_tile_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                           (const double *)P->d_sq, (const double *)P->d_sqch, P->d_inner_idx, P->d_tinv, (int32_t)ni);
    else
        hipLaunchKernelGGL(k_unit_state_out, grid1(ni), dim3(kBlock), 0, stream, d_qch, d_qfull,
                           (const double *)P->d_x, n, (const double *)P->d_qch, P->d_lag, P->d_inner_pos, (int32_t)ni, total);
}

template <typename TIn>
int uh_convolve_core(const double *d_kernel, double *d_state, const TIn *d_lateral, double *d_out, int64_t T, int64_t n_ks, int64_t n, hipStream_t stream, uint32_t sel);

int unit_like(rr_plan *P, double *q_ch, double *q_full, const Rows &io_in, int64_t T, int64_t nsub,
              hipStream_t stream, bool q_on_host, double *d_q_final = nullptr, double *uh_state_inout = nullptr)
{
    const int64_t n = P->h.n, ni = (int64_t)P->h.inner_pos.size();
    if (n == 0 || T == 0) return RR_OK;
    Rows io = io_in;
    const bool host_rows = io.host_in != nullptr || io.host_out != nullptr;
    {
        // rows in device arrays -- convolved lateral inflow, or runoff depths (float64 / float32) with the unit-hydrograph kernel: the direct row path applies
        const bool plain = !host_rows && (io.dev_in || io.dev_in32) && (io.dev_out || io.dev_out32) && !io.runoff && (io.uh_kernel || !io.dev_in32) && (!io.uh_kernel || uh_state_inout);
        int rc = prepare_call(P, Mode::Unit, T, nsub, false, false, !host_rows, plain, 0, 0, io.dev_out32 != nullptr, io.uh_kernel != nullptr);
        if (rc == RR_OK && host_rows && !P->wave_now) rc = prepare_call(P, Mode::Unit, T, nsub, true, true, false);
        if (rc == RR_OK && host_rows && P->wave_now) rc = host_pipe_prepare(P);
        if (rc) return rc;
    }
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
    if (rc == RR_OK && P->direct_now && io.uh_kernel) {
        // the direct row path: the convolution first, into the work rows (choose_schedule: uh), carry-over state updated in place;
        // the lanes then route those rows
        if (T * n > P->mrows_cap) rc = fail(RR_E_STATE, "route: the convolved rows of the direct row path were not reserved");
        else if (io.dev_in32) rc = uh_convolve_core<float>(io.uh_kernel, uh_state_inout, io.dev_in32, P->d_mrows, T, io.uh_nks, n, stream, P->in32_big_endian ? kSelSwap : kSelNative);
        else rc = uh_convolve_core<double>(io.uh_kernel, uh_state_inout, io.dev_in, P->d_mrows, T, io.uh_nks, n, stream, kSelNative);
        io.dev_in = P->d_mrows; io.dev_in32 = nullptr; io.rows_in = T; io.uh_kernel = nullptr; io.uh_state = nullptr;
    }
    if (rc) { if (tmp) (void)hipFree(tmp); return rc; }
    if (rc == RR_OK) rc = piped ? route_host_pipelined(P, Mode::Unit, T, nsub, io.host_in, io.host_out, stream) : route_core(P, Mode::Unit, T, nsub, io, stream);
    if (rc == RR_OK && wave && d_q_final)      // every reach: a headwater's state is its last lateral inflow, an inner reach's q_full
        hipLaunchKernelGGL(k_tile_state_out, grid1(n), dim3(kBlock), 0, stream, d_q_final, (const double *)P->d_sq, P->d_tinv, (int32_t)n);
    if (rc == RR_OK && P->direct_now && d_q_final) {      // the lanes' columns hold exactly that; the skeleton's reaches join them
        const int64_t np = P->dp.skel.np;
        if (np > 0) hipLaunchKernelGGL(k_skel_state_out, grid1(np), dim3(kBlock), 0, stream, P->d_full, (const double *)P->d_ksq, P->d_kperm, P->d_kpmeta, (int32_t)np);
        (void)hipMemcpyAsync(d_q_final, P->d_full, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream);
    }
    if (rc == RR_OK && io.uh_kernel && uh_state_inout) {      // carry-over state of the fused convolution, in place, after every batch has read the old one
        const dim3 gt((unsigned)((n + kUhTailThreads - 1) / kUhTailThreads));
        const int32_t nks = (int32_t)io.uh_nks;
        if (io.dev_in32) {      // float32 depth rows
            const uint32_t sel = P->in32_big_endian ? kSelSwap : kSelNative;
            if (nks <= 16) hipLaunchKernelGGL((k_uh_tail<16, float>), gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in32, T, nks, n, sel);
            else if (nks <= 48) hipLaunchKernelGGL((k_uh_tail<48, float>), gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in32, T, nks, n, sel);
            else hipLaunchKernelGGL((k_uh_tail<0, float>), gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in32, T, nks, n, sel);
        }
        else if (nks <= 16) hipLaunchKernelGGL(k_uh_tail<16>, gt, dim3(kUhTailThreads), 0, stream, io.uh_kernel, uh_state_inout, io.dev_in, T, nks, n);
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

template <typename TIn>
int uh_convolve_core(const double *d_kernel, double *d_state, const TIn *d_lateral, double *d_out, int64_t T,
                     int64_t n_ks, int64_t n, hipStream_t stream, uint32_t sel)
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
            (void)hipFuncSetAttribute((const void *)k_uh_convolve_ring<NK_, NT_, R_, D_, TIn>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_uh_convolve_ring<NK_, NT_, R_, D_, TIn>), g, dim3(kUhThreads), lds, stream, d_kernel, (const double *)d_state,     \
                               d_lateral, d_out, T, (int32_t)n_ks, n, seg_rows, sel);                              \
        } while (0)
        if (n_ks <= 5) RR_UH_LAUNCH(8, 5, 4, 2);            // NK >= NT + R - 1 window slots
        else if (n_ks <= 13) RR_UH_LAUNCH(16, 13, 4, 2);
        else if (n_ks <= 24) RR_UH_LAUNCH(32, 24, 8, 2);
        else if (n_ks <= 29) RR_UH_LAUNCH(32, 29, 4, 2);
        else if (n_ks <= 48) RR_UH_LAUNCH(64, 48, 8, 2);
        else RR_UH_LAUNCH(64, 57, 8, 2);
#undef RR_UH_LAUNCH
    } else {
        dim3 g((unsigned)((n + kBlock - 1) / kBlock), (unsigned)((T + TB - 1) / TB));
        hipLaunchKernelGGL((k_uh_convolve<TB, TIn>), g, dim3(kBlock), 0, stream, d_kernel, (const double *)d_state, d_lateral,
                           d_out, T, (int32_t)n_ks, n, sel);
    }
    // carry-over state, in place, after the rows above have read the old one (same stream)
    const dim3 gt((unsigned)((n + kUhTailThreads - 1) / kUhTailThreads));
    if (n_ks <= 16) hipLaunchKernelGGL((k_uh_tail<16, TIn>), gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n, sel);
    else if (n_ks <= 48) hipLaunchKernelGGL((k_uh_tail<48, TIn>), gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n, sel);
    else hipLaunchKernelGGL((k_uh_tail<0, TIn>), gt, dim3(kUhTailThreads), 0, stream, d_kernel, d_state, d_lateral, T, (int32_t)n_ks, n, sel);
    HIPCHK(hipGetLastError());
    return RR_OK;
}

}  // namespace
