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
//
// One translation unit, in this order: rr_common.hpp (constants, index helpers), rr_kernels_tick.hpp, rr_kernels_tile.hpp,
// rr_kernels_uh.hpp, rr_kernels_rec.hpp, rr_kernels_runoff.hpp, rr_kernels_direct.hpp (device code), rr_exec.hpp (plan object, executor), then the
// C ABI below.
#include "rr_common.hpp"
#include "rr_kernels_tick.hpp"
#include "rr_kernels_tile.hpp"
#include "rr_kernels_uh.hpp"
#include "rr_kernels_rec.hpp"
#include "rr_kernels_runoff.hpp"
#include "rr_kernels_direct.hpp"
#include "rr_exec.hpp"

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
                        P->d_coef, P->d_coef_unit, P->d_sq, P->d_ss, P->d_si, P->d_sqch, P->d_full, P->d_chan,
                        P->d_tmeta, P->d_pmeta,
                        P->d_tperm, P->d_tinv, P->d_inner_idx, P->d_colmeta, P->d_ghostmeta, P->d_c4_params,
                        P->d_c3, P->d_c4, P->d_x, P->d_isum, P->d_qch, P->d_a2, P->d_c1own, P->d_z, P->d_ring, P->d_stage, P->d_mrows,
                        P->d_slot_a[0], P->d_slot_a[1], P->d_slot_b[0], P->d_slot_b[1], P->d_m_index[0], P->d_m_index[1],
                        P->d_dtiles, P->d_dlane, P->d_dsend_ptr, P->d_dsend_lane, P->d_dcoef, P->d_dq, P->d_ktmeta, P->d_kpmeta, P->d_kperm, P->d_kholecol, P->d_kcoef, P->d_ksq, P->d_kss, P->d_ksi, P->d_ksqch,
                        P->d_kholemeta, P->d_kghostmeta};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        P->pipe.destroy();
        for (hipEvent_t e : P->ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : P->aux_ev) (void)hipEventDestroy(e);
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
    // Run-time switches, all for tests (each is exercised by tests/test_gpu_*.py): RR_WAVE=0 the streaming kernel for every call,
    // =1 the time-tiled one wherever it applies; RR_WAVE_K ticks per task; RR_TILE_BLOCK tile capacity (many small tiles);
    // RR_TILE_LEAN=0 the general tick; RR_UH_PAIRS=0 one record batch per fused-convolution launch; RR_DIRECT=0 records also where the
    // params order would allow the direct row path; RR_VERBOSE=1 logs the schedule.
    if (const char *e = getenv("RR_WAVE")) { P->wave_enabled = atoi(e) != 0; P->wave_forced = atoi(e) == 1; }
    if (const char *e = getenv("RR_WAVE_K")) P->wave_K = std::max(kRec, atoi(e) / kRec * kRec);
    std::string err;
    int rc = rr::build_host_plan(n, csc_indptr, csc_indices, P->h, err);
    if (rc) { delete P; return fail(rc, err); }
    if (device != RR_DEVICE_NONE) {      // the tile size below depends on the card's CU count
        int count = rr_device_count(), cus = 0;
        if (device >= 0 && device < count && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) P->cu_count = cus;
    }
    {   // Time-tiled schedule: one position per thread, tiles of at most kTileThreads positions (rr_exec.hpp: tile_kernel)
        P->wave_threads = kTileThreads;
        int32_t block = P->wave_threads;
        if (const char *e = getenv("RR_TILE_BLOCK")) block = std::max(8, std::min(block, atoi(e)));     // tests: many small tiles
        std::vector<int32_t> lag_of((size_t)n);
        for (int64_t i = 0; i < n; ++i) lag_of[i] = P->h.lag[P->h.inv[i]];
        rr::build_tile_plan(P->h.down, lag_of, block, P->tp);
        // the direct row path where the params order numbers small subtrees contiguously (any depth-first post-order); `why` says why not
        if (const char *e = getenv("RR_DIRECT")) P->direct_enabled = atoi(e) != 0;
        P->direct_block = block;
        rr::build_direct_plan(P->h.down, lag_of, std::min<int32_t>(kDirectLanes, block), kDirectMaxWindow - 2, block, P->dp);
        P->direct_window = 3;
        for (int32_t sp : P->dp.tile_span) P->direct_window = std::max(P->direct_window, sp + 3);
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
        }
        if (const char *e2 = getenv("RR_TILE_LEAN")) P->lean_enabled = atoi(e2) != 0;
        for (int v = 0; v < 7; ++v)    // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (v = 4, 5, 6: the short ticks of Rapid, Unit, channel-only)
            if (hipFuncSetAttribute((const void *)tile_kernel(v < 4 ? (v & 1) != 0 : v == 5, v < 4 && (v & 2) != 0, v >= 4, v == 6), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)tile_lds_bytes(P->wave_threads)) != hipSuccess) {
                (void)hipGetLastError();
                P->wave_enabled = false;
            }
        if (const char *e2 = getenv("RR_UH_PAIRS")) P->uh_pairs = atoi(e2) != 0;
        for (int v = 0; v < 4; ++v)
            (void)hipFuncSetAttribute((const void *)rec_in_uh_kernel((v & 1) != 0, 64, 1 + (v >> 1)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rec_in_uh_lds_bytes(64, 1 + (v >> 1)));
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
            for (int64_t i = 0; i < n; ++i)      // a headwater column is flagged: UnitMuskingum's out-pass leaves it unclamped
                cm[i] = make_int2(TP.inv[i], (TP.lag[TP.inv[i]] & kLagMask) | (H.child_ptr[H.inv[i] + 1] == H.child_ptr[H.inv[i]] ? kColHeadwater : 0));
            std::vector<int32_t> inner_idx;
            inner_idx.reserve(ni);
            for (int64_t i = 0; i < n; ++i) if (H.child_ptr[H.inv[i] + 1] > H.child_ptr[H.inv[i]]) inner_idx.push_back((int32_t)i);
            rc = dev_alloc(&P->d_colmeta, n);
            if (!rc) rc = dev_upload(P->d_colmeta, cm);
            if (!rc) rc = dev_alloc(&P->d_inner_idx, ni);
            if (!rc) rc = dev_upload(P->d_inner_idx, inner_idx);
            if (!rc) rc = upload_tile_meta(P, TP.lag, TP.xpos, TP.tile_flags);
            for (int32_t f : TP.tile_flags) P->n_wide_tiles += (f & kTileWide) ? 1 : 0;
            if (!rc) rc = dev_alloc(&P->d_tperm, np);
            if (!rc) rc = dev_upload(P->d_tperm, TP.perm);
            if (!rc) rc = dev_alloc(&P->d_tinv, n);
            if (!rc) rc = dev_upload(P->d_tinv, TP.inv);
            if (!rc) rc = dev_alloc(&P->d_coef, 3 * np);
            if (!rc) rc = dev_alloc(&P->d_coef_unit, 3 * np);
            if (!rc) rc = dev_alloc(&P->d_sq, np);
            if (!rc) rc = dev_alloc(&P->d_ss, np);
            if (!rc) rc = dev_alloc(&P->d_si, np);
            if (!rc) rc = dev_alloc(&P->d_sqch, np);
            if (!rc) rc = dev_alloc(&P->d_full, n);
            if (!rc) rc = dev_alloc(&P->d_chan, n);
        }
        if (!rc) rc = upload_direct_plan(P);      // direct row path: per-column constants, the skeleton's tile arrays, the holes' out-pass
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
    info[6] = P->wave_threads; info[7] = kRecRows;
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

int rr_plan_direct_info(const rr_plan *P, int64_t info[8], char *why, int64_t why_cap)
{
    if (!P || !info) return fail(RR_E_INVALID, "rr_plan_direct_info: null argument");
    const rr::DirectPlan &D = P->dp;
    info[0] = D.ok ? 1 : 0; info[1] = D.n_tiles; info[2] = D.n_holes; info[3] = D.n_exports; info[4] = D.skel.np; info[5] = D.skel.n_tiles;
    info[6] = D.skel.n_levels; info[7] = P->direct_window;
    if (why && why_cap > 0) { std::strncpy(why, D.why.c_str(), (size_t)why_cap - 1); why[why_cap - 1] = 0; }
    return RR_OK;
}

int rr_plan_direct_layout(const rr_plan *P, int32_t *tile_c0, int32_t *tile_nc, int32_t *tile_lag_lo, int32_t *tile_span, int32_t *delay, int32_t *up3, int32_t *xinfo)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_direct_layout: null plan");
    const rr::DirectPlan &D = P->dp;
    if (!D.ok) return fail(RR_E_UNSUPPORTED, "rr_plan_direct_layout: the direct row path does not apply to this plan: " + D.why);
    auto copy = [](auto *dst, const auto &src) { if (dst && !src.empty()) std::memcpy(dst, src.data(), src.size() * sizeof(src[0])); };
    copy(tile_c0, D.tile_c0); copy(tile_nc, D.tile_nc); copy(tile_lag_lo, D.tile_lag_lo); copy(tile_span, D.tile_span);
    copy(delay, D.delay); copy(up3, D.up3); copy(xinfo, D.xinfo);
    return RR_OK;
}

int rr_plan_last_kernel(const rr_plan *P)
{
    if (!P) return -1;
    return P->ses.rows_direct ? RR_KERNEL_DIRECT : (P->ses.wave ? RR_KERNEL_TILE : RR_KERNEL_TICK);
}

int rr_plan_set_options(rr_plan *P, int64_t rows_per_chunk, int64_t sample_every)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_set_options: null plan");
    if (rows_per_chunk > 0) P->chunk_rows = rows_per_chunk;
    if (sample_every >= 0) P->sample_every = sample_every;
    return RR_OK;
}

int rr_plan_set_row_format(rr_plan *P, int in32_big_endian, int out32_big_endian)
{
    if (!P) return fail(RR_E_INVALID, "rr_plan_set_row_format: null plan");
    if (P->ses.open) return fail(RR_E_STATE, "rr_plan_set_row_format: a routing call is open");
    P->in32_big_endian = in32_big_endian != 0; P->out32_big_endian = out32_big_endian != 0;
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
        P->h_coef.assign(3 * TP.np, 0.0);
        for (int64_t p = 0; p < TP.np; ++p) {
            if (TP.lag[p] & kTileGhostBit) continue;
            const int32_t i = TP.perm[p];
            P->h_coef[3 * p] = c1row[H.inv[i]]; P->h_coef[3 * p + 1] = c2[i]; P->h_coef[3 * p + 2] = c3[i];
        }
        rc = upload_tile_coef(P);
    }
    {   // direct row path: {c1row, c2, c3, c4dt} per column, kept on the host too (rr_plan_set_boundary lays the direct plan out again)
        P->h_dcoef.assign(4 * (size_t)n, 0.0);
        for (int64_t i = 0; i < n; ++i) { P->h_dcoef[4 * i] = c1row[H.inv[i]]; P->h_dcoef[4 * i + 1] = c2[i]; P->h_dcoef[4 * i + 2] = c3[i]; P->h_dcoef[4 * i + 3] = c4_dt ? c4_dt[i] : 0.0; }
        if (!rc) rc = upload_direct_coef(P);
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

int rr_plan_set_unit_weights(rr_plan *P, const double *c1, const double *a_data)
{
    int rc = need_device(P);
    if (rc) return rc;
    if (P->ses.open) return fail(RR_E_STATE, "rr_plan_set_unit_weights: a routing call is open");
    if (!c1 && !a_data) { P->unit_general = false; return RR_OK; }
    const rr::HostPlan &H = P->h;
    const int64_t n = H.n;
    if (!c1 || (H.n_edges > 0 && !a_data)) return fail(RR_E_INVALID, "rr_plan_set_unit_weights: give both arrays, or neither to go back to unit weights");
    if (P->n_ghost > 0 || P->n_export > 0) return fail(RR_E_UNSUPPORTED, "rr_plan_set_unit_weights: general edge data on a partitioned network is not supported");
    std::vector<double> a2(n, 0.0), own(n, 0.0);
    for (int64_t p = 0; p < n; ++p) {
        const int32_t i = H.perm[p], e = H.edge_of[i];
        a2[p] = e >= 0 ? a_data[e] : 0.0;
        own[p] = c1[i];
    }
    if (!P->d_a2) rc = dev_alloc(&P->d_a2, n);
    if (!rc && !P->d_c1own) rc = dev_alloc(&P->d_c1own, n);
    if (!rc && !P->d_z) rc = dev_alloc(&P->d_z, 3 * n);
    if (!rc) rc = dev_upload(P->d_a2, a2);
    if (!rc) rc = dev_upload(P->d_c1own, own);
    if (rc) return rc;
    P->unit_general = true;
    return RR_OK;
}

int rr_plan_reserve(rr_plan *P, int mode, int64_t T, int64_t nsub, int host_rows, int64_t info[8])
{
    int rc = need_device(P);
    if (rc) return rc;
    if (mode < RR_MODE_RAPID || mode > RR_MODE_UNIT || T < 0 || nsub < 1) return fail(RR_E_INVALID, "rr_plan_reserve: unknown mode, negative step count or sub-steps < 1");
    if (!P->coeffs_set) return fail(RR_E_STATE, "rr_plan_reserve before rr_plan_set_coeffs (per-edge weights decide which kernel routes)");
    if (P->ses.open) return fail(RR_E_STATE, "rr_plan_reserve: a routing call is open");
    const Mode m = mode == RR_MODE_RAPID ? Mode::Rapid : (mode == RR_MODE_MUSKINGUM ? Mode::Muskingum : Mode::Unit);
    Schedule sch;
    // host rows reach the time-tiled kernel through the PCIe pipeline's device rings; where it does not apply they stream
    const bool host = (host_rows & 1) != 0, plain = !host && (host_rows & RR_ROWS_NOT_PLAIN) == 0;      // rows in device arrays (float64 or float32): the direct row path applies
    rc = reserve_core(P, m, T, nsub, false, false, &sch, plain, (host_rows & RR_ROWS_F32_OUT) != 0, (host_rows & RR_ROWS_UH) != 0);
    if (rc == RR_OK && host && !sch.tiled) rc = reserve_core(P, m, T, nsub, true, true, &sch);
    if (rc == RR_OK && host && sch.tiled) rc = host_pipe_prepare(P);
    if (rc) return rc;
    if (info) {
        info[0] = sch.direct ? 2 : (sch.tiled ? 1 : 0); info[1] = (sch.tiled || sch.direct) ? sch.KC * kRec : 1; info[2] = sch.chunks;
        info[3] = (P->ring_cap + P->mrows_cap + P->stage_cap) * (int64_t)sizeof(double);
        info[4] = host && sch.tiled ? 2 * P->pipe.dev_cap * (int64_t)sizeof(double) : 0;
        info[5] = host && sch.tiled ? 2 * HostPipe::kPinned * P->pipe.pin_cap * (int64_t)sizeof(double) : 0;
        info[6] = sch.direct ? P->h.depth - 1 + (int64_t)P->dp.skel.n_levels * sch.KC * kRec
                             : (sch.tiled ? P->h.depth - 1 + (int64_t)P->tp.n_levels * sch.KC * kRec : P->h.depth - 1);      // pipeline depth in ticks
        info[7] = sch.ring * (int64_t)sizeof(double);
    }
    return RR_OK;
}

int rr_plan_profile(rr_plan *P, double prof[10])
{
    if (!P || !prof) return fail(RR_E_INVALID, "rr_plan_profile: null argument");
    for (int k = 0; k < 10; ++k) prof[k] = 0.0;
    prof[0] = (double)P->prof_launches;
    prof[8] = (double)P->prof_brackets;
    prof[9] = (P->ses.wave || P->ses.rows_direct) ? (double)(P->ses.KC * kRec) : 1.0;
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

int rr_plan_profile_aux(rr_plan *P, double aux[12])
{
    if (!P || !aux) return fail(RR_E_INVALID, "rr_plan_profile_aux: null argument");
    for (int k = 0; k < 12; ++k) aux[k] = 0.0;
    for (int k = 0; k < rr_plan::kAuxKinds; ++k) aux[3 * k] = (double)P->aux_launches[k];
    if (P->device < 0 || P->aux_kind.empty()) return RR_OK;
    HIPCHK(hipSetDevice(P->device));
    if (P->ev_last) HIPCHK(hipEventSynchronize(P->ev_last));
    for (size_t i = 0; i < P->aux_kind.size(); ++i) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(P->aux_ev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms, P->aux_ev[2 * i], P->aux_ev[2 * i + 1]));
        aux[3 * P->aux_kind[i] + 1] += 1.0;
        aux[3 * P->aux_kind[i] + 2] += ms;
    }
    return RR_OK;
}

// ---- partitioned networks: boundary reaches + streaming calls ----

int rr_plan_set_boundary(rr_plan *P, int64_t n_ghost, const int64_t *ghost_reaches, int64_t n_export,
                         const int64_t *export_reaches)
{
    if (!P) return fail(RR_E_INVALID, "null plan");
    const bool host_only = P->device < 0;      // a host-only plan takes the boundary too: its layouts can be inspected (rr_plan_direct_info)
    int rc = host_only ? RR_OK : need_device(P);
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
    if (!host_only) {
        rc = dev_upload(P->d_lag, lag);
        if (!rc) rc = dev_upload(P->d_bidx, bidx);
        if (rc) return rc;
    }
    P->n_ghost = n_ghost; P->n_export = n_export;
    P->ghost_min_lag = n_ghost ? gmin : 0;
    P->export_max_lag = n_export ? emax : 0;
    P->ghost_reach.assign(n_ghost, 0); P->export_reach.assign(n_export, 0);
    for (int64_t g = 0; g < n_ghost; ++g) P->ghost_reach[g] = (int32_t)ghost_reaches[g];
    for (int64_t e = 0; e < n_export; ++e) P->export_reach[e] = (int32_t)export_reaches[e];
    {   // the direct row path, laid out again around the boundary reaches: a ghost's column is passed through (its series becomes the record
        // of the skeleton's ghost that mirrors it), an export is stored by the skeleton's kernel or by its lane (rr_plan.hpp: build_direct_plan)
        std::vector<uint8_t> gmask((size_t)n, 0);
        std::vector<int32_t> eslot((size_t)n, -1), lag_of((size_t)n);
        for (int64_t g = 0; g < n_ghost; ++g) gmask[ghost_reaches[g]] = 1;
        for (int64_t e = 0; e < n_export; ++e) eslot[export_reaches[e]] = (int32_t)e;
        for (int64_t i = 0; i < n; ++i) lag_of[i] = H.lag[H.inv[i]];
        rr::build_direct_plan(H.down, lag_of, std::min<int32_t>(kDirectLanes, P->direct_block), kDirectMaxWindow - 2, P->direct_block, P->dp,
                              n_ghost ? &gmask : nullptr, n_export ? &eslot : nullptr);
        rc = upload_direct_plan(P);
        if (!rc) rc = upload_direct_coef(P);
        if (rc) return rc;
    }
    if (host_only) return RR_OK;
    if (P->tp.ok) {      // the same flags and slots in the tile layout
        const rr::TilePlan &TP = P->tp;
        // flags in the tile layout; an export reach's slot in the export series travels in xpos[], the word a reach mirrored
        // by another tile's ghost uses for that ghost's position -- an export is normally an outlet of its part and has no
        // such ghost; where it has one, the plan keeps to the streaming kernel
        std::vector<int32_t> tlag(TP.lag), txpos(TP.xpos);
        P->export_inside = false;
        for (int64_t g = 0; g < n_ghost; ++g) { const int32_t p = TP.inv[ghost_reaches[g]]; tlag[p] |= kGhostBit; }
        for (int64_t e = 0; e < n_export; ++e) {
            const int32_t p = TP.inv[export_reaches[e]];
            if (tlag[p] & kTileExportBit) P->export_inside = true;
            tlag[p] |= kExportBit; txpos[p] = (int32_t)e;
        }
        std::vector<int32_t> tflags(TP.tile_flags);      // a tile with a boundary export stores it tick by tick
        for (int64_t e = 0; e < n_export; ++e) tflags[TP.tile_of[TP.inv[export_reaches[e]]]] |= kTileExports;
        rc = upload_tile_meta(P, tlag, P->export_inside ? TP.xpos : txpos, tflags);
        if (!rc) rc = upload_tile_coef(P);      // zeros at the boundary ghosts
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
    rc = prepare_call(P, mode, T, nsub, false, false, true, true, has_lateral ? lat_rows : 0, out_rows);
    if (rc) return rc;
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
    if (P->unit_general) return fail(RR_E_UNSUPPORTED, "rr_stream_begin_unit: general edge data (rr_plan_set_unit_weights) on a partitioned network is not supported");
    rc = prepare_call(P, Mode::Unit, T, nsub, false, false, true);
    if (rc) return rc;
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

int rr_postorder(int64_t n, const int64_t *down_index, int64_t *order)
{
    if (n < 0 || (n > 0 && (!down_index || !order))) return fail(RR_E_INVALID, "rr_postorder: bad argument");
    if (!rr::postorder(down_index, n, order)) return fail(RR_E_INVALID, "rr_postorder: a downstream index is out of range, or the network has a cycle");
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
// the tick-rows of a batch (128); otherwise RR_E_UNSUPPORTED and the caller uses the float64 form + rr_resample_cast_dev.
static int f32_output_applies(rr_plan *P, Mode mode, int64_t T, int64_t nsub, int64_t factor, bool plain = false, bool uh = false)
{
    if (factor < 1 || T % factor != 0) return fail(RR_E_INVALID, "float32 output: the number of rows must be a multiple of factor >= 1");
    if (kRecRows % (factor * nsub) != 0) return fail(RR_E_UNSUPPORTED, "float32 output: factor x sub-steps must divide the rows of a record batch (128)");
    const Schedule sch = choose_schedule(P, mode, T, nsub, false, false, plain, 0, 0, true, uh);
    if (!sch.tiled && !sch.direct) return fail(RR_E_UNSUPPORTED, "float32 output needs the time-tiled kernel or the direct row path, which this call does not get");
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
    rc = f32_output_applies(P, Mode::Rapid, T, nsub, factor, true);
    if (rc) return rc;
    Rows io; io.dev_in = qlateral; io.rows_in = ql_rows; io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor;
    return rapid_like(P, Mode::Rapid, q_t, io, T, nsub, (hipStream_t)stream, false);
}

int rr_rapid_route_f32in_dev(rr_plan *P, double *q_t, const float *qlateral32, int64_t ql_rows, double *discharge, int64_t out_rows,
                             float *discharge32, int64_t factor, int64_t T, int64_t nsub, void *stream)
{
    int rc = check_route_args(P, true, T, nsub);
    if (rc) return rc;
    const bool f32 = discharge32 != nullptr;
    if (P->h.n > 0 && T > 0 && (!q_t || !qlateral32 || ql_rows < 1 || (!discharge && !discharge32) || (discharge && discharge32) || (discharge && out_rows < 1)))
        return fail(RR_E_INVALID, "rr_rapid_route_f32in_dev: null array, empty row count, or both or neither output");
    if (P->h.n == 0 || T == 0) return RR_OK;
    if (f32) { rc = f32_output_applies(P, Mode::Rapid, T, nsub, factor, true); if (rc) return rc; }
    else { const Schedule sch = choose_schedule(P, Mode::Rapid, T, nsub, false, false, true); if (!sch.tiled && !sch.direct) return fail(RR_E_UNSUPPORTED, "rr_rapid_route_f32in_dev needs the time-tiled kernel or the direct row path, which this call does not get"); }
    Rows io; io.dev_in32 = qlateral32; io.dev_in = P->d_c4_params; io.rows_in = ql_rows;      // (dev_in only has to be non-NULL for the executor)
    if (f32) { io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor; }
    else { io.dev_out = discharge; io.rows_out = out_rows; }
    return rapid_like(P, Mode::Rapid, q_t, io, T, nsub, (hipStream_t)stream, false);
}

int rr_muskingum_route_f32_dev(rr_plan *P, double *q_t, float *discharge32, int64_t n_out, int64_t n_per_out, void *stream)
{
    int rc = check_route_args(P, false, n_out, n_per_out);
    if (rc) return rc;
    if (P->h.n > 0 && n_out > 0 && (!q_t || !discharge32)) return fail(RR_E_INVALID, "rr_muskingum_route_f32_dev: null array");
    if (P->h.n == 0 || n_out == 0) return RR_OK;
    rc = f32_output_applies(P, Mode::Muskingum, n_out, n_per_out, 1, true);
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
    rc = f32_output_applies(P, Mode::Unit, T, nsub, factor, true);
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
    else if (!choose_schedule(P, Mode::Rapid, T, 1, false, false).tiled) return fail(RR_E_UNSUPPORTED, "rr_rapid_route_runoff_dev needs the time-tiled kernel, which this call does not get");
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
    if (f32) { rc = f32_output_applies(P, Mode::Unit, T, nsub, factor, true, true); if (rc) return rc; }
    else { const Schedule sch = choose_schedule(P, Mode::Unit, T, nsub, false, false, true, 0, 0, false, true); if (!sch.tiled && !sch.direct) return fail(RR_E_UNSUPPORTED, "rr_unit_route_uh_dev needs the time-tiled kernel or the direct row path, which this call does not get"); }
    Rows io; io.dev_in = depth; io.rows_in = T;
    if (f32) { io.dev_out32 = discharge32; io.out_factor = factor; io.rows_out = T / factor; }
    else { io.dev_out = discharge; io.rows_out = T; }
    io.uh_kernel = uh_kernel; io.uh_state = uh_state; io.uh_nks = n_ks;
    return unit_like(P, q_ch, q_full, io, T, nsub, (hipStream_t)stream, false, q_final, uh_state);
}

int rr_unit_route_uh_f32in_dev(rr_plan *P, double *q_ch, double *q_full, double *q_final, const double *uh_kernel, double *uh_state,
                               int64_t n_ks, const float *depth32, double *discharge, float *discharge32, int64_t factor, int64_t T,
                               int64_t nsub, void *stream)
{
    int rc = check_route_args(P, false, T, nsub);
    if (rc) return rc;
    const bool f32 = discharge32 != nullptr;
    if (P->h.n > 0 && T > 0 && (!depth32 || !uh_kernel || !uh_state || (!discharge && !discharge32) || (discharge && discharge32) || n_ks < 1 ||
                                (!P->h.inner_pos.empty() && (!q_ch || !q_full))))
        return fail(RR_E_INVALID, "rr_unit_route_uh_f32in_dev: null array, both or neither output, or n_ks < 1");
    if (P->h.n == 0 || T == 0) return RR_OK;
    if (n_ks > kUhFusedMaxTaps) return fail(RR_E_UNSUPPORTED, "rr_unit_route_uh_f32in_dev: more than 64 kernel steps: convert the rows, convolve with rr_uh_convolve_dev, then rr_unit_route_dev");
    if (f32) { rc = f32_output_applies(P, Mode::Unit, T, nsub, factor, true, true); if (rc) return rc; }
    else { const Schedule sch = choose_schedule(P, Mode::Unit, T, nsub, false, false, true, 0, 0, false, true); if (!sch.tiled && !sch.direct) return fail(RR_E_UNSUPPORTED, "rr_unit_route_uh_f32in_dev needs the time-tiled kernel or the direct row path, which this call does not get"); }
    Rows io; io.dev_in32 = depth32; io.dev_in = uh_kernel; io.rows_in = T;      // (dev_in only has to be non-NULL for the executor)
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
    return uh_convolve_core<double>(kernel, state, lateral, out, T, n_ks, n, (hipStream_t)stream, kSelNative);
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
    if (!rc) rc = uh_convolve_core<double>(d_k, d_s, d_l, d_o, T, n_ks, n, nullptr, kSelNative);
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
        const dim3 g((unsigned)((count + kBlock - 1) / kBlock));      // one element per thread (count <= 2^31 x 256 elements)
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

// ---- rows between a file and the device (the routers' qlateral / discharge files) ----
//
// A (rows, row_bytes) block that lies in a file at `file_offset` with `file_pitch` bytes from row to row (NetCDF-3: a fixed variable is
// contiguous, a record variable has the other record variables of each record between its rows) goes to / comes from a device array without
// ever existing as a host array: reader threads pread() whole rows into three pinned chunks while the copy engine uploads the chunk
// before (downloads: the copy engine fills a chunk while writer threads pwrite() the one before).  Byte order is the device's business
// (rr_plan_set_row_format).  Returns when the block has moved.
namespace {
struct RowPipe {
    static constexpr int kChunks = 3, kThreads = 4;
    char *pin[kChunks] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[kChunks] = {nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    int64_t cap = 0;
    ~RowPipe()
    {
        for (int k = 0; k < kChunks; ++k) { if (pin[k]) (void)hipHostFree(pin[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
        if (st) (void)hipStreamDestroy(st);
    }
    int prepare(int64_t bytes)
    {
        if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int k = 0; k < kChunks; ++k) if (!ev[k]) HIPCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        if (cap >= bytes) return RR_OK;
        for (int k = 0; k < kChunks; ++k) {
            if (pin[k]) { (void)hipHostFree(pin[k]); pin[k] = nullptr; }
            if (hipHostMalloc((void **)&pin[k], (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); cap = 0; return fail(RR_E_ALLOC, "rr_rows_*: pinned staging could not be allocated"); }
        }
        cap = bytes;
        return RR_OK;
    }
};
thread_local RowPipe g_row_pipe;

// rows [r0, r1) of the file block <-> packed rows in `buf`, split over the threads; false on a short read / write
bool file_rows(int fd, bool write, char *buf, int64_t file_offset, int64_t file_pitch, int64_t row_bytes, int64_t r0, int64_t r1)
{
    const int64_t rows = r1 - r0;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(RowPipe::kThreads, rows));
    std::vector<std::thread> pool;
    std::vector<int> ok((size_t)nt, 1);
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            for (int64_t r = r0 + rows * t / nt; r < r0 + rows * (t + 1) / nt; ++r) {
                char *p = buf + (r - r0) * row_bytes;
                int64_t done = 0;
                while (done < row_bytes) {
                    const ssize_t got = write ? pwrite(fd, p + done, (size_t)(row_bytes - done), (off_t)(file_offset + r * file_pitch + done))
                                              : pread(fd, p + done, (size_t)(row_bytes - done), (off_t)(file_offset + r * file_pitch + done));
                    if (got <= 0) { ok[t] = 0; return; }
                    done += got;
                }
            }
        });
    for (auto &th : pool) th.join();
    for (int v : ok) if (!v) return false;
    return true;
}
}  // namespace

static int rows_transfer(int device, bool upload, void *dev, int64_t dev_pitch, const char *path, int64_t file_offset, int64_t file_pitch, int64_t row_bytes,
                         int64_t n_rows, void *stream)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_rows_*: no such HIP device");
    if (!dev || !path || row_bytes < 1 || n_rows < 0 || dev_pitch < row_bytes || file_pitch < row_bytes || file_offset < 0) return fail(RR_E_INVALID, "rr_rows_*: bad argument");
    if (n_rows == 0) return RR_OK;
    HIPCHK(hipSetDevice(device));
    const int fd = open(path, upload ? O_RDONLY : O_WRONLY);
    if (fd < 0) return fail(RR_E_INVALID, std::string("rr_rows_*: cannot open ") + path);
    RowPipe &R = g_row_pipe;
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t{64} << 20) / row_bytes);      // 64 MB: long enough for the copy engine, short enough to overlap
    int rc = R.prepare(chunk_rows * row_bytes);
    const int64_t n_chunks = (n_rows + chunk_rows - 1) / chunk_rows;
    hipError_t e = hipSuccess;
    if (!rc) {      // the block is used by / comes from work on the caller's stream (NULL: the default stream)
        hipEvent_t fence = nullptr;
        e = hipEventCreateWithFlags(&fence, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(fence, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(R.st, fence, 0);
        if (fence) (void)hipEventDestroy(fence);
    }
    bool io_ok = true;
    if (!rc && e == hipSuccess) {
        if (upload) {
            for (int64_t c = 0; c < n_chunks && io_ok && e == hipSuccess; ++c) {
                const int k = (int)(c % RowPipe::kChunks);
                const int64_t r0 = c * chunk_rows, r1 = std::min(n_rows, r0 + chunk_rows);
                if (c >= RowPipe::kChunks) e = hipEventSynchronize(R.ev[k]);      // the chunk that used this buffer has left
                if (e != hipSuccess) break;
                io_ok = file_rows(fd, false, R.pin[k], file_offset, file_pitch, row_bytes, r0, r1);
                if (!io_ok) break;
                e = hipMemcpy2DAsync((char *)dev + r0 * dev_pitch, (size_t)dev_pitch, R.pin[k], (size_t)row_bytes, (size_t)row_bytes, (size_t)(r1 - r0), hipMemcpyHostToDevice, R.st);
                if (e == hipSuccess) e = hipEventRecord(R.ev[k], R.st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(R.st);
        } else {
            // chunk c is downloaded while chunk c - 1 is written
            auto fetch = [&](int64_t c) {
                const int k = (int)(c % RowPipe::kChunks);
                const int64_t r0 = c * chunk_rows, r1 = std::min(n_rows, r0 + chunk_rows);
                hipError_t ee = hipMemcpy2DAsync(R.pin[k], (size_t)row_bytes, (const char *)dev + r0 * dev_pitch, (size_t)dev_pitch, (size_t)row_bytes, (size_t)(r1 - r0), hipMemcpyDeviceToHost, R.st);
                if (ee == hipSuccess) ee = hipEventRecord(R.ev[k], R.st);
                return ee;
            };
            e = fetch(0);
            for (int64_t c = 0; c < n_chunks && io_ok && e == hipSuccess; ++c) {
                const int k = (int)(c % RowPipe::kChunks);
                if (c + 1 < n_chunks) e = fetch(c + 1);      // (buffer (c + 1) % 3 was written out two iterations ago)
                if (e == hipSuccess) e = hipEventSynchronize(R.ev[k]);
                if (e != hipSuccess) break;
                const int64_t r0 = c * chunk_rows, r1 = std::min(n_rows, r0 + chunk_rows);
                io_ok = file_rows(fd, true, R.pin[k], file_offset, file_pitch, row_bytes, r0, r1);
            }
        }
    }
    (void)close(fd);
    if (R.st && (rc || e != hipSuccess || !io_ok)) (void)hipStreamSynchronize(R.st);      // a failed transfer leaves no copy in flight behind: the caller may free the rows
    if (rc) return rc;
    if (e != hipSuccess) return fail(RR_E_HIP, std::string("rr_rows_*: ") + hipGetErrorString(e));
    if (!io_ok) return fail(RR_E_INVALID, std::string("rr_rows_*: short read or write on ") + path);
    return RR_OK;
}

int rr_rows_upload(int device, void *dst_dev, int64_t dst_pitch, const char *path, int64_t file_offset, int64_t file_pitch, int64_t row_bytes, int64_t n_rows, void *stream)
{
    return rows_transfer(device, true, dst_dev, dst_pitch, path, file_offset, file_pitch, row_bytes, n_rows, stream);
}

int rr_rows_download(int device, const void *src_dev, int64_t src_pitch, const char *path, int64_t file_offset, int64_t file_pitch, int64_t row_bytes, int64_t n_rows, void *stream)
{
    return rows_transfer(device, false, const_cast<void *>(src_dev), src_pitch, path, file_offset, file_pitch, row_bytes, n_rows, stream);
}

int rr_dev_synchronize(int device)
{
    if (device < 0 || device >= rr_device_count()) return fail(RR_E_NO_DEVICE, "rr_dev_synchronize: no such HIP device");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return RR_OK;
}

}  // extern "C"
