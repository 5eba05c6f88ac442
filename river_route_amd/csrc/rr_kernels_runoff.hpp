// rr_kernels_runoff.hpp -- gridded runoff -> catchment inflow (river_route/runoff.py:288-330), stand-alone and fused into the record in-pass; the copy-rate probe.
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

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
    for (int j = 1; j <= kRunoffRows; ++j) {      // (no break: the loop must unroll for acc[] to stay in registers -- it was 144 B of scratch per thread)
        double v = (cumulative && t0 + j - 1 > 0) ? acc[j] - acc[j - 1] : acc[j];
        if (force_positive) v = v < 0.0 ? 0.0 : v;      // np.clip leaves NaN alone, as does this comparison
        if (v != v && !keep_nan) v = 0.0;
        if (j <= nt) out[(t0 + j - 1) * n_rivers + r] = area ? v * a : v;
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
    __shared__ int64_t slot[kRunoffInThreads];      // offset of the record in the ring (rec_elem), -1: no record
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kRunoffInThreads + tid;
    const int k = blockIdx.y;
    const bool cumulative = g.flags & RR_RUNOFF_CUMULATIVE, force_positive = g.flags & RR_RUNOFF_FORCE_POSITIVE,
               keep_nan = g.flags & RR_RUNOFF_KEEP_NAN;
    slot[tid] = -1;
    if (i < a.n) {
        const int2 meta = a.colmeta[i];
        const int32_t lag = meta.y & kLagMask, o = lag & 15;
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
        slot[tid] = rec_elem(a.rec_chunks.mod(chunk), a.np, meta.x);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int piece = it * kRunoffInThreads + tid, r = piece >> 3, part = piece & 7;      // 8 consecutive lanes = one record
        const int64_t sl = slot[r];
        if (sl < 0) continue;
        reinterpret_cast<double2 *>(a.rec + sl)[part] = make_double2(stage[r][2 * part], stage[r][2 * part + 1]);
    }
}

// Device copy rate probe (bench.py reports it beside the nominal HBM peak): one 16-byte element per thread, non-temporal --
// the shape that copies fastest on this chip (6.3-6.7 TB/s; the grid-stride form of rounds 1-2 read 4.5-5.7 TB/s on the same
// boxes, profiles/r03_hbm_probe_*.txt: workgroups dispatched in order keep the chip's accesses inside a narrow window).
__global__ __launch_bounds__(kBlock) void k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t count)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const d2 *>(src) + i), reinterpret_cast<d2 *>(dst) + i);
}

}  // namespace
