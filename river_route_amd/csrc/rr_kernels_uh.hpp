// rr_kernels_uh.hpp -- unit-hydrograph convolution kernels (UnitHydrograph.py:93-107, direct form).
// Part of the one translation unit rr_engine.hip builds (included from there, in order; not a stand-alone header).
#pragma once

namespace {

// Unit-hydrograph convolution, direct form (UnitHydrograph.py:93-107):
//   out[t, i] = [t < n_ks] state[t, i] + sum_{s=0}^{min(t, n_ks-1)} kernel[s, i] * lateral[t - s, i]
// One reach per lane, TB consecutive outputs per thread held in registers; per tap one kernel value and one
// new lateral value are loaded and the TB-wide window slides in registers.
// TIn: the rows' type (float: runoff depths as the file stores them, exact in float64; sel: their byte order, rr_plan_set_row_format)
template <typename TIn>
__device__ __forceinline__ double uh_row_value(const TIn *__restrict__ rows, int64_t off, uint32_t sel)
{
    if constexpr (std::is_same<TIn, float>::value) return (double)f32_from_file(rows[off], sel);
    else return (double)rows[off];
}

template <int TB, typename TIn = double>
__global__ __launch_bounds__(kBlock) void k_uh_convolve(const double *__restrict__ kernel,
                                                        const double *__restrict__ state,
                                                        const TIn *__restrict__ lateral_rows,
                                                        double *__restrict__ out, int64_t T, int32_t n_ks, int64_t n, uint32_t sel = kSelNative)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    auto lateral_at = [&](int64_t off) { return uh_row_value<TIn>(lateral_rows, off, sel); };
    const int64_t t0 = (int64_t)blockIdx.y * TB;
    double acc[TB], win[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int64_t t = t0 + j;
        acc[j] = (t < n_ks && t < T) ? state[t * n + i] : 0.0;
        win[j] = (t < T) ? lateral_at(t * n + i) : 0.0;   // lateral[t0 + j - s] for s = 0
    }
    for (int32_t s = 0; s < n_ks; ++s) {
        const double kv = kernel[(int64_t)s * n + i];
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[j] += kv * win[j];
        // slide: win[j] <- lateral[t0 + j - (s+1)]
#pragma unroll
        for (int j = TB - 1; j > 0; --j) win[j] = win[j - 1];
        const int64_t tn = t0 - (s + 1);
        win[0] = (tn >= 0) ? lateral_at(tn * n + i) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < TB; ++j)
        if (t0 + j < T) out[(t0 + j) * n + i] = acc[j];
}

// Long-series form of the same convolution: one reach per lane walks its whole time segment once.  The kernel
// column sits in registers (static indices: the tap loop is unrolled over the padded length NK), the last NK
// lateral values in an LDS ring [slot][lane] (conflict-free), so HBM sees each lateral row and each output row
// exactly once and the kernel taps once per segment -- k_uh_convolve re-reads the taps for every 8 rows.
constexpr int kUhThreads = 128;
constexpr int kUhTailThreads = 64;

// NK window slots (power of two), R outputs per pass (every window value read from LDS feeds R accumulators),
// D passes of lateral rows in flight.  The window costs NK * 8 B of LDS per thread, which caps the kernel at about
// one wave per SIMD: latency is hidden by depth instead (R * D rows per lane in flight; registers are free at
// that occupancy).
template <int NK, int NT, int R, int D, typename TIn = double>    // NT taps held in registers (n_ks <= NT <= NK - (R - 1))
__global__ __launch_bounds__(kUhThreads) void k_uh_convolve_ring(const double *__restrict__ kernel,
                                                                const double *__restrict__ state,
                                                                const TIn *__restrict__ lateral_rows,
                                                                double *__restrict__ out, int64_t T, int32_t n_ks,
                                                                int64_t n, int64_t seg_rows, uint32_t sel = kSelNative)
{
    extern __shared__ __attribute__((aligned(16))) double win[];   // [NK][kUhThreads]
    auto lateral_at = [&](int64_t off) { return uh_row_value<TIn>(lateral_rows, off, sel); };
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
        win[(size_t)((uint64_t)t & (NK - 1)) * kUhThreads + tid] = (t >= 0 && s < n_ks) ? lateral_at(t * n + col) : 0.0;
    }
    double nxt[D][R];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int j = 0; j < R; ++j) nxt[dd][j] = lateral_at(min(t0 + dd * R + j, T - 1) * n + col);
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
            const TIn *lat_g = lateral_rows + tb * n + col;     // row tb of this column
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
                for (int j = 0; j < R; ++j) nxt[dd][j] = uh_row_value<TIn>(lat_g, (int64_t)(pp * R + R * D + j) * n, sel);
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
                for (int j = 0; j < R; ++j) nxt[dd][j] = lateral_at(min(t + R * D + j, T - 1) * n + col);     // D passes ahead
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
template <int NK, typename TIn = double>      // TIn: the rows' type (float: runoff depths as the file stores them, exact in float64)
__global__ __launch_bounds__(kUhTailThreads) void k_uh_tail(const double *__restrict__ kernel, double *__restrict__ state,
                                                            const TIn *__restrict__ lateral, int64_t T, int32_t n_ks, int64_t n, uint32_t sel = kSelNative)
{
    auto row_value = [&](int64_t off) -> double {      // (float rows may be a big-endian file's: rr_plan_set_row_format)
        if constexpr (std::is_same<TIn, float>::value) return (double)f32_from_file(lateral[off], sel);
        else return (double)lateral[off];
    };
    const int64_t i = (int64_t)blockIdx.x * kUhTailThreads + threadIdx.x;
    if (i >= n) return;
    if (NK > 0) {
        double kv[NK > 0 ? NK : 1], lat[NK > 0 ? NK : 1];      // lat[j] = lateral[T - 1 - j]
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            kv[k] = k < n_ks ? kernel[(int64_t)k * n + i] : 0.0;
            lat[k] = (k < n_ks - 1 && k < T) ? row_value((T - 1 - k) * n + i) : 0.0;
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
                acc += kernel[(int64_t)k * n + i] * row_value(tt * n + i);
            }
            state[(int64_t)s * n + i] = s == n_ks - 1 ? 0.0 : acc;
        }
    }
}

}  // namespace
