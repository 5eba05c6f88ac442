// microbenchmark: the two access shapes of the record passes (k_rec_in / k_rec_out), each alone, against what a plain
// streaming kernel reaches on the same box (hbm_probe.hip).  The alias build of the engine (RR_ALIAS) says the passes are
// bound by memory, and that each side alone is slow: rows only 297 us per 1.14 GB (3.9 TB/s), records only 334 us per 1.05 GB.
//   rows:    a workgroup reads R = 143 rows x C columns of a (rows, n) float64 array, row pitch n * 8 B (8 MB at n = 1M)
//            C = 32 / 64 / 128 columns (256 B / 512 B / 1 KiB per row piece), 8 B or 16 B per lane, plain or XCD-contiguous order
//   records: 8 records of 128 B per column into 8 planes of np x 128 B, position = random permutation of the column
//            (the engine's case), the identity (upper bound: no scatter), records of 256 / 512 B (fewer, larger pieces)
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/rec_probe.hip -o profiles/microbench/rec_probe.bin && ./rec_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));
typedef double v2u __attribute__((ext_vector_type(2), aligned(8)));      // a 16-byte access at element alignment

__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t nwg)
{
    const uint32_t q = nwg >> 3, r = nwg & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
}

// rows x C tile per workgroup, all loads in flight, result discarded
template <int C, int VEC, int NT, int SWZ>
__global__ __launch_bounds__(256) void tile_read(const double *__restrict__ rows, int64_t n, int R, int64_t row0, double *sink)
{
    constexpr int LPR = C / VEC;               // lanes per row piece
    constexpr int RPW = 256 / LPR;             // rows per pass of the workgroup
    const int tid = threadIdx.x, c = (tid % LPR) * VEC, r0 = tid / LPR;
    const int64_t col0 = (int64_t)(SWZ ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * C;
    const int64_t i = col0 + c < n - VEC ? col0 + c : n - VEC;
    double acc = 0.0;
    constexpr int MAXQ = (143 + RPW - 1) / RPW;
    double v[MAXQ * VEC];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int r = r0 + q * RPW < R ? r0 + q * RPW : R - 1;
        const double *p = rows + (row0 + r) * n + i;
        if (VEC == 2) {
            const v2u t = NT ? __builtin_nontemporal_load((const v2u *)p) : *(const v2u *)p;      // n odd: rows are not 16-byte aligned
            v[2 * q] = t.x; v[2 * q + 1] = t.y;
        } else {
            v[q] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ * VEC; ++q) acc += v[q];
    if (acc == 123.456) sink[0] = acc;
}

// write side of the in-pass: per column 8 records (RECB bytes each) into planes k = 0..7; LPRc lanes of 16 B per record
template <int RECB, int NT>
__global__ __launch_bounds__(256) void rec_write(double *__restrict__ rec, const int32_t *__restrict__ pos, int64_t n, int64_t np, int planes, int cols_per_wg)
{
    constexpr int LPRc = RECB / 16;
    const int tid = threadIdx.x;
    const int pieces = cols_per_wg * planes * LPRc;
    const int64_t col0 = (int64_t)blockIdx.x * cols_per_wg;
    for (int piece = tid; piece < pieces; piece += 256) {
        const int c = piece / (planes * LPRc), k = (piece / LPRc) % planes, part = piece % LPRc;
        const int64_t i = col0 + c;
        if (i >= n) continue;
        const int32_t p = pos[i];
        v2 *dst = reinterpret_cast<v2 *>(rec + ((int64_t)k * np + p) * (RECB / 8)) + part;
        const v2 v = {1.0, 2.0};
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}

template <int RECB, int NT>
__global__ __launch_bounds__(256) void rec_read(const double *__restrict__ rec, const int32_t *__restrict__ pos, int64_t n, int64_t np, int planes, int cols_per_wg, double *sink)
{
    constexpr int LPRc = RECB / 16;
    const int tid = threadIdx.x;
    const int pieces = cols_per_wg * planes * LPRc;
    const int64_t col0 = (int64_t)blockIdx.x * cols_per_wg;
    v2 acc = {0.0, 0.0};
    for (int piece = tid; piece < pieces; piece += 256) {
        const int c = piece / (planes * LPRc), k = (piece / LPRc) % planes, part = piece % LPRc;
        const int64_t i = col0 + c;
        if (i >= n) continue;
        const int32_t p = pos[i];
        const v2 *src = reinterpret_cast<const v2 *>(rec + ((int64_t)k * np + p) * (RECB / 8)) + part;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    if (acc.x == 123.456) sink[0] = acc.x;
}

// write side of the out-pass: 128 rows x C columns per workgroup, coalesced
template <int C, int NT, int SWZ>
__global__ __launch_bounds__(256) void tile_write(double *__restrict__ rows, int64_t n, int R)
{
    const int tid = threadIdx.x, c = tid % C;
    const int64_t col0 = (int64_t)(SWZ ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * C;
    const int64_t i = col0 + c;
    if (i >= n) return;
    for (int r = tid / C; r < R; r += 256 / C) {
        double *p = rows + (int64_t)r * n + i;
        if (NT) __builtin_nontemporal_store(1.0, p); else *p = 1.0;
    }
}

// k_tile's record traffic alone: a persistent workgroup of 512 lanes reads the 64 KB of one (tile, chunk) -- 512 records of
// 128 B -- with eight 16-byte loads per lane and writes them back in place.  SHAPE 0: the engine's (rounds 1-2): four lanes per
// 64-byte sector, an instruction touches 16 half lines, the next one their other halves.  SHAPE 1: eight lanes per 128-byte
// record, an instruction touches 8 whole lines.  SHAPE 2: lane-linear (an instruction = 1 KiB contiguous).
template <int SHAPE, int NT>
__global__ __launch_bounds__(512) void tile_rw(double *__restrict__ rec, int64_t n_blocks, int64_t out_shift = 0)      // out_shift: write block b + out_shift (not in place)
{
    const int tid = threadIdx.x, ln = tid & 63, wb = tid - ln;      // wb: first position of the wave inside the tile
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        char *base = reinterpret_cast<char *>(rec) + b * 65536;
        char *obase = reinterpret_cast<char *>(rec) + (b + out_shift) * 65536;
        v2 v[8];
        int off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SHAPE == 0) { const int i = j >> 1, half = j & 1; off[j] = (wb + (i >> 1) * 32 + 16 * (i & 1) + (ln >> 2)) * 128 + half * 64 + (ln & 3) * 16; }
            else if (SHAPE == 1) off[j] = (wb + 8 * j + (ln >> 3)) * 128 + (ln & 7) * 16;
            else off[j] = wb * 128 + j * 1024 + ln * 16;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(base + off[j])) : *reinterpret_cast<const v2 *>(base + off[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] += 1.0;
            if (NT) __builtin_nontemporal_store(v[j], reinterpret_cast<v2 *>(obase + off[j])); else *reinterpret_cast<v2 *>(obase + off[j]) = v[j];
        }
    }
}

// The in-pass without its transpose: a workgroup reads its 143 x 32 tile of rows (18 loads per thread), then writes 8 records
// per column (8 stores per thread) whose values depend on what it read -- no LDS, no barrier.  SYNC: one __syncthreads between
// the two halves, as the real kernel has.  What the mix of the two access streams in one workgroup costs by itself.
template <int SYNC, int NTL>
__global__ __launch_bounds__(256) void tile_read_rec_write(const double *__restrict__ rows, double *__restrict__ rec, const int32_t *__restrict__ pos,
                                                           int64_t n, int64_t np, int R)
{
    __shared__ double dummy[256];
    const int tid = threadIdx.x, c = tid % 32, r0 = tid / 32;
    const int64_t col0 = (int64_t)blockIdx.x * 32;
    const int64_t i = col0 + c < n ? col0 + c : n - 1;
    double v[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const int r = r0 + q * 8 < R ? r0 + q * 8 : R - 1;
        v[q] = NTL ? __builtin_nontemporal_load(rows + (int64_t)r * n + i) : rows[(int64_t)r * n + i];
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 18; ++q) acc += v[q];
    if (SYNC) { dummy[tid] = acc; __syncthreads(); acc = dummy[tid ^ 1]; }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int piece = it * 256 + tid;
        const int cc = piece / 64, k = (piece >> 3) % 8, part = piece & 7;
        const int64_t col = col0 + cc;
        if (col >= n) continue;
        const int32_t p = pos[col];
        *(reinterpret_cast<v2 *>(rec + ((int64_t)k * np + p) * 16) + part) = v2{acc, acc};
    }
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    template <typename F> float best(F f, int reps = 5)
    {
        float b = 1e30f;
        f();
        for (int r = 0; r < reps; ++r) {
            (void)hipEventRecord(e0, nullptr);
            f();
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            b = std::min(b, ms);
        }
        return b;
    }
};

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000, np = n + n / 28, R = 143, planes = 8;
    Timer T;
    double *rows, *rec, *sink; int32_t *pos_rand, *pos_id;
    const int64_t nrows = 288;
    CK(hipMalloc(&rows, nrows * n * 8)); CK(hipMemset(rows, 0, nrows * n * 8));
    CK(hipMalloc(&rec, 32 * np * 128)); CK(hipMemset(rec, 0, 32 * np * 128));      // 32 planes of np records: room for 512-byte records in 8 planes
    CK(hipMalloc(&sink, 64));
    std::vector<int32_t> perm((size_t)n); std::iota(perm.begin(), perm.end(), 0);
    CK(hipMalloc(&pos_id, n * 4)); CK(hipMemcpy(pos_id, perm.data(), n * 4, hipMemcpyHostToDevice));
    std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
    CK(hipMalloc(&pos_rand, n * 4)); CK(hipMemcpy(pos_rand, perm.data(), n * 4, hipMemcpyHostToDevice));
    printf("n = %lld columns, row pitch %lld B, %lld rows per tile; rates in TB/s of the bytes the shape moves\n", (long long)n, (long long)n * 8, (long long)R);
#define ROW(name, bytes, call) do { const float ms = T.best([&] { call; }); printf("  %-72s %8.1f us  %6.3f TB/s\n", name, ms * 1e3, (double)(bytes) / (ms * 1e-3) / 1e12); fflush(stdout); } while (0)
    const double rb = (double)R * n * 8;
    printf("rows read by column tiles (in-pass, row side):\n");
    ROW("32 columns (256 B pieces),  8 B per lane", rb, (tile_read<32, 1, 0, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("32 columns,  8 B per lane, non-temporal", rb, (tile_read<32, 1, 1, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("32 columns,  8 B per lane, XCD-contiguous tiles", rb, (tile_read<32, 1, 0, 1><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("32 columns, 16 B per lane", rb, (tile_read<32, 2, 0, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("32 columns, 16 B per lane, XCD-contiguous tiles", rb, (tile_read<32, 2, 0, 1><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("64 columns (512 B pieces),  8 B per lane", rb, (tile_read<64, 1, 0, 0><<<(unsigned)((n + 63) / 64), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("64 columns, 16 B per lane", rb, (tile_read<64, 2, 0, 0><<<(unsigned)((n + 63) / 64), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("64 columns, 16 B per lane, XCD-contiguous tiles", rb, (tile_read<64, 2, 0, 1><<<(unsigned)((n + 63) / 64), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("64 columns, 16 B per lane, non-temporal", rb, (tile_read<64, 2, 1, 0><<<(unsigned)((n + 63) / 64), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("128 columns (1 KiB pieces), 8 B per lane", rb, (tile_read<128, 1, 0, 0><<<(unsigned)((n + 127) / 128), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("128 columns, 16 B per lane", rb, (tile_read<128, 2, 0, 0><<<(unsigned)((n + 127) / 128), 256>>>(rows, n, (int)R, 0, sink)));
    ROW("128 columns, 16 B per lane, XCD-contiguous tiles", rb, (tile_read<128, 2, 0, 1><<<(unsigned)((n + 127) / 128), 256>>>(rows, n, (int)R, 0, sink)));
    const double wb = 128.0 * n * 8;
    printf("rows written by column tiles (out-pass, row side), 128 rows:\n");
    ROW("32 columns", wb, (tile_write<32, 0, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, 128)));
    ROW("32 columns, XCD-contiguous tiles", wb, (tile_write<32, 0, 1><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, 128)));
    ROW("32 columns, XCD-contiguous tiles, non-temporal", wb, (tile_write<32, 1, 1><<<(unsigned)((n + 31) / 32), 256>>>(rows, n, 128)));
    ROW("64 columns, XCD-contiguous tiles", wb, (tile_write<64, 0, 1><<<(unsigned)((n + 63) / 64), 256>>>(rows, n, 128)));
    ROW("128 columns, XCD-contiguous tiles", wb, (tile_write<128, 0, 1><<<(unsigned)((n + 127) / 128), 256>>>(rows, n, 128)));
    const double cb = (double)planes * n * 128;
    printf("records, 8 per column (in-pass writes / out-pass reads), 32 columns per workgroup:\n");
    ROW("write 128 B records, random positions", cb, (rec_write<128, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, (int)planes, 32)));
    ROW("write 128 B records, random positions, non-temporal", cb, (rec_write<128, 1><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, (int)planes, 32)));
    ROW("write 128 B records, positions in column order (no scatter)", cb, (rec_write<128, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_id, n, np, (int)planes, 32)));
    ROW("write 256 B records x 4 per column, random positions", cb, (rec_write<256, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, 4, 32)));
    ROW("write 512 B records x 2 per column, random positions", cb, (rec_write<512, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, 2, 32)));
    ROW("read 128 B records, random positions", cb, (rec_read<128, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, (int)planes, 32, sink)));
    ROW("read 128 B records, random positions, non-temporal", cb, (rec_read<128, 1><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, (int)planes, 32, sink)));
    ROW("read 128 B records, positions in column order (no scatter)", cb, (rec_read<128, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_id, n, np, (int)planes, 32, sink)));
    ROW("read 256 B records x 4 per column, random positions", cb, (rec_read<256, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, 4, 32, sink)));
    ROW("read 512 B records x 2 per column, random positions", cb, (rec_read<512, 0><<<(unsigned)((n + 31) / 32), 256>>>(rec, pos_rand, n, np, 2, 32, sink)));
    printf("the in-pass without its transpose: 143 x 32 rows read, 8 records per column written by the same workgroup (2.2 GB per launch):\n");
    ROW("read then write, no barrier", rb + cb, (tile_read_rec_write<0, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, rec, pos_rand, n, np, (int)R)));
    ROW("read then write, one barrier between", rb + cb, (tile_read_rec_write<1, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, rec, pos_rand, n, np, (int)R)));
    ROW("non-temporal reads, one barrier", rb + cb, (tile_read_rec_write<1, 1><<<(unsigned)((n + 31) / 32), 256>>>(rows, rec, pos_rand, n, np, (int)R)));
    ROW("read then write, positions in column order", rb + cb, (tile_read_rec_write<1, 0><<<(unsigned)((n + 31) / 32), 256>>>(rows, rec, pos_id, n, np, (int)R)));
    {
        const int64_t blocks = 32 * np * 128 / 65536;      // the whole record buffer: 4.2 GB at n = 1M
        const double tb = 2.0 * blocks * 65536;
        printf("records read and written back in place by persistent 512-lane workgroups, 64 KB per step (k_tile's traffic), 2 per CU:\n");
        ROW("four lanes per 64-byte sector (two instructions per line)", tb, (tile_rw<0, 0><<<512, 512>>>(rec, blocks)));
        ROW("eight lanes per 128-byte record (whole lines)", tb, (tile_rw<1, 0><<<512, 512>>>(rec, blocks)));
        ROW("lane-linear (1 KiB per instruction)", tb, (tile_rw<2, 0><<<512, 512>>>(rec, blocks)));
        ROW("eight lanes per record, non-temporal", tb, (tile_rw<1, 1><<<512, 512>>>(rec, blocks)));
        ROW("eight lanes per record, 1024 workgroups", tb, (tile_rw<1, 0><<<1024, 512>>>(rec, blocks)));
        ROW("four lanes per sector, 1024 workgroups", tb, (tile_rw<0, 0><<<1024, 512>>>(rec, blocks)));
        ROW("eight lanes per record, written to the other half of the buffer", tb / 2, (tile_rw<1, 0><<<512, 512>>>(rec, blocks / 2, blocks / 2)));
        ROW("lane-linear, written to the other half of the buffer", tb / 2, (tile_rw<2, 0><<<512, 512>>>(rec, blocks / 2, blocks / 2)));
        ROW("lane-linear, one workgroup per 64 KB (not persistent), in place", tb, (tile_rw<2, 0><<<(unsigned)blocks, 512>>>(rec, blocks)));
        ROW("lane-linear, one workgroup per 64 KB, other half", tb / 2, (tile_rw<2, 0><<<(unsigned)(blocks / 2), 512>>>(rec, blocks / 2, blocks / 2)));
    }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    return 0;
}
