// microbenchmark (round 5): the memory side of k_direct without any routing, second version.
// Round 4's probe (row_stream_probe.hip) showed a 4.5x cliff from 16 to 32 rows in flight; its ISA (hipcc -save-temps) shows why: the
// 32-deep register ring plus one 64-bit address per row in flight does not fit the wave's registers, the ring is spilled to scratch,
// and every scratch reload waits for ALL outstanding loads (s_waitcnt vmcnt(0)): one row in flight, not 32.  Not a property of the
// memory system.  Here the row address is one scalar base stepped per row (a buffer descriptor per row, a constant per-lane offset),
// as in k_direct, so nothing spills at any depth, and the rows can also arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no register
// ring at all, the rows in flight live in an LDS ring).
//
// One workgroup of 512 threads per CU walks column-range tiles of `cols` columns through K rows of an (rows x n) array:
//   LOADER 0 none | 1 waves 0, 1: a 16-byte load per lane and row into a register ring of DEPTH rows
//          | 2 wave 0: two LDS-DMA instructions per row (1 KiB each) into an LDS ring, DEPTH rows in flight
//          | 3 waves 0, 1: one LDS-DMA instruction per row each
//   STORER 0 none | 1 wave 2: two 16-byte stores per lane and row from registers | 2 wave 2: the same from the LDS ring (two ds_read_b128)
//          | 3 waves 2, 3: one store each from the LDS ring
//   REC    0 none | N: wave 4 stores eight scattered 128-byte records (one store instruction) every N-th row (k_direct's wave 7)
//   BARRIER: an LDS-only workgroup barrier per row, as k_direct's tick has
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/row_stream_probe2.hip -o /tmp/row_stream_probe2 && /tmp/row_stream_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr uint32_t kFlags = 0x00020000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)kFlags); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Args {
    const double *in; double *out, *rec, *sink;
    int64_t n; int32_t cols, n_tiles, K, np, rec_chunks;
};

template <int DEPTH, int LOADER, int STORER, int REC, bool BARRIER>
__global__ __launch_bounds__(512, 1) void probe(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int R = DEPTH + 2;                      // rows of the LDS ring (2 KiB each)
    const int tid = threadIdx.x, wave = tid >> 6, ln = tid & 63;
    const uint32_t row_bytes = (uint32_t)a.n * 8u;
    v2 acc = {0.0, 0.0};
    for (int32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const uint32_t c0 = (uint32_t)t * (uint32_t)a.cols;
        if (BARRIER || LOADER >= 2) __syncthreads();
        if (LOADER == 1 && wave < 2) {
            const uint32_t voff = (c0 + (uint32_t)wave * 128u + 2u * ln) * 8u;
            const double *row = a.in;
            v2 ring[DEPTH];
            auto req = [&](v2 &dst) { const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(row, row_bytes), (int)voff, 0, 0); __builtin_memcpy(&dst, &b, 16); row += a.n; };
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) req(ring[j]);
            for (int32_t r0 = 0; r0 < a.K; r0 += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    acc += ring[j];
                    req(ring[j]);      // (rows past K: the array has DEPTH spare rows)
                    if (BARRIER) barrier_lds();
                }
            }
        } else if ((LOADER == 2 && wave == 0) || (LOADER == 3 && wave < 2)) {
            constexpr int PER = LOADER == 2 ? 2 : 1;       // DMA instructions per row and wave
            const uint32_t voff = (c0 + (LOADER == 3 ? (uint32_t)wave * 128u : 0u) + 2u * ln) * 8u;
            const double *row = a.in;
            int32_t slot = 0;
            auto req = [&]() {
                const __amdgpu_buffer_rsrc_t src = rsrc_of(row, row_bytes);
                lds_void *dst = (lds_void *)(lds + (size_t)slot * 256 + (LOADER == 3 ? wave * 128 : 0));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src, dst, 16, (int)voff, 0, 0, 0);
                if (PER == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(src, dst, 16, (int)voff, 0, 1024, 0);
                row += a.n; slot = slot + 1 == R ? 0 : slot + 1;
            };
            for (int j = 0; j < DEPTH; ++j) req();
            for (int32_t r = 0; r < a.K; ++r) {
                req();
                wait_vm<PER * DEPTH>();      // row r has landed
                if (BARRIER) barrier_lds();
            }
            wait_vm<0>();
        } else if (STORER && (wave == 2 || (STORER == 3 && wave == 3))) {
            const bool two = STORER != 3;
            const uint32_t va = (c0 + (STORER == 3 ? (uint32_t)(wave - 2) * 128u : 0u) + 2u * ln) * 8u, tile_end = (c0 + (uint32_t)a.cols) * 8u;
            double *row = a.out;
            int32_t slot = 0;
            const v2 val = {1.0 + t, 2.0};
            for (int32_t r = 0; r < a.K; ++r) {
                v2 xa = val, xb = val;
                if (STORER >= 2) {
                    const char *src = reinterpret_cast<const char *>(lds) + (size_t)slot * 2048 + (STORER == 3 ? (wave - 2) * 1024 : 0) + ln * 16;
                    xa = *reinterpret_cast<const v2 *>(src);
                    if (two) xb = *reinterpret_cast<const v2 *>(src + 1024);
                    slot = slot + 1 == R ? 0 : slot + 1;
                }
                const __amdgpu_buffer_rsrc_t dst = rsrc_of(row, tile_end);
                u32x4 b; __builtin_memcpy(&b, &xa, 16);
                __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)va, 0, 0);
                if (two) { __builtin_memcpy(&b, &xb, 16); __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)(va + 1024u), 0, 0); }
                row += a.n;
                if (BARRIER) barrier_lds();
            }
        } else if (REC && wave == 4) {
            const v2 val = {3.0, 4.0};
            uint32_t h = (uint32_t)t * 2654435761u + (uint32_t)(ln >> 3) * 40503u;
            for (int32_t r = 0; r < a.K; ++r) {
                if (r % REC == 0) {
                    h = h * 1664525u + 1013904223u;
                    const uint32_t pos = h % (uint32_t)a.np, chunk = (uint32_t)(r >> 4) % (uint32_t)a.rec_chunks;
                    double *dst = a.rec + ((int64_t)chunk * a.np + pos) * 16 + 2 * (ln & 7);
                    *reinterpret_cast<v2 *>(dst) = val;
                }
                if (BARRIER) barrier_lds();
            }
        } else if (BARRIER) {
            for (int32_t r = 0; r < a.K; ++r) barrier_lds();
        }
    }
    if (acc.x + acc.y == 12345.678) a.sink[tid] = acc.x;      // keeps the loads
}

template <int DEPTH, int L, int S, int RC, bool B>
static double run(const Args &a, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<DEPTH, L, S, RC, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<DEPTH, L, S, RC, B>), dim3(grid), dim3(512), 150 * 1024, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    Args a{};
    a.n = 1000000; a.K = 512; a.cols = argc > 1 ? atoi(argv[1]) : 200; a.n_tiles = (int32_t)(a.n / a.cols) - 1; a.np = 100263; a.rec_chunks = 1200;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    const size_t rows = (size_t)a.K + 72;
    double *in, *out;
    CK(hipMalloc(&in, rows * a.n * 8 + 4096)); CK(hipMalloc(&out, rows * a.n * 8 + 4096)); CK(hipMalloc(&a.sink, 4096));
    CK(hipMalloc(&a.rec, (size_t)a.rec_chunks * a.np * 128));
    CK(hipMemset(in, 0, rows * a.n * 8)); CK(hipMemset(out, 0, rows * a.n * 8));
    a.in = in; a.out = out;
    const double bytes = (double)a.n_tiles * a.K * a.cols * 8;
    printf("%d tiles of %d columns x %d rows on %d workgroups of 512 threads; rows n = 1M doubles apart\n", a.n_tiles, a.cols, a.K, grid);
#define RUN(D, LD, ST, RC, BR, label) { const double ms = run<D, LD, ST, RC, BR>(a, grid); const double b = (LD ? bytes : 0) + (ST ? bytes : 0); \
    printf("  %-86s %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", label, ms * 1e3, b / ms / 1e9, b / ms / 1e6 / grid); }
    RUN(16, 1, 0, 0, false, "loads, register ring, 16 rows in flight, no barrier");
    RUN(32, 1, 0, 0, false, "loads, register ring, 32 rows in flight, no barrier");
    RUN(48, 1, 0, 0, false, "loads, register ring, 48 rows in flight, no barrier");
    RUN(32, 1, 0, 0, true,  "loads, register ring, 32 rows in flight, barrier per row");
    RUN(16, 2, 0, 0, false, "loads, LDS-DMA by one wave (2 x 1 KiB per row), 16 rows in flight, no barrier");
    RUN(30, 2, 0, 0, false, "loads, LDS-DMA by one wave, 30 rows in flight, no barrier");
    RUN(30, 2, 0, 0, true,  "loads, LDS-DMA by one wave, 30 rows in flight, barrier per row");
    RUN(32, 3, 0, 0, false, "loads, LDS-DMA by two waves (1 KiB per row each), 32 rows in flight, no barrier");
    RUN(48, 3, 0, 0, false, "loads, LDS-DMA by two waves, 48 rows in flight, no barrier");
    RUN(48, 3, 0, 0, true,  "loads, LDS-DMA by two waves, 48 rows in flight, barrier per row");
    RUN(16, 0, 1, 0, false, "stores from registers, one wave, no barrier");
    RUN(16, 0, 1, 0, true,  "stores from registers, one wave, barrier per row");
    RUN(16, 0, 2, 0, true,  "stores from LDS (2 ds_read_b128), one wave, barrier per row");
    RUN(16, 0, 3, 0, true,  "stores from LDS, two waves, barrier per row");
    RUN(16, 0, 0, 1, true,  "records only: eight scattered 128-byte records per row, barrier per row");
    RUN(16, 0, 0, 4, true,  "records only: eight records every 4th row, barrier per row");
    RUN(32, 1, 1, 0, true,  "both: register ring 32 + stores from registers, barrier per row");
    RUN(30, 2, 2, 0, true,  "both: LDS-DMA one wave 30 + stores from LDS one wave, barrier per row");
    RUN(48, 3, 2, 0, true,  "both: LDS-DMA two waves 48 + stores from LDS one wave, barrier per row");
    RUN(48, 3, 3, 0, true,  "both: LDS-DMA two waves 48 + stores from LDS two waves, barrier per row");
    RUN(48, 3, 3, 0, false, "both: LDS-DMA two waves 48 + stores two waves, NO barrier (stores read rows that may not have landed)");
    RUN(48, 3, 3, 4, true,  "both (two + two waves) + eight records every 4th row, barrier per row");
    RUN(48, 3, 3, 1, true,  "both (two + two waves) + eight records every row, barrier per row");
    RUN(32, 1, 1, 4, true,  "both, register ring 32 + register stores + eight records every 4th row, barrier per row");
    return 0;
}
