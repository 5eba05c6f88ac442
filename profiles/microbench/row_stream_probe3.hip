// microbenchmark (round 5), third version: WHICH load / store instruction shapes a CU streams row segments fastest with.
// row_stream_probe2 showed 50 % spreads between shapes that move the same bytes (stores through a descriptor that ends at the tile's last
// column: 1,034 us; round 4's exec-masked plain stores: 681 us), so the shape matters more than the roles around it.  Same geometry as
// probe2 (one workgroup of 512 threads per CU, tiles of `cols` columns x 512 rows, rows 8 MB apart, an LDS-only barrier per row):
//   LK (loads, waves 0 and 1, 16 B per lane and row, register ring of DEPTH rows): 0 none | 1 raw buffer load, lanes past the tile dropped by
//      the range check (k_direct r04) | 2 plain global load under an exec mask | 3 raw buffer load under an exec mask
//   SK (stores, wave 2, two 16-byte stores per lane and row): 0 none | 1 raw buffer store through a descriptor that ends at the tile
//      (k_direct r04) | 2 plain global store under an exec mask | 3 raw buffer store under an exec mask (descriptor = whole row)
// (a 32-deep ring of plain global loads spills -- one 64-bit address per row in flight -- and is not run: that was round 4's cliff)
//   HINT: 0 default | 1 non-temporal (nt) | 2 sc1 (stores: write-through, the line dropped from L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kFlags = 0x00020000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)kFlags); }
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct Args { const double *in; double *out, *sink; int64_t n; int32_t cols, n_tiles, K; };
template <int HINT> struct Aux { static constexpr int v = HINT == 1 ? 2 : (HINT == 2 ? 16 : 0); };      // gfx940+: sc0 = 1, nt = 2, sc1 = 16

template <int DEPTH, int LK, int SK, int HL, int HS>
__global__ __launch_bounds__(512, 1) void probe(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, ln = tid & 63;
    const uint32_t row_bytes = (uint32_t)a.n * 8u;
    v2 acc = {0.0, 0.0};
    for (int32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const uint32_t c0 = (uint32_t)t * (uint32_t)a.cols;
        __syncthreads();
        if (LK && wave < 2) {
            const uint32_t col = (uint32_t)wave * 128u + 2u * ln;
            const bool in_tile = col < (uint32_t)a.cols;
            const uint32_t voff = LK == 1 ? (in_tile ? (c0 + col) * 8u : 0xFFFFFFF0u) : (c0 + col) * 8u;
            const double *row = a.in;
            v2 ring[DEPTH];
            auto req = [&](v2 &dst) {
                if (LK == 2) { if (in_tile) dst = HL == 1 ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(row + c0 + col)) : *reinterpret_cast<const v2 *>(row + c0 + col); }
                else if (LK == 3) { if (in_tile) { const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(row, row_bytes), (int)voff, 0, Aux<HL>::v); __builtin_memcpy(&dst, &b, 16); } }
                else { const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(row, row_bytes), (int)voff, 0, Aux<HL>::v); __builtin_memcpy(&dst, &b, 16); }
                row += a.n;
            };
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) { ring[j] = acc; req(ring[j]); }
            for (int32_t r0 = 0; r0 < a.K; r0 += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) { acc += ring[j]; req(ring[j]); barrier_lds(); }
            }
        } else if (SK && wave == 2) {
            const uint32_t ca = 2u * ln, cb = 128u + 2u * ln;
            const bool ina = ca < (uint32_t)a.cols, inb = cb < (uint32_t)a.cols;
            const uint32_t tile_end = (c0 + (uint32_t)a.cols) * 8u;
            double *row = a.out;
            const v2 val = {1.0 + t, 2.0};
            u32x4 b; __builtin_memcpy(&b, &val, 16);
            for (int32_t r = 0; r < a.K; ++r) {
                if (SK == 1) {
                    const __amdgpu_buffer_rsrc_t dst = rsrc_of(row, tile_end);
                    __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)((c0 + ca) * 8u), 0, Aux<HS>::v);
                    __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)((c0 + cb) * 8u), 0, Aux<HS>::v);
                } else if (SK == 2) {
                    if (HS == 1) { if (ina) __builtin_nontemporal_store(val, reinterpret_cast<v2 *>(row + c0 + ca)); if (inb) __builtin_nontemporal_store(val, reinterpret_cast<v2 *>(row + c0 + cb)); }
                    else { if (ina) *reinterpret_cast<v2 *>(row + c0 + ca) = val; if (inb) *reinterpret_cast<v2 *>(row + c0 + cb) = val; }
                } else {
                    const __amdgpu_buffer_rsrc_t dst = rsrc_of(row, row_bytes);
                    if (ina) __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)((c0 + ca) * 8u), 0, Aux<HS>::v);
                    if (inb) __builtin_amdgcn_raw_buffer_store_b128(b, dst, (int)((c0 + cb) * 8u), 0, Aux<HS>::v);
                }
                row += a.n;
                barrier_lds();
            }
        } else {
            for (int32_t r = 0; r < a.K; ++r) barrier_lds();
        }
    }
    if (acc.x + acc.y == 12345.678) a.sink[tid] = acc.x;
}

template <int DEPTH, int LK, int SK, int HL, int HS>
static double run(const Args &a, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<DEPTH, LK, SK, HL, HS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<DEPTH, LK, SK, HL, HS>), dim3(grid), dim3(512), 150 * 1024, 0, a);
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
    a.n = 1000000; a.K = 512;
    const int grid = 256;
    const size_t rows = (size_t)a.K + 40;
    double *in, *out;
    CK(hipMalloc(&in, rows * a.n * 8 + 4096)); CK(hipMalloc(&out, rows * a.n * 8 + 4096)); CK(hipMalloc(&a.sink, 4096));
    CK(hipMemset(in, 0, rows * a.n * 8)); CK(hipMemset(out, 0, rows * a.n * 8));
    a.in = in; a.out = out;
    for (int cols : {200, 256}) {
        a.cols = cols; a.n_tiles = (int32_t)(a.n / cols) - 1;
        const double bytes = (double)a.n_tiles * a.K * a.cols * 8;
        printf("%d tiles of %d columns x %d rows on %d workgroups of 512 threads; an LDS-only barrier per row\n", a.n_tiles, a.cols, a.K, grid);
#define RUN(D, LK, SK, HL, HS, label) { const double ms = run<D, LK, SK, HL, HS>(a, grid); const double b = (LK ? bytes : 0) + (SK ? bytes : 0); \
    printf("  %-100s %8.1f us  %6.2f TB/s\n", label, ms * 1e3, b / ms / 1e9); }
        RUN(16, 1, 0, 0, 0, "loads: raw buffer, range-check drop, 16 in flight");
        RUN(24, 1, 0, 0, 0, "loads: raw buffer, range-check drop, 24 in flight");
        RUN(32, 1, 0, 0, 0, "loads: raw buffer, range-check drop, 32 in flight");
        RUN(16, 1, 0, 1, 0, "loads: raw buffer, range-check drop, 16 in flight, nt");
        RUN(32, 1, 0, 1, 0, "loads: raw buffer, range-check drop, 32 in flight, nt");
        RUN(16, 2, 0, 0, 0, "loads: plain global under an exec mask, 16 in flight");
        RUN(16, 2, 0, 1, 0, "loads: plain global under an exec mask, 16 in flight, nt");
        RUN(32, 3, 0, 0, 0, "loads: raw buffer under an exec mask, 32 in flight");
        RUN(16, 3, 0, 0, 0, "loads: raw buffer under an exec mask, 16 in flight");
        RUN(32, 3, 0, 1, 0, "loads: raw buffer under an exec mask, 32 in flight, nt");
        RUN(16, 0, 1, 0, 0, "stores: raw buffer, descriptor ends at the tile");
        RUN(16, 0, 1, 0, 1, "stores: raw buffer, descriptor ends at the tile, nt");
        RUN(16, 0, 1, 0, 2, "stores: raw buffer, descriptor ends at the tile, sc1");
        RUN(16, 0, 2, 0, 0, "stores: plain global under an exec mask");
        RUN(16, 0, 2, 0, 1, "stores: plain global under an exec mask, nt");
        RUN(16, 0, 3, 0, 0, "stores: raw buffer under an exec mask");
        RUN(16, 0, 3, 0, 1, "stores: raw buffer under an exec mask, nt");
        RUN(16, 0, 3, 0, 2, "stores: raw buffer under an exec mask, sc1");
        RUN(32, 1, 1, 0, 0, "both: k_direct r04's shapes (range-check drops, 32 in flight)");
        RUN(16, 1, 1, 0, 0, "both: range-check drops, 16 in flight");
        RUN(16, 2, 2, 0, 0, "both: plain global under exec masks, 16 in flight");
        RUN(32, 3, 3, 0, 0, "both: raw buffer under exec masks, 32 in flight");
        RUN(16, 2, 2, 1, 1, "both: plain global under exec masks, 16 in flight, nt loads and stores");
        RUN(16, 3, 3, 1, 1, "both: raw buffer under exec masks, 16 in flight, nt loads and stores");
        RUN(24, 3, 3, 1, 1, "both: raw buffer under exec masks, 24 in flight, nt loads and stores");
        RUN(24, 3, 3, 1, 2, "both: raw buffer under exec masks, 24 in flight, nt loads, sc1 stores");
        RUN(24, 3, 3, 0, 0, "both: raw buffer under exec masks, 24 in flight");
    }
    return 0;
}
