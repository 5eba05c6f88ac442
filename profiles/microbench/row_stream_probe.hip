// microbenchmark: the memory side of k_direct without any routing ("what bounds k_direct is still open").
// One workgroup per CU (the LDS request keeps a second one away) walks column-range tiles of `cols` columns through `K` rows of
// an (rows x n) array, the next row `n` doubles further on: wave 0 and 1 load a row segment (16 bytes per lane, `depth` rows in
// flight per wave in a register ring), wave 2 stores one into a second array, as k_direct's rows-in and rows-out waves do.
// Variants: loads only / stores only / both; a workgroup barrier per row or none; row stride n (8 MB) or the rows of a tile
// packed one behind the other (stride = cols).
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/row_stream_probe.hip -o /tmp/row_stream_probe && /tmp/row_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

template <int DEPTH, bool LOADS, bool STORES, bool BARRIER>
__global__ __launch_bounds__(512, 1) void probe(const double *__restrict__ in, double *__restrict__ out, double *sink, int64_t row_stride, int32_t cols,
                                              int32_t n_tiles, int32_t K, int64_t tile_stride)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, ln = tid & 63;
    v2 acc = {0.0, 0.0};
    for (int32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const double *src = in + (int64_t)t * tile_stride + (wave & 1) * 128 + 2 * ln;
        double *dst = out + (int64_t)t * tile_stride + 2 * ln;
        const bool lane_in = (wave & 1) * 128 + 2 * ln < cols;
        if (wave < 2 && LOADS) {
            v2 ring[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) ring[j] = lane_in ? *reinterpret_cast<const v2 *>(src + (int64_t)j * row_stride) : acc;
            for (int32_t r0 = 0; r0 < K; r0 += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    acc += ring[j];
                    const int32_t nxt = r0 + j + DEPTH;
                    if (lane_in && nxt < K) ring[j] = *reinterpret_cast<const v2 *>(src + (int64_t)nxt * row_stride);
                    if (BARRIER) __builtin_amdgcn_s_barrier();
                }
            }
        } else if (wave == 2 && STORES) {
            const v2 val = {1.0 + t, 2.0};
            for (int32_t r = 0; r < K; ++r) {
                if (2 * ln < cols) *reinterpret_cast<v2 *>(dst + (int64_t)r * row_stride) = val;
                if (128 + 2 * ln < cols) *reinterpret_cast<v2 *>(dst + (int64_t)r * row_stride + 128) = val;
                if (BARRIER) __builtin_amdgcn_s_barrier();
            }
        } else if (BARRIER) {
            for (int32_t r = 0; r < K; ++r) __builtin_amdgcn_s_barrier();
        }
    }
    if (acc.x + acc.y == 12345.678) sink[tid] = acc.x;      // keeps the loads
    if (tid == 0 && lds[0] == 1.5) sink[0] = 1.0;
}

template <int DEPTH, bool L, bool S, bool B>
static double run(const double *in, double *out, double *sink, int64_t row_stride, int cols, int n_tiles, int K, int64_t tile_stride, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<DEPTH, L, S, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<DEPTH, L, S, B>), dim3(grid), dim3(512), 150 * 1024, 0, in, out, sink, row_stride, cols, n_tiles, K, tile_stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int64_t n = 1000000;
    const int K = 512, cols = argc > 1 ? atoi(argv[1]) : 200, n_tiles = 4864, grid = 256;
    double *in, *out, *sink;
    CK(hipMalloc(&in, (size_t)K * n * 8 + 4096)); CK(hipMalloc(&out, (size_t)K * n * 8 + 4096)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(in, 0, (size_t)K * n * 8)); CK(hipMemset(out, 0, (size_t)K * n * 8));
    const double bytes_l = (double)n_tiles * K * cols * 8, bytes_s = bytes_l;
    printf("%d tiles of %d columns x %d rows on %d workgroups of 512 (one per CU); loads 2 waves x 16 B per lane, stores 1 wave\n", n_tiles, cols, K, grid);
    struct { const char *name; int64_t row_stride, tile_stride; } lay[2] = {{"rows n = 1M doubles apart (k_direct)", n, (int64_t)cols}, {"a tile's rows packed", cols, (int64_t)cols * K}};
    for (auto &L : lay) {
        printf("layout: %s\n", L.name);
#define RUN(D, LD, ST, BR, label) { const double ms = run<D, LD, ST, BR>(in, out, sink, L.row_stride, cols, n_tiles, K, L.tile_stride, grid); \
        const double b = (LD ? bytes_l : 0) + (ST ? bytes_s : 0); \
        printf("  %-44s %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", label, ms * 1e3, b / ms / 1e9, b / ms / 1e6 / grid); }
        RUN(32, true, false, false, "loads only, 32 rows in flight, no barrier");
        RUN(16, true, false, false, "loads only, 16 rows in flight, no barrier");
        RUN(32, true, false, true, "loads only, 32 in flight, barrier per row");
        RUN(32, false, true, false, "stores only, no barrier");
        RUN(32, false, true, true, "stores only, barrier per row");
        RUN(32, true, true, false, "both, no barrier");
        RUN(32, true, true, true, "both, barrier per row");
        RUN(16, true, true, true, "both, 16 in flight, barrier per row");
    }
    return 0;
}
