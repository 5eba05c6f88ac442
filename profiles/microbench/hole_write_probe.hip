// microbenchmark (round 5): what bounds the holes' out-pass of the direct row path (k_rec_out over the skeleton's columns: 50k of 1M
// columns, 128 rows a launch, one 8-byte store per hole and row into rows 8 MB apart -- 6.4M isolated stores in 140 us = 46 G stores/s,
// 20 % of the post-order year).  Only the stores: values come from registers.  Holes are every 20th column with a jitter; the row block
// moves through a 4096-row array so that no line is in a cache.  Shapes (threads of a 256-thread workgroup -> (hole, row)):
//   0  16 holes x 16 rows per step, 8 steps (k_rec_out's)        1  256 holes x 1 row per step, 128 steps (a row's holes together)
//   2  64 holes x 4 rows, 32 steps                                3  1 hole x 256... (a column's rows together: 2 holes x 128 rows)
//   4  shape 0 with the non-temporal hint   5  shape 1 with it   6  shape 0, sc1 (write-through)   7  shape 1, sc1
//   8  read-modify-write of the whole 64-byte line by eight lanes (shape: 32 holes x 1 row per step), plain stores
//   9  the same with non-temporal stores
// usage: hole_write_probe [n=1000000] [rows_per_launch=128] [launches=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kFlags = 0x00020000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)kFlags); }
struct Args { double *out; const int32_t *cols; int64_t n; int32_t n_holes, rows; int64_t row0; };

template <int AUX>
__device__ __forceinline__ void put(double *row, uint32_t row_bytes, int32_t col, double v)
{
    if (AUX == 0) { row[col] = v; return; }
    u32x2 b; __builtin_memcpy(&b, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(b, rsrc_of(row, row_bytes), col * 8, 0, AUX);
}

// HC holes x (256 / HC) rows per step
template <int HC, int AUX>
__global__ __launch_bounds__(256) void k_scatter(const Args a)
{
    const int tid = threadIdx.x, c = tid % HC, r0 = tid / HC;
    constexpr int RS = 256 / HC;
    const int64_t h = (int64_t)blockIdx.x * HC + c;
    if (h >= a.n_holes) return;
    const int32_t col = a.cols[h];
    const uint32_t row_bytes = (uint32_t)a.n * 8u;
    for (int r = r0; r < a.rows; r += RS) put<AUX>(a.out + (a.row0 + r) * a.n, row_bytes, col, 1.0 + r + c);
}

// pairs of holes that share a line: hole h and a second store 16 (FAR = 0: the same 64-byte half... of an aligned pair) or 64 bytes on
template <int FAR>
__global__ __launch_bounds__(256) void k_pairs(const Args a)
{
    const int tid = threadIdx.x, c = tid % 16, r0 = tid / 16;
    const int64_t h = 2 * ((int64_t)blockIdx.x * 16 + c);
    if (h >= a.n_holes) return;
    const int32_t col = a.cols[h] & ~15;      // first column of a 128-byte line
    for (int r = r0; r < a.rows; r += 16) {
        double *row = a.out + (a.row0 + r) * a.n;
        row[col + 1] = 1.0 + r;
        row[col + (FAR ? 9 : 3)] = 2.0 + r;
    }
}

// whole 64-byte lines WRITTEN, nothing read (what a patch pass could do if the hole's seven neighbours came from a side buffer): four lanes x
// 16 bytes per line, LPW lines per workgroup and step
typedef double d2 __attribute__((ext_vector_type(2)));
template <int AUX>
__global__ __launch_bounds__(256) void k_full(const Args a)
{
    const int tid = threadIdx.x, g = tid >> 2, part = tid & 3;      // 64 lines per step
    const int64_t h = (int64_t)blockIdx.x * 64 + g;
    if (h >= a.n_holes) return;
    const int32_t c8 = a.cols[h] & ~7;
    const uint32_t row_bytes = (uint32_t)a.n * 8u;
    for (int r = 0; r < a.rows; ++r) {
        double *row = a.out + (a.row0 + r) * a.n;
        const d2 v = {1.0 + r, 2.0 + part};
        if (AUX == 0) *reinterpret_cast<d2 *>(row + c8 + 2 * part) = v;
        else { typedef unsigned int u32x4 __attribute__((ext_vector_type(4))); u32x4 b; __builtin_memcpy(&b, &v, 16);
               __builtin_amdgcn_raw_buffer_store_b128(b, rsrc_of(row, row_bytes), (c8 + 2 * part) * 8, 0, AUX); }
    }
}

// the whole 64-byte line: lanes 8 g ... 8 g + 7 read the line of hole g of this step, lane (col & 7) swaps its value, all write
template <int AUX>
__global__ __launch_bounds__(256) void k_rmw(const Args a)
{
    const int tid = threadIdx.x, g = tid >> 3, part = tid & 7;
    const int64_t h = (int64_t)blockIdx.x * 32 + g;
    if (h >= a.n_holes) return;
    const int32_t col = a.cols[h], c8 = col & ~7;
    const uint32_t row_bytes = (uint32_t)a.n * 8u;
    for (int r = 0; r < a.rows; ++r) {
        double *row = a.out + (a.row0 + r) * a.n;
        double v = row[c8 + part];
        if (part == (col & 7)) v = 1.0 + r;
        put<AUX>(row, row_bytes, c8 + part, v);
    }
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int rows = argc > 2 ? atoi(argv[2]) : 128, launches = argc > 3 ? atoi(argv[3]) : 64;
    const int64_t total_rows = 4096;
    std::vector<int32_t> cols;
    uint32_t s = 12345;
    for (int64_t c = 0; c + 20 <= n; c += 20) { s = s * 1664525u + 1013904223u; cols.push_back((int32_t)(c + (s >> 16) % 20)); }
    const int32_t nh = (int32_t)cols.size();
    double *out; int32_t *d_cols;
    CK(hipMalloc(&out, (size_t)total_rows * n * 8)); CK(hipMemset(out, 0, (size_t)total_rows * n * 8));
    CK(hipMalloc(&d_cols, (size_t)nh * 4)); CK(hipMemcpy(d_cols, cols.data(), (size_t)nh * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("n=%lld holes=%d rows/launch=%d launches=%d (stores per launch %.2fM)\n", (long long)n, nh, rows, launches, (double)nh * rows / 1e6);
    for (int shape = 0; shape < 19; ++shape) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < launches; ++l) {
                Args a{out, d_cols, n, nh, rows, ((int64_t)l * rows * 7) % (total_rows - rows)};
                switch (shape) {
                case 0: hipLaunchKernelGGL((k_scatter<16, 0>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;
                case 1: hipLaunchKernelGGL((k_scatter<256, 0>), dim3((nh + 255) / 256), dim3(256), 0, 0, a); break;
                case 2: hipLaunchKernelGGL((k_scatter<64, 0>), dim3((nh + 63) / 64), dim3(256), 0, 0, a); break;
                case 3: hipLaunchKernelGGL((k_scatter<2, 0>), dim3((nh + 1) / 2), dim3(256), 0, 0, a); break;
                case 4: hipLaunchKernelGGL((k_scatter<16, 2>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;
                case 5: hipLaunchKernelGGL((k_scatter<256, 2>), dim3((nh + 255) / 256), dim3(256), 0, 0, a); break;
                case 6: hipLaunchKernelGGL((k_scatter<16, 16>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;
                case 7: hipLaunchKernelGGL((k_scatter<256, 16>), dim3((nh + 255) / 256), dim3(256), 0, 0, a); break;
                case 10: hipLaunchKernelGGL((k_scatter<16, 1>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;      // sc0
                case 11: hipLaunchKernelGGL((k_scatter<16, 3>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;      // sc0 nt
                case 12: hipLaunchKernelGGL((k_scatter<16, 17>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;     // sc0 sc1
                case 13: hipLaunchKernelGGL((k_scatter<16, 18>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;     // nt sc1
                case 14: hipLaunchKernelGGL((k_scatter<16, 19>), dim3((nh + 15) / 16), dim3(256), 0, 0, a); break;     // sc0 nt sc1
                case 15: hipLaunchKernelGGL((k_pairs<0>), dim3((nh / 2 + 15) / 16), dim3(256), 0, 0, a); break;        // two holes in one 64-byte line (half the lines, the same stores)
                case 16: hipLaunchKernelGGL((k_pairs<1>), dim3((nh / 2 + 15) / 16), dim3(256), 0, 0, a); break;        // ... in one 128-byte line, different 64-byte halves
                case 17: hipLaunchKernelGGL((k_full<0>), dim3((nh + 63) / 64), dim3(256), 0, 0, a); break;      // whole 64-byte lines written, nothing read
                case 18: hipLaunchKernelGGL((k_full<2>), dim3((nh + 63) / 64), dim3(256), 0, 0, a); break;      // ... non-temporal
                case 8: hipLaunchKernelGGL((k_rmw<0>), dim3((nh + 31) / 32), dim3(256), 0, 0, a); break;
                case 9: hipLaunchKernelGGL((k_rmw<2>), dim3((nh + 31) / 32), dim3(256), 0, 0, a); break;
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("shape %d: %8.1f us per launch, %6.1f G stores/s\n", shape, ms * 1e3 / launches, (double)nh * rows * launches / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
