// microbenchmark: what shapes of store reach the rate the runtime's memset shows (6.7 TB/s) -- the engine's stores (records in
// k_rec_in, rows in k_rec_out, records in k_tile) run at 5.1-5.6 TB/s on their own (rec_probe.hip), its loads at 6.2-7.0.
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/write_probe.hip -o profiles/microbench/write_probe.bin && ./write_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ void st(v2 *p, v2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// one 16-byte store per thread, lane-linear
template <int NT> __global__ __launch_bounds__(256) void w_flat(v2 *dst, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) st<NT>(dst + i, v2{1.0, 2.0});
}
// U stores per thread; the workgroup covers U * 4 KiB contiguous, every instruction lane-linear
template <int U, int NT> __global__ __launch_bounds__(256) void w_flat_u(v2 *dst, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < count) st<NT>(dst + i + u * 256, v2{1.0, 2.0});
}
// each thread writes 64 contiguous bytes (4 stores): an instruction covers 64 x 16 B at a 64-byte stride
template <int NT> __global__ __launch_bounds__(256) void w_thread64(v2 *dst, int64_t count)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u < count) st<NT>(dst + i + u, v2{1.0, 2.0});
}
// persistent grid-stride
template <int U, int NT> __global__ __launch_bounds__(256) void w_stride(v2 *dst, int64_t count)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < count; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(dst + i + u * stride, v2{1.0, 2.0});
    }
}
// 8-byte stores, lane-linear (the out-pass writes doubles)
template <int NT> __global__ __launch_bounds__(256) void w_flat8(double *dst, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) { if (NT) __builtin_nontemporal_store(1.0, dst + i); else dst[i] = 1.0; }
}
// eight 128-byte records per instruction, in PLANES planes (the in-pass: 8 lanes per record, consecutive records of a column in
// consecutive planes); plane stride = count / PLANES elements
template <int PLANES, int NT> __global__ __launch_bounds__(256) void w_planes(v2 *dst, int64_t count)
{
    const int64_t per = count / PLANES;      // 16-byte elements per plane
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rec = g >> 3;              // (column, plane) pairs, plane fastest
    const int part = (int)(g & 7);
    const int64_t col = rec / PLANES;
    const int plane = (int)(rec % PLANES);
    const int64_t i = (int64_t)plane * per + col * 8 + part;
    if (col * 8 + 7 < per) st<NT>(dst + i, v2{1.0, 2.0});
}
// buffer stores with cache-policy bits: aux 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1, 3 sc0 nt
template <int AUX> __global__ __launch_bounds__(256) void w_buffer(v2 *dst, int64_t count)
{
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst + i0, 0, 4096, 0x00020000);
    const u4 bits = {1u, 2u, 3u, 4u};
    if (i0 + 255 < count) __builtin_amdgcn_raw_buffer_store_b128(bits, r, (int)threadIdx.x * 16, 0, AUX);
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    template <typename F> float best(F f, int reps = 5)
    {
        float b = 1e30f;
        f();
        for (int r = 0; r < reps; ++r) {
            (void)hipEventRecord(e0, nullptr); f(); (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); b = std::min(b, ms);
        }
        return b;
    }
};

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const int64_t bytes = (int64_t)(gib * (1 << 30)), count = bytes / 16;
    v2 *b; CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes));
    Timer T;
    printf("write-only, %.1f GiB buffer\n", gib);
#define ROW(name, call) do { const float ms = T.best([&] { call; }); printf("  %-64s %8.3f ms  %6.3f TB/s\n", name, ms, (double)bytes / (ms * 1e-3) / 1e12); fflush(stdout); } while (0)
    const unsigned gf = (unsigned)((count + 255) / 256);
    ROW("hipMemsetAsync", (void)hipMemsetAsync(b, 0, bytes, nullptr));
    ROW("flat, 16 B per thread", (w_flat<0><<<gf, 256>>>(b, count)));
    ROW("flat, 16 B per thread, nt", (w_flat<1><<<gf, 256>>>(b, count)));
    ROW("flat, 2 x 16 B per thread (workgroup = 8 KiB)", (w_flat_u<2, 0><<<gf / 2, 256>>>(b, count)));
    ROW("flat, 4 x 16 B per thread (workgroup = 16 KiB)", (w_flat_u<4, 0><<<gf / 4, 256>>>(b, count)));
    ROW("flat, 4 x 16 B per thread, nt", (w_flat_u<4, 1><<<gf / 4, 256>>>(b, count)));
    ROW("flat, 16 x 16 B per thread (workgroup = 64 KiB)", (w_flat_u<16, 0><<<gf / 16, 256>>>(b, count)));
    ROW("flat, thread writes 64 contiguous bytes", (w_thread64<0><<<gf / 4, 256>>>(b, count)));
    ROW("flat, 8 B per thread", (w_flat8<0><<<gf * 2, 256>>>((double *)b, count * 2)));
    ROW("flat, 8 B per thread, nt", (w_flat8<1><<<gf * 2, 256>>>((double *)b, count * 2)));
    ROW("persistent 1024 workgroups, 1 in flight", (w_stride<1, 0><<<1024, 256>>>(b, count)));
    ROW("persistent 4096 workgroups, 4 in flight", (w_stride<4, 0><<<4096, 256>>>(b, count)));
    ROW("flat, 8 records of 128 B per instruction in 8 planes", (w_planes<8, 0><<<gf, 256>>>(b, count)));
    ROW("flat, 8 records per instruction in 8 planes, nt", (w_planes<8, 1><<<gf, 256>>>(b, count)));
    ROW("flat, 8 records per instruction in 2 planes", (w_planes<2, 0><<<gf, 256>>>(b, count)));
    ROW("flat, 8 records per instruction in 1 plane (lane-linear)", (w_planes<1, 0><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, plain", (w_buffer<0><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, sc0", (w_buffer<1><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, nt", (w_buffer<2><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, sc0 nt", (w_buffer<3><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, sc1", (w_buffer<16><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, sc0 sc1", (w_buffer<17><<<gf, 256>>>(b, count)));
    ROW("flat buffer store, sc1 nt", (w_buffer<18><<<gf, 256>>>(b, count)));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    return 0;
}
