// microbenchmark: 128-byte records gathered / scattered over a LARGE span (like the record-mode ring:
// 1M positions x `chunks` chunks of 128 MB; every record of a batch lands in chunk base + lag/16 + k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void k(const double* __restrict__ lin, double* __restrict__ ring, const int* __restrict__ pos, const int* __restrict__ chunk, long nrec, long n, int chunks, int scatter) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long rec = g / 8; int part = g % 8;           // 8 lanes x 16 B per record
    if (rec >= nrec) return;
    long col = rec / 4; int kk = rec % 4;
    long addr = (((long)(chunk[col] + kk) % chunks) * n + pos[col]) * 16 + 2 * part;
    if (scatter) *reinterpret_cast<double2*>(ring + addr) = *reinterpret_cast<const double2*>(lin + rec * 16 + 2 * part);
    else *reinterpret_cast<double2*>(const_cast<double*>(lin) + rec * 16 + 2 * part) = *reinterpret_cast<const double2*>(ring + addr);
}
int main() {
    const long n = 1000000; const long nrec = 4 * n;
    std::vector<int> pos(n), ch(n); std::mt19937 rng(1);
    for (long i = 0; i < n; ++i) pos[i] = (int)i; std::shuffle(pos.begin(), pos.end(), rng);
    int *d_pos, *d_ch; CK(hipMalloc(&d_pos, n * 4)); CK(hipMalloc(&d_ch, n * 4)); CK(hipMemcpy(d_pos, pos.data(), n * 4, hipMemcpyHostToDevice));
    double* lin; CK(hipMalloc(&lin, nrec * 128)); CK(hipMemset(lin, 0, nrec * 128));
    for (int spread : {1, 8, 32, 120}) {
        int chunks = spread + 8;
        double* ring; CK(hipMalloc(&ring, (size_t)chunks * n * 128)); CK(hipMemset(ring, 0, (size_t)chunks * n * 128));
        for (long i = 0; i < n; ++i) ch[i] = (int)(((long)pos[i] * spread) / n);   // chunk offset grows with position like lag/16
        CK(hipMemcpy(d_ch, ch.data(), n * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int scatter = 0; scatter < 2; ++scatter) {
            float best = 1e9;
            for (int it = 0; it < 5; ++it) {
                (void)hipEventRecord(e0);
                k<<<(unsigned)((nrec * 8 + 255) / 256), 256>>>(lin, ring, d_pos, d_ch, nrec, n, chunks, scatter);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("span %3d chunks (%5.1f GB)  %s  %.3f ms -> %.2f TB/s (read+write of 512 MB each)\n", chunks, chunks * n * 128 / 1e9, scatter ? "scatter" : "gather ", best, 2.0 * nrec * 128 / best / 1e9);
        }
        (void)hipFree(ring);
    }
    return 0;
}
