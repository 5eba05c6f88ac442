// microbenchmark: scattered 64-byte (and 128-byte) record copies vs streaming copy, 1M records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
// each record REC doubles; LPR lanes per record (REC doubles / LPR per lane = 1)
template<int REC> __global__ void gather_rec(const double* __restrict__ src, double* __restrict__ dst, const int* __restrict__ idx, long nrec, int rows) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one lane per double
    long rec = g / REC; int w = g % REC;
    if (rec >= nrec) return;
    int s = idx[rec];
    for (int r = 0; r < rows; ++r)
        dst[(long)r * nrec * REC + rec * REC + w] = src[(long)r * nrec * REC + (long)s * REC + w];
}
template<int REC> __global__ void scatter_rec(const double* __restrict__ src, double* __restrict__ dst, const int* __restrict__ idx, long nrec, int rows) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long rec = g / REC; int w = g % REC;
    if (rec >= nrec) return;
    int s = idx[rec];
    for (int r = 0; r < rows; ++r)
        dst[(long)r * nrec * REC + (long)s * REC + w] = src[(long)r * nrec * REC + rec * REC + w];
}
int main() {
    const long nrec = 1000000; const int rows = 4;   // rows = independent chunks (each nrec records)
    std::vector<int> perm(nrec); for (long i = 0; i < nrec; ++i) perm[i] = (int)i;
    std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
    int* d_idx; CK(hipMalloc(&d_idx, nrec * 4)); CK(hipMemcpy(d_idx, perm.data(), nrec * 4, hipMemcpyHostToDevice));
    for (int REC : {4, 8, 16, 32}) {
        size_t bytes = (size_t)rows * nrec * REC * 8;
        double *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            long threads = nrec * REC; dim3 g((unsigned)((threads + 255) / 256));
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (REC == 4) { if (mode == 0) gather_rec<4><<<g, 256>>>(a, b, d_idx, nrec, rows); else scatter_rec<4><<<g, 256>>>(a, b, d_idx, nrec, rows); }
                if (REC == 8) { if (mode == 0) gather_rec<8><<<g, 256>>>(a, b, d_idx, nrec, rows); else scatter_rec<8><<<g, 256>>>(a, b, d_idx, nrec, rows); }
                if (REC == 16) { if (mode == 0) gather_rec<16><<<g, 256>>>(a, b, d_idx, nrec, rows); else scatter_rec<16><<<g, 256>>>(a, b, d_idx, nrec, rows); }
                if (REC == 32) { if (mode == 0) gather_rec<32><<<g, 256>>>(a, b, d_idx, nrec, rows); else scatter_rec<32><<<g, 256>>>(a, b, d_idx, nrec, rows); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("record %3d B  %s  %.3f ms  -> %.2f TB/s (read+write)\n", REC * 8, mode == 0 ? "gather " : "scatter", best, 2.0 * bytes / best / 1e9);
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
