// What the host-pointer boundary can hope for on this box: pageable vs registered vs staged-through-pinned copies, one way
// and both ways at once.    hipcc --offload-arch=gfx950 -O3 -pthread host_copy.hip -o /tmp/host_copy && /tmp/host_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char *dst, const char *src, size_t bytes, int threads)
{
    std::vector<std::thread> pool;
    const size_t per = (bytes / threads + 4095) / 4096 * 4096;
    for (int t = 0; t < threads; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= bytes) break;
        pool.emplace_back([=] { memcpy(dst + o, src + o, std::min(per, bytes - o)); });
    }
    for (auto &th : pool) th.join();
}
int main()
{
    const size_t bytes = (size_t)2 << 30;
    char *h_in = (char *)malloc(bytes), *h_out = (char *)malloc(bytes);
    memset(h_in, 1, bytes); memset(h_out, 2, bytes);
    char *d_a, *d_b;
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    double t0 = now();
    CK(hipMemcpy(d_a, h_in, bytes, hipMemcpyHostToDevice));
    printf("pageable H2D            %6.1f GB/s\n", bytes / (now() - t0) / 1e9);
    t0 = now();
    CK(hipMemcpy(h_out, d_a, bytes, hipMemcpyDeviceToHost));
    printf("pageable D2H            %6.1f GB/s\n", bytes / (now() - t0) / 1e9);
    t0 = now();
    CK(hipHostRegister(h_in, bytes, hipHostRegisterDefault));
    const double treg = now() - t0;
    CK(hipHostRegister(h_out, bytes, hipHostRegisterDefault));
    printf("hipHostRegister         %6.1f GB/s (%.0f ms per GB)\n", bytes / treg / 1e9, treg / (bytes / 1e9) * 1e3);
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        printf("registered H2D          %6.1f GB/s\n", bytes / (now() - t0) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        printf("registered D2H          %6.1f GB/s\n", bytes / (now() - t0) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        printf("registered both ways    %6.1f GB/s each\n", bytes / (now() - t0) / 1e9);
    }
    t0 = now();
    CK(hipHostUnregister(h_in)); CK(hipHostUnregister(h_out));
    printf("hipHostUnregister x2    %6.0f ms\n", (now() - t0) * 1e3);
    char *p_in, *p_out;
    CK(hipHostMalloc(&p_in, bytes / 8)); CK(hipHostMalloc(&p_out, bytes / 8));
    for (int th : {1, 2, 4, 8, 16}) {
        t0 = now();
        for (int k = 0; k < 8; ++k) par_copy(p_in, h_in + (size_t)k * (bytes / 8), bytes / 8, th);
        printf("memcpy pageable->pinned %6.1f GB/s with %d threads\n", bytes / (now() - t0) / 1e9, th);
    }
    t0 = now();
    for (int k = 0; k < 8; ++k) { CK(hipMemcpyAsync(d_a, p_in, bytes / 8, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
    printf("pinned H2D (256 MB)     %6.1f GB/s\n", bytes / (now() - t0) / 1e9);
    return 0;
}
