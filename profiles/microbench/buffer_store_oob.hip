#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef double v1d;
__global__ void k(double *out, const double *in, int n, int lo)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * 8, 0x00020000);
    const double v = in[p < n ? p : 0];
    const uint32_t off = (p >= lo && p < n) ? (uint32_t)p * 8u : 0xFFFFFFF0u;
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u2 bits; __builtin_memcpy(&bits, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(bits, r, off, 0, 0);
}
int main() {
    const int n = 1000;
    double *d_in, *d_out; hipMalloc(&d_in, n*8); hipMalloc(&d_out, n*8 + 4096);
    double h[1000]; for (int i=0;i<n;++i) h[i]=i+1; hipMemcpy(d_in,h,n*8,hipMemcpyHostToDevice); hipMemset(d_out,0,n*8+4096);
    k<<<4,256>>>(d_out,d_in,n,500); hipDeviceSynchronize();
    double o[1000]; hipMemcpy(o,d_out,n*8,hipMemcpyDeviceToHost);
    int bad=0; for(int i=0;i<n;++i){ double e = i>=500? i+1:0; if(o[i]!=e) ++bad; }
    printf("bad=%d o[499]=%g o[500]=%g o[999]=%g\n",bad,o[499],o[500],o[999]); return bad!=0;
}
