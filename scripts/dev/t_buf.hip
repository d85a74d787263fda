// Development aid: semantics of raw buffer b128 loads (range check) and umulhi on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *p, float *o, unsigned *q, int nbytes, unsigned rcp)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16u;
    if (threadIdx.x == 5) off = 0x80000000u;
    auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = __builtin_bit_cast(float, t[i]);
    q[threadIdx.x] = __umulhi(threadIdx.x * 7u, rcp);
}
int main()
{
    float h[1024], *d, *o;
    unsigned *q, hq[64];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096); hipMalloc(&q, 256);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    unsigned rcp = (unsigned)(((1ULL << 32) + 39) / 40);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, q, 600, rcp);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hq, q, 256, hipMemcpyDeviceToHost);
    for (int t = 0; t < 64; t += 1) printf("t%d: %g %g %g %g  q=%u (want %u)\n", t, h[4 * t], h[4 * t + 1], h[4 * t + 2], h[4 * t + 3], hq[t], t * 7 / 40);
    return 0;
}
