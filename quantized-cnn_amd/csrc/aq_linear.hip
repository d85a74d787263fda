// MyLinear (adp_qtz_fan.py:646-647): y = x @ W + bias with W stored (in, out), and its gradients.
// The layer is tiny (batch x 64 x 10 in ResNet-20) and latency-bound: the kernels spread every
// reduction over a wavefront (lanes stride over the reduced index, xor-tree at the end) so that no
// thread walks a long dependent chain of loads.  Fixed summation order, no atomics.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

__device__ inline float lin_wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// one wavefront per batch row n: lanes hold x[n, i] (i = lane, lane+64, ..) and reduce every output
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ b, float *__restrict__ y, int N, int In, int Out)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int o = 0; o < Out; ++o) {
        float a = 0.0f;
        for (int i = lane; i < In; i += 64) a = __builtin_fmaf(x[(size_t)n * In + i], W[(size_t)i * Out + o], a);
        a = lin_wave_sum(a);
        if (lane == 0) y[(size_t)n * Out + o] = a + (b ? b[o] : 0.0f);
    }
}

// one launch, three jobs selected by block range:
//   dx   [N*In]    one thread per element (Out terms each)
//   dW   [In*Out]  one wavefront per element, lanes stride over the batch
//   db   [Out]     one wavefront per element
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ dy, float *__restrict__ dx,
                                                         float *__restrict__ dW, float *__restrict__ db, int N, int In, int Out,
                                                         int blocks_dx, int blocks_dw)
{
    int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (b < blocks_dx) {
        const int t = b * 256 + threadIdx.x;
        if (t >= N * In) return;
        const int n = t / In, i = t % In;
        float a = 0.0f;
        for (int o = 0; o < Out; ++o) a = __builtin_fmaf(dy[(size_t)n * Out + o], W[(size_t)i * Out + o], a);
        dx[t] = a;
        return;
    }
    b -= blocks_dx;
    if (b < blocks_dw) {
        const int t = b * 4 + wave;
        if (t >= In * Out) return;
        const int i = t / Out, o = t % Out;
        float a = 0.0f;
        for (int n = lane; n < N; n += 64) a = __builtin_fmaf(x[(size_t)n * In + i], dy[(size_t)n * Out + o], a);
        a = lin_wave_sum(a);
        if (lane == 0) dW[t] = a;
        return;
    }
    b -= blocks_dw;
    const int o = b * 4 + wave;
    if (o >= Out) return;
    float a = 0.0f;
    for (int n = lane; n < N; n += 64) a += dy[(size_t)n * Out + o];
    a = lin_wave_sum(a);
    if (lane == 0) db[o] = a;
}

} // namespace aq

extern "C" {

int aq_linear_fwd(const float *x, const float *W, const float *bias, float *y, int N, int In, int Out, void *stream)
{
    AQ_REQUIRE(N >= 0 && In >= 1 && Out >= 1, -EINVAL, "aq_linear_fwd: bad sizes N=%d In=%d Out=%d", N, In, Out);
    AQ_REQUIRE((long long)N * Out < (1LL << 31) && (long long)N * In < (1LL << 31), -E2BIG, "aq_linear_fwd: too large");
    if (N == 0) return 0;
    AQ_REQUIRE(x && W && y, -EFAULT, "aq_linear_fwd: NULL pointer");
    hipLaunchKernelGGL(aq::linear_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, W, bias, y, N, In, Out);
    return aq::check_launch("aq_linear_fwd");
}

int aq_linear_bwd(const float *x, const float *W, const float *dy, float *dx, float *dW, float *dbias, int N, int In, int Out, void *stream)
{
    AQ_REQUIRE(N >= 0 && In >= 1 && Out >= 1, -EINVAL, "aq_linear_bwd: bad sizes N=%d In=%d Out=%d", N, In, Out);
    AQ_REQUIRE((long long)N * Out < (1LL << 31) && (long long)N * In < (1LL << 31), -E2BIG, "aq_linear_bwd: too large");
    AQ_REQUIRE(x && W && dy, -EFAULT, "aq_linear_bwd: NULL pointer");
    const int bdx = dx ? (N * In + 255) / 256 : 0, bdw = dW ? (In * Out + 3) / 4 : 0, bdb = dbias ? (Out + 3) / 4 : 0;
    if (bdx + bdw + bdb == 0) return 0;
    hipLaunchKernelGGL(aq::linear_bwd_kernel, dim3(bdx + bdw + bdb), dim3(256), 0, static_cast<hipStream_t>(stream), x, W, dy, dx, dW, dbias, N, In,
                       Out, bdx, bdw);
    return aq::check_launch("aq_linear_bwd");
}

} // extern "C"
