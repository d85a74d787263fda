// MyLinear (adp_qtz_fan.py:646-647): y = x @ W + bias with W stored (in, out), and its gradients.
// The layer is tiny (batch x 64 x 10 in ResNet-20), so these are plain one-thread-per-output
// kernels with a fixed summation order (ascending index), no atomics.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ b, float *__restrict__ y, int N, int In, int Out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * Out) return;
    const int n = t / Out, o = t % Out;
    float a = 0.0f;
    for (int i = 0; i < In; ++i) a = __builtin_fmaf(x[n * In + i], W[i * Out + o], a);
    y[t] = a + (b ? b[o] : 0.0f);
}

// one launch, three jobs selected by block range: dx (N*In outputs), dW (In*Out), dbias (Out)
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ dy, float *__restrict__ dx,
                                                         float *__restrict__ dW, float *__restrict__ db, int N, int In, int Out,
                                                         int blocks_dx, int blocks_dw)
{
    int b = blockIdx.x;
    if (b < blocks_dx) {
        const int t = b * 256 + threadIdx.x;
        if (t >= N * In) return;
        const int n = t / In, i = t % In;
        float a = 0.0f;
        for (int o = 0; o < Out; ++o) a = __builtin_fmaf(dy[n * Out + o], W[i * Out + o], a);
        dx[t] = a;
        return;
    }
    b -= blocks_dx;
    if (b < blocks_dw) {
        const int t = b * 256 + threadIdx.x;
        if (t >= In * Out) return;
        const int i = t / Out, o = t % Out;
        float a = 0.0f;
        for (int n = 0; n < N; ++n) a = __builtin_fmaf(x[n * In + i], dy[n * Out + o], a);
        dW[t] = a;
        return;
    }
    b -= blocks_dw;
    const int o = b * 256 + threadIdx.x;
    if (o >= Out) return;
    float a = 0.0f;
    for (int n = 0; n < N; ++n) a += dy[n * Out + o];
    db[o] = a;
}

} // namespace aq

extern "C" {

int aq_linear_fwd(const float *x, const float *W, const float *bias, float *y, int N, int In, int Out, void *stream)
{
    AQ_REQUIRE(N >= 0 && In >= 1 && Out >= 1, -EINVAL, "aq_linear_fwd: bad sizes N=%d In=%d Out=%d", N, In, Out);
    AQ_REQUIRE((long long)N * Out < (1LL << 31) && (long long)N * In < (1LL << 31), -E2BIG, "aq_linear_fwd: too large");
    if (N == 0) return 0;
    AQ_REQUIRE(x && W && y, -EFAULT, "aq_linear_fwd: NULL pointer");
    hipLaunchKernelGGL(aq::linear_fwd_kernel, dim3((N * Out + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, W, bias, y, N, In, Out);
    return aq::check_launch("aq_linear_fwd");
}

int aq_linear_bwd(const float *x, const float *W, const float *dy, float *dx, float *dW, float *dbias, int N, int In, int Out, void *stream)
{
    AQ_REQUIRE(N >= 0 && In >= 1 && Out >= 1, -EINVAL, "aq_linear_bwd: bad sizes N=%d In=%d Out=%d", N, In, Out);
    AQ_REQUIRE((long long)N * Out < (1LL << 31) && (long long)N * In < (1LL << 31), -E2BIG, "aq_linear_bwd: too large");
    AQ_REQUIRE(x && W && dy, -EFAULT, "aq_linear_bwd: NULL pointer");
    const int bdx = dx ? (N * In + 255) / 256 : 0, bdw = dW ? (In * Out + 255) / 256 : 0, bdb = dbias ? (Out + 255) / 256 : 0;
    if (bdx + bdw + bdb == 0) return 0;
    hipLaunchKernelGGL(aq::linear_bwd_kernel, dim3(bdx + bdw + bdb), dim3(256), 0, static_cast<hipStream_t>(stream), x, W, dy, dx, dW, dbias, N, In,
                       Out, bdx, bdw);
    return aq::check_launch("aq_linear_bwd");
}

} // extern "C"
