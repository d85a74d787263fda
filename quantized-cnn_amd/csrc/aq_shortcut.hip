// Shortcut of the dimension-increasing blocks: nn.AvgPool2d(2) followed by nn.ZeroPad2d((0, 0, 0, 0,
// C/2, C/2)), i.e. 2x2 average pooling and zero channels in front of and behind the pooled ones
// (adp_qtz_fan.py:669-671), and its gradient.  One thread per output element; pure streaming.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

__global__ __launch_bounds__(256) void shortcut_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int N, int C, int H, int Wd,
                                                           int front, int Co, int Ho, int Wo)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)N * Co * Ho * Wo) return;
    const int ox = t % Wo;
    long long r = t / Wo;
    const int oy = r % Ho;
    r /= Ho;
    const int co = r % Co, n = r / Co;
    const int c = co - front;
    float v = 0.0f;
    if (c >= 0 && c < C) {
        const float *p = x + (((long long)n * C + c) * H + 2 * oy) * Wd + 2 * ox;
        v = (((p[0] + p[1]) + p[Wd]) + p[Wd + 1]) * 0.25f;  // summation order of torch's avg_pool2d
    }
    y[t] = v;
}

__global__ __launch_bounds__(256) void shortcut_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int N, int C, int H, int Wd,
                                                           int front, int Co, int Ho, int Wo)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)N * C * H * Wd) return;
    const int ix = t % Wd;
    long long r = t / Wd;
    const int iy = r % H;
    r /= H;
    const int c = r % C, n = r / C;
    const int oy = iy >> 1, ox = ix >> 1;
    float v = 0.0f;
    if (oy < Ho && ox < Wo) v = 0.25f * dy[(((long long)n * Co + c + front) * Ho + oy) * Wo + ox];
    dx[t] = v;
}

static int shortcut_check(const char *fn, int N, int C, int H, int Wd, int front, int back)
{
    AQ_REQUIRE(N >= 0 && C >= 1 && H >= 2 && Wd >= 2 && front >= 0 && back >= 0, -EINVAL, "%s: bad sizes N=%d C=%d H=%d W=%d pad=%d,%d", fn, N, C, H,
               Wd, front, back);
    AQ_REQUIRE((long long)N * (C + front + back) * H * Wd < (1LL << 38), -E2BIG, "%s: tensor too large", fn);
    return 0;
}

} // namespace aq

extern "C" {

int aq_shortcut_fwd(const float *x, float *y, int N, int C, int H, int Wd, int pad_front, int pad_back, void *stream)
{
    using namespace aq;
    if (int rc = shortcut_check("aq_shortcut_fwd", N, C, H, Wd, pad_front, pad_back)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(x && y, -EFAULT, "aq_shortcut_fwd: NULL pointer");
    const int Co = C + pad_front + pad_back, Ho = H / 2, Wo = Wd / 2;
    const long long total = (long long)N * Co * Ho * Wo;
    hipLaunchKernelGGL(shortcut_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, C, H, Wd,
                       pad_front, Co, Ho, Wo);
    return check_launch("aq_shortcut_fwd");
}

int aq_shortcut_bwd(const float *dy, float *dx, int N, int C, int H, int Wd, int pad_front, int pad_back, void *stream)
{
    using namespace aq;
    if (int rc = shortcut_check("aq_shortcut_bwd", N, C, H, Wd, pad_front, pad_back)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(dy && dx, -EFAULT, "aq_shortcut_bwd: NULL pointer");
    const int Co = C + pad_front + pad_back, Ho = H / 2, Wo = Wd / 2;
    const long long total = (long long)N * C * H * Wd;
    hipLaunchKernelGGL(shortcut_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, dx, N, C, H, Wd,
                       pad_front, Co, Ho, Wo);
    return check_launch("aq_shortcut_bwd");
}

} // extern "C"
