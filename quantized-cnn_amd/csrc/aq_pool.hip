// nn.MaxPool2d(3, 2, 1) of the ResNet-18 stem and its gradient (the reference's own models pool with AvgPool2d only,
// adp_qtz_fan.py:669-671, 716; this layer belongs to the torchvision layout of resnet18.py).  The forward pass keeps the
// position of each window's maximum as one byte (ky * 3 + kx, first maximum in scan order, NaN wins as in torch), so the
// gradient is a gather: an input pixel lies in at most four windows and sums the gradients of those that chose it, in a
// fixed order - no atomics, no index tensor of 8 bytes per element.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, uint8_t *__restrict__ idx, long long total,
                                                          int H, int W, int OH, int OW)
{
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int ox = (int)(o % OW);
    const long long t = o / OW;
    const int oy = (int)(t % OH);
    const long long nc = t / OH;
    const float *p = x + nc * (long long)H * W;
    float best = -__builtin_inff();
    int bi = -1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float v = p[(long long)iy * W + ix];
            if (bi < 0 || v > best || v != v) {  // (torch's rule: a NaN replaces anything, nothing replaces a NaN but a later NaN)
                best = v;
                bi = ky * 3 + kx;
            }
        }
    }
    y[o] = best;
    idx[o] = (uint8_t)bi;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ idx, float *__restrict__ dx,
                                                          long long total, int H, int W, int OH, int OW)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ix = (int)(i % W);
    const long long t = i / W;
    const int iy = (int)(t % H);
    const long long nc = t / H;
    const float *g = dy + nc * (long long)OH * OW;
    const uint8_t *c = idx + nc * (long long)OH * OW;
    // windows that contain (iy, ix): 2 oy - 1 <= iy <= 2 oy + 1
    const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;
    float s = 0.0f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= OH) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= OW) continue;
            const int k = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));
            const long long o = (long long)oy * OW + ox;
            if ((int)c[o] == k) s += g[o];
        }
    }
    dx[i] = s;
}

// The same two passes with one image plane per blockIdx.y (32-bit index arithmetic, no 64-bit divisions) and, in the
// gradient, four consecutive input pixels per thread (one 16-byte store; the up to six window gradients they can
// receive are loaded once).  Sums in the order of maxpool_bwd_kernel (window rows outer, window columns inner), so both
// forms give the same bits.  W a multiple of 4.
__global__ __launch_bounds__(256) void maxpool_fwd_plane_kernel(const float *__restrict__ x, float *__restrict__ y, uint8_t *__restrict__ idx, int H,
                                                                int W, int OH, int OW)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= OH * OW) return;
    const int oy = o / OW, ox = o - oy * OW;
    const float *p = x + (long long)blockIdx.y * H * W;
    float best = -__builtin_inff();
    int bi = -1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float v = p[iy * W + ix];
            if (bi < 0 || v > best || v != v) {
                best = v;
                bi = ky * 3 + kx;
            }
        }
    }
    const long long q = (long long)blockIdx.y * OH * OW + o;
    y[q] = best;
    idx[q] = (uint8_t)bi;
}

__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ idx, float *__restrict__ dx, int H,
                                                               int W, int OH, int OW)
{
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W4) return;
    const int iy = t / W4, ix0 = (t - iy * W4) << 2;
    const float *g = dy + (long long)blockIdx.y * OH * OW;
    const uint8_t *c = idx + (long long)blockIdx.y * OH * OW;
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, oxb = ix0 >> 1;  // window columns oxb .. oxb + 2 cover the four pixels
    float gv[2][3];
    int cv[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a == 0 ? oy0 : oy1;
        const bool rowok = oy < OH && (a == 0 || oy1 != oy0);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ox = oxb + b;
            const bool ok = rowok && ox < OW;
            gv[a][b] = ok ? g[oy * OW + ox] : 0.0f;
            cv[a][b] = ok ? (int)c[oy * OW + ox] : -1;
        }
    }
    // pixel ix0 + k lies in window columns oxb + blo[k] .. oxb + bhi[k] (ix0 is a multiple of 4)
    constexpr int blo[4] = {0, 0, 1, 1}, bhi[4] = {0, 1, 1, 2};
    float o4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ix = ix0 + k;
        float s = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a == 0 ? oy0 : oy1;
            if (a == 1 && oy1 == oy0) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b < blo[k] || b > bhi[k]) continue;
                const int kk = (iy - (2 * oy - 1)) * 3 + (ix - (2 * (oxb + b) - 1));
                if (cv[a][b] == kk) s += gv[a][b];  // (cv = -1 outside the plane)
            }
        }
        o4[k] = s;
    }
    typedef float pool_f4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<pool_f4 *>(dx + (long long)blockIdx.y * H * W + iy * W + ix0) = pool_f4{o4[0], o4[1], o4[2], o4[3]};
}

} // namespace aq

extern "C" {

int aq_maxpool3s2_fwd(const float *x, float *y, uint8_t *idx, int N, int C, int H, int Wd, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 0 && C >= 1 && H >= 1 && Wd >= 1, -EINVAL, "aq_maxpool3s2_fwd: bad sizes");
    if (N == 0) return 0;
    AQ_REQUIRE(x && y && idx, -EFAULT, "aq_maxpool3s2_fwd: NULL pointer");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (Wd + 2 - 3) / 2 + 1;
    const long long total = (long long)N * C * OH * OW;
    AQ_REQUIRE((total + 255) / 256 < (1LL << 31), -E2BIG, "aq_maxpool3s2_fwd: tensor too large");
    if ((long long)N * C <= 65535 && (long long)H * Wd < (1LL << 30)) {
        hipLaunchKernelGGL(maxpool_fwd_plane_kernel, dim3((unsigned)((OH * OW + 255) / 256), (unsigned)(N * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, y, idx, H, Wd, OH, OW);
        return check_launch("aq_maxpool3s2_fwd");
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, idx, total, H, Wd,
                       OH, OW);
    return check_launch("aq_maxpool3s2_fwd");
}

int aq_maxpool3s2_bwd(const float *dy, const uint8_t *idx, float *dx, int N, int C, int H, int Wd, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 0 && C >= 1 && H >= 1 && Wd >= 1, -EINVAL, "aq_maxpool3s2_bwd: bad sizes");
    if (N == 0) return 0;
    AQ_REQUIRE(dy && dx && idx, -EFAULT, "aq_maxpool3s2_bwd: NULL pointer");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (Wd + 2 - 3) / 2 + 1;
    const long long total = (long long)N * C * H * Wd;
    AQ_REQUIRE((total + 255) / 256 < (1LL << 31), -E2BIG, "aq_maxpool3s2_bwd: tensor too large");
    if ((Wd & 3) == 0 && (long long)N * C <= 65535 && (long long)H * Wd < (1LL << 30) && ((uintptr_t)dx & 15) == 0) {
        hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3((unsigned)((H * (Wd / 4) + 255) / 256), (unsigned)(N * C)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), dy, idx, dx, H, Wd, OH, OW);
        return check_launch("aq_maxpool3s2_bwd");
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, idx, dx, total, H, Wd,
                       OH, OW);
    return check_launch("aq_maxpool3s2_bwd");
}

} // extern "C"
