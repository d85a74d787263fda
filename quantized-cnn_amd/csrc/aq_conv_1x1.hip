// 1x1 convolutions (the projection shortcuts of ResNet-18-style networks: stride 2, no padding) as a plain GEMM on the
// fp32 MFMA: forward y[n][co][oy][ox] = sum_ci W[co][ci] x[n][ci][s oy][s ox] and the input gradient
// dx[n][ci][s oy][s ox] = sum_co W[co][ci] dy[n][co][oy][ox], zero at the positions no output reads.  The reference's
// convolution is generic in kernel size and stride (adp_qtz_fan.py:574-601); its own models hold no 1x1 layer.
//
// No tap shares an input, so nothing is staged through LDS: a wavefront owns 16 pixels x 16*MB output channels, walks the
// reduction channels in blocks of 16 and fetches its MFMA fragments straight from memory one block ahead of the matrix work.
// Lane (i = lane % 16, kq = lane / 16) takes the channels k0 + 4 kq + j (j = 0..3) of a block for the j-th of its four
// MFMAs (the order inside the sum is free as long as A and B agree), so that a forward lane's four weights are one
// 16-byte load.  The weights (<= 512 KiB) stay in L2; every input element is read once per group of 16*MB channels.
// The input gradient writes the whole s x s cell of its pixel: the value and zeros (or the tensor to add), so no
// separate clear of dx runs.
#include "aq_conv_shared.h"

namespace aq {

struct C1Args {
    const float *in, *W, *res;
    float *out;
    int M, K;        // output / reduction channels of the GEMM
    int GH, GW;      // pixel grid of the GEMM: the convolution's output (forward), dy (input gradient)
    int SH, SW, s;   // the strided tensor's spatial size (x forward, dx for the input gradient) and the stride
    long long P;     // N * GH * GW
};

template <int MB, bool DGRAD>
__global__ __launch_bounds__(256) void conv1x1_kernel(const C1Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * (16 * MB);
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * 16;
    if (p0 >= a.P) return;
    // this lane's pixel as a B operand (clamped: lanes beyond the last pixel load a valid address and store nothing)
    const long long pb = p0 + i < a.P ? p0 + i : a.P - 1;
    const int gsz = a.GH * a.GW;
    const long long n = pb / gsz;
    const int rem = (int)(pb - n * gsz), gy = rem / a.GW, gx = rem - gy * a.GW;
    const long long kstride = DGRAD ? (long long)gsz : (long long)a.SH * a.SW;
    const float *bsrc = DGRAD ? a.in + (n * a.K * (long long)gsz + rem)
                              : a.in + ((n * a.K * (long long)a.SH + (long long)a.s * gy) * a.SW + (long long)a.s * gx);
    bsrc += (long long)(4 * kq) * kstride;
    // A operand: forward W[m][k] (row-major in k), input gradient W[k][m]
    const float *asrc = DGRAD ? a.W + ((long long)(4 * kq) * a.M + m0 + i) : a.W + ((long long)(m0 + i) * a.K + 4 * kq);

    floatx4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    float bc[4], bn[4];
    floatx4 ac[MB], an[MB];
    auto fetch = [&](int k0, float (&b)[4], floatx4 (&av)[MB]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bsrc[(long long)(k0 + j) * kstride];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            if (DGRAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[mb][j] = asrc[(long long)(k0 + j) * a.M + 16 * mb];
            } else {
                av[mb] = *reinterpret_cast<const floatx4 *>(asrc + (long long)(16 * mb) * a.K + k0);
            }
        }
    };
    fetch(0, bc, ac);
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const bool more = k0 + 16 < a.K;
        if (more) fetch(k0 + 16, bn, an);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mb][j], bc[j], acc[mb], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bc[j] = bn[j];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) ac[mb] = an[mb];
        }
    }

    // accumulator r of block mb: channel m0 + 16 mb + 4 kq + r, pixel p0 + i
    if (p0 + i >= a.P) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mb + 4 * kq + r;
            const float v = acc[mb][r];
            if (!DGRAD) {
                const long long o = (n * a.M + m) * (long long)gsz + rem;
                a.out[o] = a.res ? v + a.res[o] : v;
            } else {
                const int y0 = a.s * gy, x0 = a.s * gx;
                const long long o = ((n * a.M + m) * (long long)a.SH + y0) * a.SW + x0;
                // cells of the last grid row / column also own the rows / columns behind them that no output reads
                const int ye = gy == a.GH - 1 ? a.SH : y0 + a.s, xe = gx == a.GW - 1 ? a.SW : x0 + a.s;
                for (int yy = y0; yy < ye; ++yy)
                    for (int xx = x0; xx < xe; ++xx) {
                        const long long oo = o + (long long)(yy - y0) * a.SW + (xx - x0);
                        const float base = a.res ? a.res[oo] : 0.0f;
                        a.out[oo] = (yy == y0 && xx == x0) ? v + base : base;
                    }
            }
        }
}

template <bool DGRAD>
static int launch_1x1(const ConvArgs &c, int N, hipStream_t st, const char *who, bool *taken)
{
    *taken = false;
    // forward: M = Cout, K = Cin, grid = output; input gradient: M = Cin, K = Cout, grid = dy
    const int s = DGRAD ? c.dil : c.rs;
    if (c.ks != 1 || c.padT != 0 || (s != 1 && s != 2) || c.M % 16 || c.KC % 16 || c.idx || c.bn_on || c.stat_part || c.bwd_part) return 0;
    if (!DGRAD && ((uintptr_t)c.W & 15)) return 0;
    C1Args a;
    a.in = c.in; a.W = c.W; a.res = c.res; a.out = c.out;
    a.M = c.M; a.K = c.KC; a.s = s;
    if (DGRAD) {
        a.GH = c.IH; a.GW = c.IW; a.SH = c.OH; a.SW = c.OW;
    } else {
        a.GH = c.OH; a.GW = c.OW; a.SH = c.IH; a.SW = c.IW;
    }
    a.P = (long long)N * a.GH * a.GW;
    const long long gx = (a.P + 63) / 64;
    if (gx >= (1LL << 31)) return 0;
    *taken = true;
    const int MB = c.M % 64 == 0 ? 4 : (c.M % 32 == 0 ? 2 : 1);
    dim3 grid((unsigned)gx, (unsigned)(c.M / (16 * MB)));
    if (MB == 4) hipLaunchKernelGGL((conv1x1_kernel<4, DGRAD>), grid, dim3(256), 0, st, a);
    else if (MB == 2) hipLaunchKernelGGL((conv1x1_kernel<2, DGRAD>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv1x1_kernel<1, DGRAD>), grid, dim3(256), 0, st, a);
    return check_launch(who);
}

}  // namespace aq

int aq_c1_launch_fwd(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken) { return aq::launch_1x1<false>(a, N, st, who, taken); }
int aq_c1_launch_dgrad(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken) { return aq::launch_1x1<true>(a, N, st, who, taken); }
