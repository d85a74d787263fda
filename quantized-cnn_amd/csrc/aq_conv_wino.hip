// 3x3 / stride 1 / padding 1 convolutions with many channels by the minimal-filtering form F(2x2, 3x3) (Lavin & Gray
// 2016): forward and input gradient of layers whose channel counts are multiples of 16 (adp_qtz_fan.py:574-601 computes
// the same sums tap by tap).  An output tile of 2 x 2 pixels takes 16 multiplications per (input, output) channel pair
// instead of 36, so the matrix cores do 2.25 times less work than in conv3x3_c16_kernel; the transforms are additions:
//   V = B^T d B   of every 4 x 4 input patch d (stride 2 between patches), per input channel     [all wavefronts, via LDS]
//   U = G g G^T   of every 3 x 3 filter g, per channel pair                                      [registers, per lane]
//   M[comp] = sum over input channels U[comp] * V[comp], comp = 0..15                            [fp32 MFMA 16x16x4]
//   Y = A^T M A   (2 x 2 outputs per tile)                                                       [registers, per lane]
// A workgroup owns 64 output channels (16 per wavefront) x a block of up to 16 tiles of one image; input channels pass
// in chunks of 16 through an LDS image [pixel][channel] (20 floats per pixel, as in the direct kernel), double
// buffered, the next chunk fetched into registers during the matrix work.  The transformed patches V[comp][tile][channel]
// are the B fragments (one 16-byte LDS read per lane and component); a lane's 36 filter values of a chunk become its
// 64 A values in registers.  The 16 components of a (channel, tile) pair end up in ONE lane, so the output transform
// needs no exchange.  fp32 throughout; sums in another order than the direct form (a few 1e-7 relative), no atomics:
// bit-reproducible run to run.
#include "aq_conv_shared.h"

#include <algorithm>

namespace aq {

constexpr int WN_CS = 20;
constexpr int WN_MAXIT = 3;  // staging items (pixel, group of 4 channels) per thread: 4 * plane <= 256 * WN_MAXIT

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int ty0 = ty * a.TH, tx0 = tx * a.TW;  // first tile of the block (a.TH x a.TW tiles)
    const int m0 = blockIdx.y * 64 + wave * 16;
    const int bufsz = a.plane * WN_CS;
    float *Vs = smem + 2 * bufsz;                // [16 components][16 tiles][WN_CS]
    const int chs = a.IH * a.IW;
    const float *img = a.in + (long long)n * a.KC * chs;

    // staging of the image chunk: item = (group of 4 channels, pixel of the block's input region)
    int goff[WN_MAXIT], loff[WN_MAXIT];
    {
        const int r0 = 2 * ty0 - a.padT, c0 = 2 * tx0 - a.padT;
#pragma unroll
        for (int it = 0; it < WN_MAXIT; ++it) {
            const int i = tid + 256 * it;
            goff[it] = -2;
            loff[it] = 0;
            if (i < 4 * a.plane) {
                const int cg = i / a.plane, e = i - cg * a.plane;
                const int r = e / a.pitch, cc = e - r * a.pitch;
                const int gy = r0 + r, gx = c0 + cc;
                const bool ok = gy >= 0 && gx >= 0 && gy < a.IH && gx < a.IW;
                goff[it] = ok ? 4 * cg * chs + gy * a.IW + gx : -1;
                loff[it] = e * WN_CS + 4 * cg;
            }
        }
    }
    auto stage_load = [&](int cbase, f4 (&xr)[WN_MAXIT]) {
        const float *src = img + (long long)cbase * chs;
#pragma unroll
        for (int it = 0; it < WN_MAXIT; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            const float *s0 = src + o;
            const f4 v = f4{s0[0], s0[chs], s0[2 * chs], s0[3 * chs]};
            xr[it] = goff[it] >= 0 ? v : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](float *buf, const f4 (&xr)[WN_MAXIT]) {
#pragma unroll
        for (int it = 0; it < WN_MAXIT; ++it)
            if (goff[it] != -2) *reinterpret_cast<f4 *>(buf + loff[it]) = xr[it];
    };

    // the lane's filter values of a chunk: g[t][tap] for its output channel and the input channels cbase + 4 kq + t
    const int row = m0 + jn;
    const bool rok = row < a.M;
    auto load_g = [&](int cbase, float (&A)[36]) {
        if (MODE == MODE_DGRAD) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *w = a.W + ((long long)(cbase + 4 * kq + t) * a.Cin + (rok ? row : 0)) * 9;
#pragma unroll
                for (int i = 0; i < 9; ++i) A[9 * t + i] = w[i];
            }
        } else {
            const float *w = a.W + (long long)(rok ? row : 0) * a.K + (long long)(cbase + 4 * kq) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const f4 v = *reinterpret_cast<const f4 *>(w + 4 * i);
                A[4 * i] = v.x; A[4 * i + 1] = v.y; A[4 * i + 2] = v.z; A[4 * i + 3] = v.w;
            }
        }
        if (!rok) {
#pragma unroll
            for (int i = 0; i < 36; ++i) A[i] = 0.0f;
        }
    };

    // input transform: this thread takes row i = wave of B^T d B for (tile tt, channel group cg)
    const int tt = lane & 15, cg = lane >> 4;
    const int ntile = a.TH * a.TW;
    const int tu = tt < ntile ? tt : 0;  // (unused slots repeat tile 0: any finite values will do, their outputs are not stored)
    const int tyy = tu / a.TW, txx = tu - tyy * a.TW;
    // rows of d that row `wave` of B^T d combines: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sb = wave == 1 ? 1.0f : -1.0f;
    const int pa = ((2 * tyy + ra) * a.pitch + 2 * txx) * WN_CS + 4 * cg, pb = ((2 * tyy + rb) * a.pitch + 2 * txx) * WN_CS + 4 * cg;
    float *vdst = Vs + ((4 * wave) * 16 + tt) * WN_CS + 4 * cg;
    auto transform = [&](const float *buf) {
        f4 e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4 da = *reinterpret_cast<const f4 *>(buf + pa + c * WN_CS), db = *reinterpret_cast<const f4 *>(buf + pb + c * WN_CS);
            e[c] = da + sb * db;
        }
        *reinterpret_cast<f4 *>(vdst) = e[0] - e[2];
        *reinterpret_cast<f4 *>(vdst + 16 * WN_CS) = e[1] + e[2];
        *reinterpret_cast<f4 *>(vdst + 32 * WN_CS) = e[2] - e[1];
        *reinterpret_cast<f4 *>(vdst + 48 * WN_CS) = e[1] - e[3];
    };

    floatx4 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = a.KC >> 4;
    float A[36], An[36];
    f4 xr[WN_MAXIT];
    stage_load(0, xr);
    load_g(0, A);
    stage_store(smem, xr);
    __syncthreads();
    const float *vsrc = Vs + jn * WN_CS + 4 * kq;
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) {
            stage_load(16 * (c + 1), xr);
            load_g(16 * (c + 1), An);
        }
        transform(smem + (c & 1) * bufsz);
        __syncthreads();
        // U = G g G^T per input channel t, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; the input gradient runs the flipped filter
        float U[16][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float g[3][3], Gg[4][3];
#pragma unroll
            for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = A[9 * t + (MODE == MODE_DGRAD ? 8 - i : i)];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float s = 0.5f * (g[0][b] + g[2][b]), h = 0.5f * g[1][b];
                Gg[0][b] = g[0][b]; Gg[1][b] = s + h; Gg[2][b] = s - h; Gg[3][b] = g[2][b];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s = 0.5f * (Gg[i][0] + Gg[i][2]), h = 0.5f * Gg[i][1];
                U[4 * i][t] = Gg[i][0]; U[4 * i + 1][t] = s + h; U[4 * i + 2][t] = s - h; U[4 * i + 3][t] = Gg[i][2];
            }
        }
        // four components at a time: their B fragments, then 16 MFMAs in an order that spaces the uses of an accumulator
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f4 *>(vsrc + (4 * i + j) * 16 * WN_CS);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[4 * i + j][t], b[j][t], acc[4 * i + j], 0, 0, 0);
        }
        if (more) stage_store(smem + ((c + 1) & 1) * bufsz, xr);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 36; ++i) A[i] = An[i];
    }

    // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]: accumulator r of component c belongs to channel m0 + 4 kq + r, tile jn
    if (jn >= ntile) return;
    const int oty = ty0 + jn / a.TW, otx = tx0 + (jn - (jn / a.TW) * a.TW);
    const int oy = 2 * oty, ox = 2 * otx;
    if (oy >= a.OH || ox >= a.OW) return;
    const bool two_x = ox + 1 < a.OW, two_y = oy + 1 < a.OH, pair = two_x && !(a.OW & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = m0 + 4 * kq + r;
        if (ch >= a.M) continue;
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
            t1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
        float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
        const long long o = (((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox;
        if (a.res) {
            y00 += a.res[o];
            if (two_x) y01 += a.res[o + 1];
            if (two_y) {
                y10 += a.res[o + a.OW];
                if (two_x) y11 += a.res[o + a.OW + 1];
            }
        }
        if (pair) {
            *reinterpret_cast<f2 *>(a.out + o) = f2{y00, y01};
            if (two_y) *reinterpret_cast<f2 *>(a.out + o + a.OW) = f2{y10, y11};
        } else {
            a.out[o] = y00;
            if (two_x) a.out[o + 1] = y01;
            if (two_y) {
                a.out[o + a.OW] = y10;
                if (two_x) a.out[o + a.OW + 1] = y11;
            }
        }
    }
}

template <int MODE>
static int launch_wino(ConvArgs a, int N, hipStream_t st, const char *who, bool *taken)
{
    *taken = false;
    const int stride = MODE == MODE_DGRAD ? a.dil : a.rs;
    if (a.ks != 3 || a.padT != 1 || stride != 1 || a.dil != 1 || a.KC % 16 != 0 || a.M % 16 != 0 || a.M < 64 || a.KC < 64) return 0;
    if (a.idx || a.bn_on || a.stat_part || a.bwd_part || ((uintptr_t)a.W & 15) || ((uintptr_t)a.out & 7)) return 0;
    if ((long long)a.IH * a.IW * 16 >= (1LL << 31) || a.OH != a.IH || a.OW != a.IW) return 0;
    const int TY = (a.OH + 1) / 2, TX = (a.OW + 1) / 2;
    // block of up to 16 tiles: the shape that wastes the fewest tile slots, the squarer one on a tie (less halo)
    double best = 0.0;
    for (int tbw = 1; tbw <= 16; ++tbw) {
        const int tbh = 16 / tbw;
        const long long blocks = (long long)((TY + tbh - 1) / tbh) * ((TX + tbw - 1) / tbw);
        const int plane = (2 * tbh + 2) * (2 * tbw + 2);
        if (4 * plane > 256 * WN_MAXIT) continue;
        const double util = (double)TY * TX / ((double)blocks * 16) - 1e-4 * plane;
        if (util > best) {
            best = util;
            a.TW = tbw;
            a.TH = tbh;
        }
    }
    if (best <= 0.0) return 0;
    a.tiles_x = (TX + a.TW - 1) / a.TW;
    a.tiles_y = (TY + a.TH - 1) / a.TH;
    a.pitch = 2 * a.TW + 2;
    a.rows_in = 2 * a.TH + 2;
    a.plane = a.rows_in * a.pitch;
    const long long gx = (long long)N * a.tiles_x * a.tiles_y;
    const int mgroups = (a.M + 63) / 64;
    if (gx >= (1LL << 31) || mgroups >= 65536) return 0;
    const size_t lds = ((size_t)2 * a.plane * WN_CS + (size_t)16 * 16 * WN_CS) * 4;
    hipLaunchKernelGGL((conv3x3_wino_kernel<MODE>), dim3((unsigned)gx, (unsigned)mgroups), dim3(256), lds, st, a);
    *taken = true;
    return check_launch(who);
}

}  // namespace aq

int aq_wino_launch_fwd(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken)
{
    return aq::launch_wino<aq::MODE_FWD>(a, N, st, who, taken);
}
int aq_wino_launch_dgrad(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken)
{
    return aq::launch_wino<aq::MODE_DGRAD>(a, N, st, who, taken);
}
