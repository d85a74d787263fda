// 3x3 / pad 1 / stride 1|2 convolution of the quantized layers as implicit GEMMs on the fp32
// matrix cores of gfx950 (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate), NCHW.
//
//   forward : y[n,co,oy,ox]  = sum_{ci,t} W[co,ci,t] * x[n,ci,oy*s+ty-1,ox*s+tx-1]   (adp_qtz_fan.py:574-601)
//   dgrad   : dx[n,ci,iy,ix] = sum_{co,t} W[co,ci,t] * dy[n,co,(iy+1-ty)/s,(ix+1-tx)/s]
//   wgrad   : dW[co,ci,t]    = sum_{n,oy,ox} dy[n,co,oy,ox] * x[n,ci,oy*s+ty-1,ox*s+tx-1]
//
// forward and dgrad share one kernel: GEMM  Out[M, pixels] = A[M, K] * B[K, pixels]  with
// M = output channels, K = 9 * input channels, B gathered on the fly from an LDS image of the
// input tile (with its halo, zero padded; for the stride-2 dgrad the image is the zero-dilated
// dy).  A workgroup = 4 wavefronts owns TH x TW output pixels of one image and 16*MT output
// channels; each wavefront owns NT tiles of 16 pixels.  A fragments (weights) are read straight
// from global memory / L2 into registers, four consecutive k per lane (one 16-byte load), which
// fixes the order in which a lane meets k: step t of k-block kb uses k = 16*kb + 4*(lane>>4) + t
// for both operands.  The LDS offset of every k (channel plane + tap) comes from a table built
// once per workgroup.
//
// wgrad: GEMM  dW[Cout, 9*Cin] = dY[Cout, pixels] * Xg[pixels, 9*Cin].  A workgroup owns a group of
// MG x NG output tiles (16x16 each, all kept in registers by every wavefront) and a set of pixel
// chunks (TH x TW pixels of IPB images); its 4 wavefronts split the pixels of each chunk, are
// summed through LDS in a fixed order and the workgroup writes one partial slab; a second kernel
// adds the slabs of all workgroups in ascending order.  No atomics anywhere: bit-reproducible.
#include "aq_common.h"
#include "../../include/aq_hip.h"

#include <algorithm>

namespace aq {

typedef float floatx4 __attribute__((ext_vector_type(4)));

enum ConvMode { MODE_FWD = 0, MODE_FWD_CODES = 1, MODE_DGRAD = 2 };

struct ConvArgs {
    const float *in;     // x (forward) or dy (dgrad), NCHW
    const float *W;      // [Cout, Cin, 3, 3]
    const uint8_t *idx;  // codes, same shape as W (MODE_FWD_CODES)
    const float *q;      // levels (MODE_FWD_CODES)
    float *out;
    int L;
    int N, Cin, Cout;    // conv definition
    int M, KC, K;        // GEMM: output channels, input channels, 9*KC
    int OH, OW;          // output spatial size of this GEMM
    int IH, IW;          // spatial size of the (undilated) input tensor
    int rs;              // step between neighbouring output pixels inside the LDS image
    int dil;             // 2 when the LDS image is the zero-dilated input (stride-2 dgrad), else 1
    int TW, TH, tiles_x, tiles_y;
    int pitch, rows_in, plane;
    int Kpad;
    int vecA;            // rows of A are 16-byte aligned and K % 4 == 0
};

template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *ktab = reinterpret_cast<int *>(smem);            // [Kpad] LDS offset of k
    int *atab = ktab + a.Kpad;                            // [Kpad] weight offset of k (dgrad only)
    float *qs = reinterpret_cast<float *>(atab + (MODE == MODE_DGRAD ? a.Kpad : 0));  // [L] levels (codes only)
    float *xs = qs + (MODE == MODE_FWD_CODES ? ((a.L + 3) & ~3) : 0);                 // [KC][rows_in][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int m0 = blockIdx.y * 16 * MT;

    // ---- tables -------------------------------------------------------------------------------
    for (int k = tid; k < a.Kpad; k += 256) {
        int off = 0, aoff = 0;
        if (k < a.K) {
            const int c = k / 9, tap = k - 9 * c;
            off = c * a.plane + (tap / 3) * a.pitch + (tap % 3);
            aoff = c * a.Cin * 9 + (8 - tap);  // W[c(=co), ., 2-ty, 2-tx]
        }
        ktab[k] = off;
        if (MODE == MODE_DGRAD) atab[k] = aoff;
    }
    if (MODE == MODE_FWD_CODES)
        for (int i = tid; i < a.L; i += 256) qs[i] = a.q[i];

    // ---- LDS image of the input tile (halo included, zero outside the tensor) -------------------
    {
        const int r0 = oy0 * a.rs - 1, c0 = ox0 * a.rs - 1;  // image origin in (dilated) input coordinates
        const float *src = a.in + (long long)n * a.KC * a.IH * a.IW;
        const int total = a.KC * a.plane;
        for (int e = tid; e < total; e += 256) {
            const int c = e / a.plane, rem = e - c * a.plane;
            const int r = rem / a.pitch, cc = rem - r * a.pitch;
            int gy = r0 + r, gx = c0 + cc;
            bool ok = gy >= 0 && gx >= 0;
            if (a.dil == 2) {
                ok = ok && !((gy | gx) & 1);
                gy >>= 1;
                gx >>= 1;
            }
            ok = ok && gy < a.IH && gx < a.IW;
            xs[e] = ok ? src[((long long)c * a.IH + gy) * a.IW + gx] : 0.0f;
        }
    }
    __syncthreads();

    // ---- per-lane pixel bases -------------------------------------------------------------------
    int pixbase[NT];
    int oyx[NT];  // (oy << 16) | ox, or -1 when outside the output
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pr = (wave * NT + t) * 16 + jn;
        const int py = pr / a.TW, px = pr - py * a.TW;
        pixbase[t] = py * a.rs * a.pitch + px * a.rs;
        const int oy = oy0 + py, ox = ox0 + px;
        oyx[t] = (oy < a.OH && ox < a.OW) ? ((oy << 16) | ox) : -1;
    }

    floatx4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nkb = a.Kpad >> 4;
    for (int kb = 0; kb < nkb; ++kb) {
        const int kbase = 16 * kb + 4 * kq;
        const int4 ko = *reinterpret_cast<const int4 *>(&ktab[kbase]);
        const int kof[4] = {ko.x, ko.y, ko.z, ko.w};
        float av[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m0 + 16 * m + jn;
            const bool rok = row < a.M;
            if (MODE == MODE_DGRAD) {
                const int4 ao = *reinterpret_cast<const int4 *>(&atab[kbase]);
                const int aof[4] = {ao.x, ao.y, ao.z, ao.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) av[m][t] = (rok && kbase + t < a.K) ? a.W[aof[t] + row * 9] : 0.0f;
            } else if (MODE == MODE_FWD_CODES) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    av[m][t] = (rok && kbase + t < a.K) ? qs[a.idx[(long long)row * a.K + kbase + t]] : 0.0f;
            } else {
                if (a.vecA) {
                    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rok && kbase < a.K) w4 = *reinterpret_cast<const float4 *>(&a.W[(long long)row * a.K + kbase]);
                    av[m][0] = w4.x; av[m][1] = w4.y; av[m][2] = w4.z; av[m][3] = w4.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[m][t] = (rok && kbase + t < a.K) ? a.W[(long long)row * a.K + kbase + t] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = xs[pixbase[nt] + kof[t]];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][t], bv[nt], acc[m][nt], 0, 0, 0);
        }
    }

    // ---- epilogue: D[row = 4*(lane>>4)+r][col = lane&15] -----------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (oyx[nt] < 0) continue;
            const int oy = oyx[nt] >> 16, ox = oyx[nt] & 0xffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = m0 + 16 * m + 4 * kq + r;
                if (ch < a.M) a.out[(((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox] = acc[m][nt][r];
            }
        }
}

// ------------------------------------------------------------------------------------------ wgrad

struct WgradArgs {
    const float *x, *dy;
    float *part;  // [chunks_groups][Cout*Cin*9]
    int N, Cin, H, Wd, Cout, stride, Ho, Wo;
    int K9;                      // Cin*9
    int TW, TH, tiles_x, tiles_y;
    int pitch, rows_in, plane;   // LDS image of x
    int cpitch;                  // LDS pitch of a dy channel (TH*TW + 2)
    int IPB;                     // images per workgroup
    int ngroups_img;             // ceil(N / IPB)
};

template <int MG, int NG>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npix = a.TH * a.TW;
    float *dys = smem;                           // [16*MG][cpitch]
    float *xs = smem + 16 * MG * a.cpitch;       // [Cin][rows_in][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int ig = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int m0 = blockIdx.y * 16 * MG;   // first output channel of this group
    const int j0 = blockIdx.z * 16 * NG;   // first (ci, tap) column of this group

    // column (ci, tap) owned by this lane in each n-tile -> offset inside the x image
    int joff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int jcol = j0 + 16 * g + jn;
        int off = 0;
        if (jcol < a.K9) {
            const int ci = jcol / 9, tap = jcol - 9 * ci;
            off = ci * a.plane + (tap / 3) * a.pitch + (tap % 3);
        }
        joff[g] = off;
    }
    // pixel 4t+kq of a 16-pixel block (raster order inside the TH x TW tile)
    int toff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int pp = 4 * t + kq;
        const int py = pp / a.TW, px = pp - py * a.TW;
        toff[t] = py * a.stride * a.pitch + px * a.stride;
    }

    floatx4 acc[MG][NG];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[m][g] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int r0 = oy0 * a.stride - 1, c0 = ox0 * a.stride - 1;
    const int nblk16 = npix >> 4;
    for (int im = 0; im < a.IPB; ++im) {
        const int n = ig * a.IPB + im;
        if (n >= a.N) break;
        __syncthreads();  // previous chunk fully consumed
        // dy tile: [16*MG channels][npix], zero outside the output / beyond Cout
        {
            const float *src = a.dy + (long long)n * a.Cout * a.Ho * a.Wo;
            const int total = 16 * MG * npix;
            for (int e = tid; e < total; e += 256) {
                const int c = e / npix, pr = e - c * npix;
                const int py = pr / a.TW, px = pr - py * a.TW;
                const int oy = oy0 + py, ox = ox0 + px, co = m0 + c;
                const bool ok = co < a.Cout && oy < a.Ho && ox < a.Wo;
                dys[c * a.cpitch + pr] = ok ? src[((long long)co * a.Ho + oy) * a.Wo + ox] : 0.0f;
            }
        }
        // x image with halo
        {
            const float *src = a.x + (long long)n * a.Cin * a.H * a.Wd;
            const int total = a.Cin * a.plane;
            for (int e = tid; e < total; e += 256) {
                const int c = e / a.plane, rem = e - c * a.plane;
                const int r = rem / a.pitch, cc = rem - r * a.pitch;
                const int gy = r0 + r, gx = c0 + cc;
                const bool ok = gy >= 0 && gx >= 0 && gy < a.H && gx < a.Wd;
                xs[e] = ok ? src[((long long)c * a.H + gy) * a.Wd + gx] : 0.0f;
            }
        }
        __syncthreads();
        for (int pb = wave; pb < nblk16; pb += 4) {
            const int p16 = pb * 16;
            const int py = p16 / a.TW, px = p16 - py * a.TW;
            const int bo = py * a.stride * a.pitch + px * a.stride;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float av[MG], bv[NG];
#pragma unroll
                for (int m = 0; m < MG; ++m) av[m] = dys[(16 * m + jn) * a.cpitch + p16 + 4 * t + kq];
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] = xs[bo + toff[t] + joff[g]];
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[g], acc[m][g], 0, 0, 0);
            }
        }
    }

    // ---- sum the 4 wavefronts in LDS (fixed order 0+1+2+3), write the workgroup's slab ------------
    float *part = a.part + (long long)blockIdx.x * a.Cout * a.K9;
    constexpr int TPR = 8;  // tiles per round: 8 tiles * 4 waves * 256 floats = 32 KiB
    float *red = smem;
    for (int base = 0; base < MG * NG; base += TPR) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int tile = m * NG + g;
                if (tile >= base && tile < base + TPR) {
                    float *dst = red + (((tile - base) * 4 + wave) * 64 + lane) * 4;
                    *reinterpret_cast<float4 *>(dst) = make_float4(acc[m][g][0], acc[m][g][1], acc[m][g][2], acc[m][g][3]);
                }
            }
        __syncthreads();
        // 256 threads: thread = (tile-in-round tr = tid>>5? ) -> each thread sums one float4 of one tile per step
        for (int e = tid; e < TPR * 64; e += 256) {
            const int tr = e >> 6, ln = e & 63;
            const int tile = base + tr;
            if (tile >= MG * NG) break;
            const float4 s0 = *reinterpret_cast<const float4 *>(red + ((tr * 4 + 0) * 64 + ln) * 4);
            const float4 s1 = *reinterpret_cast<const float4 *>(red + ((tr * 4 + 1) * 64 + ln) * 4);
            const float4 s2 = *reinterpret_cast<const float4 *>(red + ((tr * 4 + 2) * 64 + ln) * 4);
            const float4 s3 = *reinterpret_cast<const float4 *>(red + ((tr * 4 + 3) * 64 + ln) * 4);
            const float v[4] = {((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y,
                                ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w};
            const int m = tile / NG, g = tile - m * NG;
            const int col = j0 + 16 * g + (ln & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * m + 4 * (ln >> 4) + r;
                if (row < a.Cout && col < a.K9) part[(long long)row * a.K9 + col] = v[r];
            }
        }
    }
}

// dW[i] = sum over the workgroup slabs.  A workgroup owns 32 consecutive elements; 8 thread rows
// each add every 8th slab in ascending order, then the 8 running sums are added in order 0..7.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int n, int nslab)
{
    __shared__ float sh[8][33];
    const int c = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.0f;
    if (i < n)
        for (int b = sg; b < nslab; b += 8) s += part[(long long)b * n + i];
    sh[sg][c] = s;
    __syncthreads();
    if (sg == 0 && i < n) {
        float t = sh[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][c];
        dW[i] = t;
    }
}

// ------------------------------------------------------------------------------------- host side

static int check_conv(const char *who, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad)
{
    AQ_REQUIRE(ks == 3 && pad == 1, -ENOTSUP, "%s: only 3x3 kernels with padding 1 are implemented (got k=%d pad=%d)", who, ks, pad);
    AQ_REQUIRE(stride == 1 || stride == 2, -ENOTSUP, "%s: stride must be 1 or 2 (got %d)", who, stride);
    AQ_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && Wd >= 1, -EINVAL, "%s: bad sizes", who);
    AQ_REQUIRE(H < 32768 && Wd < 32768 && (long long)Cin * 9 < (1 << 20) && (long long)Cout * 9 < (1 << 20), -E2BIG, "%s: dimension too large", who);
    return 0;
}

static void pick_tile(int OW, int npix, int &TW, int &TH)
{
    TW = OW > 16 ? 32 : (OW > 8 ? 16 : 8);
    TH = npix / TW;
}

template <int MODE>
static int launch_gemm(ConvArgs &a, int N, hipStream_t st, const char *who)
{
    // tile choice: NT = 2 (128 pixels) and MT up to 4, shrunk until the grid fills the chip
    const int mtiles = (a.M + 15) / 16;
    int MT = std::min(4, mtiles), NT = 2;
    if (MT == 3) MT = 2;
    auto blocks = [&](int mt, int nt) {
        int TW, TH;
        pick_tile(a.OW, 64 * nt, TW, TH);
        return (long long)N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW) * ((mtiles + mt - 1) / mt);
    };
    while (blocks(MT, NT) < 512 && (NT > 1 || MT > 1)) {
        if (NT > 1) NT = 1;
        else MT >>= 1;
    }
    pick_tile(a.OW, 64 * NT, a.TW, a.TH);
    a.tiles_x = (a.OW + a.TW - 1) / a.TW;
    a.tiles_y = (a.OH + a.TH - 1) / a.TH;
    a.pitch = (a.TW - 1) * a.rs + 3;
    a.rows_in = (a.TH - 1) * a.rs + 3;
    a.plane = a.rows_in * a.pitch;
    a.Kpad = (a.K + 15) & ~15;
    a.vecA = (a.K % 4 == 0) && (((uintptr_t)a.W & 15) == 0);
    size_t lds = (size_t)a.Kpad * 4 * (MODE == MODE_DGRAD ? 2 : 1) + (MODE == MODE_FWD_CODES ? ((a.L + 3) & ~3) * 4 : 0) +
                 (size_t)a.KC * a.plane * 4;
    AQ_REQUIRE(lds <= 64 * 1024, -E2BIG, "%s: input tile needs %zu bytes of LDS (limit 65536): too many channels", who, lds);
    const long long gx = (long long)N * a.tiles_x * a.tiles_y;
    AQ_REQUIRE(gx < (1LL << 31), -E2BIG, "%s: grid too large", who);
    dim3 grid((unsigned)gx, (unsigned)((mtiles + MT - 1) / MT));
#define AQ_GEMM(MM, NN) hipLaunchKernelGGL((conv_gemm_kernel<MM, NN, MODE>), grid, dim3(256), lds, st, a)
    if (MT == 4 && NT == 2) AQ_GEMM(4, 2);
    else if (MT == 4) AQ_GEMM(4, 1);
    else if (MT == 2 && NT == 2) AQ_GEMM(2, 2);
    else if (MT == 2) AQ_GEMM(2, 1);
    else if (NT == 2) AQ_GEMM(1, 2);
    else AQ_GEMM(1, 1);
#undef AQ_GEMM
    return check_launch(who);
}

static int conv_fwd_common(const float *x, const float *W, const uint8_t *idx, const float *q, int L, float *y, int N, int Cin, int H,
                           int Wd, int Cout, int ks, int stride, int pad, hipStream_t st, const char *who)
{
    if (int rc = check_conv(who, N, Cin, H, Wd, Cout, ks, stride, pad)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(x && y && (W || (idx && q)), -EFAULT, "%s: NULL pointer", who);
    ConvArgs a = {};
    a.in = x; a.W = W; a.idx = idx; a.q = q; a.L = L; a.out = y;
    a.N = N; a.Cin = Cin; a.Cout = Cout;
    a.M = Cout; a.KC = Cin; a.K = Cin * 9;
    a.OH = (H + 2 - 3) / stride + 1; a.OW = (Wd + 2 - 3) / stride + 1;
    a.IH = H; a.IW = Wd; a.rs = stride; a.dil = 1;
    return idx ? launch_gemm<MODE_FWD_CODES>(a, N, st, who) : launch_gemm<MODE_FWD>(a, N, st, who);
}

} // namespace aq

extern "C" {

int aq_conv_fwd(const float *x, const float *W, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *stream)
{
    return aq::conv_fwd_common(x, W, nullptr, nullptr, 0, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream), "aq_conv_fwd");
}

int aq_conv_fwd_codes(const float *x, const uint8_t *idx, const float *q, int L, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize,
                      int stride, int pad, void *stream)
{
    AQ_REQUIRE(L >= 1 && L <= AQ_MAX_LEVELS, -EINVAL, "aq_conv_fwd_codes: L=%d outside 1..%d", L, AQ_MAX_LEVELS);
    AQ_REQUIRE(idx && q, -EFAULT, "aq_conv_fwd_codes: NULL codes / levels");
    return aq::conv_fwd_common(x, nullptr, idx, q, L, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream), "aq_conv_fwd_codes");
}

int aq_conv_dgrad(const float *dy, const float *W, float *dx, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_dgrad", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(dy && W && dx, -EFAULT, "aq_conv_dgrad: NULL pointer");
    ConvArgs a = {};
    a.in = dy; a.W = W; a.out = dx;
    a.N = N; a.Cin = Cin; a.Cout = Cout;
    a.M = Cin; a.KC = Cout; a.K = Cout * 9;
    a.OH = H; a.OW = Wd;
    a.IH = (H + 2 - 3) / stride + 1; a.IW = (Wd + 2 - 3) / stride + 1;
    a.rs = 1; a.dil = stride;
    return launch_gemm<MODE_DGRAD>(a, N, static_cast<hipStream_t>(stream), "aq_conv_dgrad");
}

static int wgrad_plan(aq::WgradArgs &a, int &MG, int &NG, dim3 &grid, size_t &lds, size_t &ws_bytes)
{
    using namespace aq;
    a.Ho = (a.H + 2 - 3) / a.stride + 1;
    a.Wo = (a.Wd + 2 - 3) / a.stride + 1;
    a.K9 = a.Cin * 9;
    const int mtiles = (a.Cout + 15) / 16, ntiles = (a.K9 + 15) / 16;
    // tile group kept in registers by every wavefront: MG x NG <= 36 tiles
    if (ntiles <= 3) NG = 3;
    else if (ntiles <= 9) NG = 9;
    else if (ntiles <= 18) NG = 18;
    else NG = 36;
    MG = (NG <= 18 && mtiles >= 2) ? 2 : 1;
    if (NG == 3 && MG == 2) MG = 1;
    pick_tile(a.Wo, 128, a.TW, a.TH);
    if ((long long)a.Ho * a.Wo <= 64) pick_tile(a.Wo, 64, a.TW, a.TH);
    a.tiles_x = (a.Wo + a.TW - 1) / a.TW;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    a.pitch = (a.TW - 1) * a.stride + 3;
    a.rows_in = (a.TH - 1) * a.stride + 3;
    a.plane = a.rows_in * a.pitch;
    a.cpitch = a.TH * a.TW + 2;
    const int gy = (mtiles + MG - 1) / MG, gz = (ntiles + NG - 1) / NG;
    // images per workgroup: keep >= ~256 workgroups in flight, at most 8 images each
    long long chunks = (long long)a.N * a.tiles_x * a.tiles_y;
    int ipb = 1;
    while (ipb < 8 && chunks * gy * gz / (ipb * 2) >= 256 && ipb * 2 <= a.N) ipb *= 2;
    a.IPB = ipb;
    a.ngroups_img = (a.N + ipb - 1) / ipb;
    const long long gx = (long long)a.ngroups_img * a.tiles_x * a.tiles_y;
    AQ_REQUIRE(gx < (1LL << 31) && gy < 65536 && gz < 65536, -E2BIG, "aq_conv_wgrad: grid too large");
    grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)gz);
    lds = ((size_t)16 * MG * a.cpitch + (size_t)a.Cin * a.plane) * 4;
    lds = std::max(lds, (size_t)8 * 4 * 256 * 4);
    AQ_REQUIRE(lds <= 64 * 1024, -E2BIG, "aq_conv_wgrad: tiles need %zu bytes of LDS (limit 65536): too many channels", lds);
    ws_bytes = (size_t)gx * a.Cout * a.K9 * 4;
    return 0;
}

size_t aq_conv_wgrad_workspace_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    if (aq::check_conv("aq_conv_wgrad_workspace_bytes", N, Cin, H, Wd, Cout, ksize, stride, pad) || N == 0) return 0;
    aq::WgradArgs a = {};
    a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Cout = Cout; a.stride = stride;
    int MG, NG;
    dim3 grid;
    size_t lds, ws;
    if (wgrad_plan(a, MG, NG, grid, lds, ws)) return 0;
    return ws;
}

int aq_conv_wgrad(const float *x, const float *dy, float *dW, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_wgrad", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(dW, -EFAULT, "aq_conv_wgrad: NULL dW");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nW = Cout * Cin * 9;
    if (N == 0) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nW + 31) / 32), dim3(256), 0, st, nullptr, dW, nW, 0);
        return check_launch("aq_conv_wgrad");
    }
    AQ_REQUIRE(x && dy, -EFAULT, "aq_conv_wgrad: NULL pointer");
    WgradArgs a = {};
    a.x = x; a.dy = dy; a.part = static_cast<float *>(workspace);
    a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Cout = Cout; a.stride = stride;
    int MG, NG;
    dim3 grid;
    size_t lds, ws;
    if (int rc = wgrad_plan(a, MG, NG, grid, lds, ws)) return rc;
    AQ_REQUIRE(workspace && workspace_bytes >= ws, -ENOSPC, "aq_conv_wgrad: workspace too small: need %zu bytes, got %zu", ws, workspace_bytes);
#define AQ_WG(MM, NN) hipLaunchKernelGGL((conv_wgrad_kernel<MM, NN>), grid, dim3(256), lds, st, a)
    if (NG == 3) AQ_WG(1, 3);
    else if (NG == 9 && MG == 1) AQ_WG(1, 9);
    else if (NG == 9) AQ_WG(2, 9);
    else if (NG == 18 && MG == 1) AQ_WG(1, 18);
    else if (NG == 18) AQ_WG(2, 18);
    else AQ_WG(1, 36);
#undef AQ_WG
    if (int rc = check_launch("aq_conv_wgrad")) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nW + 31) / 32), dim3(256), 0, st, a.part, dW, nW, (int)grid.x);
    return check_launch("aq_conv_wgrad(reduce)");
}

} // extern "C"
