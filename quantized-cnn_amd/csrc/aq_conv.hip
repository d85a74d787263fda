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
#include "aq_conv_shared.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

namespace aq {


// Fill the LDS image xs[nch][plane] of one input tile: plane = rows_in x pitch elements starting at
// (r0, c0) in (dilated) input coordinates, zero outside the tensor.  Lanes own plane elements
// (lane, lane+64, ..), wavefronts own channels (wave, wave+4, ..).  The geometry of a lane's
// elements (row/column, bounds, global offset) is worked out once and reused for every channel, and
// the loads of four channels are all issued before the first LDS store, so one wavefront keeps up to
// 4*ceil(plane/64) loads in flight (the fill is latency-bound, not bandwidth-bound).
constexpr int STAGE_MAXJ = 10;  // plane <= 640 elements

__device__ __forceinline__ void stage_image(float *__restrict__ xs, const float *__restrict__ src, int nch, int IH, int IW, int plane,
                                            int pitch, int r0, int c0, int dil, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int chs = IH * IW;
    if (plane <= 64 * STAGE_MAXJ) {
        const int J = (plane + 63) >> 6;
        int goff[STAGE_MAXJ];
#pragma unroll
        for (int j = 0; j < STAGE_MAXJ; ++j) {
            const int e = lane + 64 * j;
            int off = -2;  // -2: beyond the plane, -1: zero padding
            if (e < plane) {
                const int r = e / pitch, cc = e - r * pitch;
                int gy = r0 + r, gx = c0 + cc;
                bool ok = gy >= 0 && gx >= 0;
                if (dil == 2) {
                    ok = ok && !((gy | gx) & 1);
                    gy >>= 1;
                    gx >>= 1;
                }
                ok = ok && gy < IH && gx < IW;
                off = ok ? gy * IW + gx : -1;
            }
            goff[j] = off;
        }
        for (int cb = wave; cb < nch; cb += 16) {
            float v[4][STAGE_MAXJ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + 4 * u;
#pragma unroll
                for (int j = 0; j < STAGE_MAXJ; ++j)
                    if (j < J) v[u][j] = (c < nch && goff[j] >= 0) ? src[(long long)c * chs + goff[j]] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + 4 * u;
#pragma unroll
                for (int j = 0; j < STAGE_MAXJ; ++j)
                    if (j < J && c < nch && goff[j] != -2) xs[c * plane + lane + 64 * j] = v[u][j];
            }
        }
    } else {  // large planes (not produced by the tile shapes in use): generic element loop
        const int total = nch * plane;
        for (int e = tid; e < total; e += 256) {
            const int c = e / plane, rem = e - c * plane;
            const int r = rem / pitch, cc = rem - r * pitch;
            int gy = r0 + r, gx = c0 + cc;
            bool ok = gy >= 0 && gx >= 0;
            if (dil == 2) {
                ok = ok && !((gy | gx) & 1);
                gy >>= 1;
                gx >>= 1;
            }
            ok = ok && gy < IH && gx < IW;
            xs[e] = ok ? src[((long long)c * IH + gy) * IW + gx] : 0.0f;
        }
    }
}

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
    stage_image(xs, a.in + (long long)n * a.KC * a.IH * a.IW, a.KC, a.IH, a.IW, a.plane, a.pitch, oy0 * a.rs - 1, ox0 * a.rs - 1, a.dil, tid);
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

    // A fragments of k-block kb: 4 consecutive k per lane and m-tile
    auto load_a = [&](int kb, float (&av)[MT][4]) {
        const int kbase = 16 * kb + 4 * kq;
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
    };

    // Software pipeline over k-blocks (a wavefront often runs alone on its SIMD, so it hides its own
    // latencies): while the MFMAs of block kb run, the B fragments and weights of kb+1 and the offset
    // table entry of kb+2 are already in flight.
    const int nkb = a.Kpad >> 4;
    auto load_k = [&](int kb, int (&kof)[4]) {
        const int4 ko = *reinterpret_cast<const int4 *>(&ktab[16 * (kb < nkb ? kb : nkb - 1) + 4 * kq]);
        kof[0] = ko.x; kof[1] = ko.y; kof[2] = ko.z; kof[3] = ko.w;
    };
    auto load_b = [&](const int (&kof)[4], float (&bv)[4][NT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[t][nt] = xs[pixbase[nt] + kof[t]];
    };
    float av[MT][4], an[MT][4], bv[4][NT], bn[4][NT];
    int kof1[4], kof2[4];
    load_a(0, av);
    load_k(0, kof1);
    load_b(kof1, bv);
    load_k(1, kof1);
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + 1 < nkb) {
            load_a(kb + 1, an);
            load_b(kof1, bn);
        }
        load_k(kb + 2, kof2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][t], bv[t][nt], acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kof1[t] = kof2[t];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m][t] = an[m][t];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[t][nt] = bn[t][nt];
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
                if (ch < a.M) {
                    const long long o = (((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox;
                    a.out[o] = a.res ? acc[m][nt][r] + a.res[o] : acc[m][nt][r];
                }
            }
        }
}

// General form of the forward / dgrad GEMM: any odd kernel size up to 7, any padding below it, stride 1 | 2, any
// number of channels.  Same tile geometry and fragment scheme as conv_gemm_kernel, but the input channels pass
// through LDS in chunks of CK (a multiple of 16 unless the tensor has fewer), the accumulators staying in
// registers across chunks; k inside a chunk is (channel, tap) like everywhere else, so a chunk's k range is a
// contiguous run of the weight matrix' k axis.
// WM = 1: the four wavefronts split the pixels of the tile (16*MT channels x 64*NT pixels per workgroup); WM = 4: they
// split the output channels (64*MT channels x 16*NT pixels), every wavefront reading all pixel tiles from LDS - the
// weight fragments, which come straight from memory, are then loaded once per workgroup instead of four times,
// which is what layers with hundreds of channels need.
// Stride-2 dgrad (a.par): by output parity class, blockIdx.z = 2*cy + cx.  dx[2a+cy][2b+cx] only receives the taps
// (ty, tx) with ty = cy + pad, tx = cx + pad (mod 2), read from the UNDILATED dy at row a + (cy + pad - ty) / 2: a quarter
// of the matrix work of the zero-dilated formulation, no zeros staged.
struct TapClass {  // taps of one spatial dimension that reach outputs of parity c
    int t0, nt, dmin;  // first tap, number of taps (t0, t0+2, ..), smallest input offset (that of the last tap)
};
__device__ __forceinline__ TapClass tap_class(int c, int pad, int ks)
{
    TapClass r;
    r.t0 = (c + pad) & 1;
    r.nt = r.t0 < ks ? (ks - r.t0 + 1) >> 1 : 0;
    const int tl = r.t0 + 2 * (r.nt - 1);
    r.dmin = r.nt > 0 ? (c + pad - tl) / 2 : 0;  // (exact: the numerator is even)
    return r;
}

template <int MT, int NT, int MODE, int WM>
__global__ __launch_bounds__(256) void conv_gemm_chunked_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *ktab = reinterpret_cast<int *>(smem);            // [Kc] LDS offset of k inside a chunk
    int *atab = ktab + a.Kc;                              // [Kc] weight offset of k inside a chunk (dgrad only)
    float *qs = reinterpret_cast<float *>(atab + (MODE == MODE_DGRAD ? a.Kc : 0));   // [L] levels (codes only)
    float *xs = qs + (MODE == MODE_FWD_CODES ? ((a.L + 3) & ~3) : 0);                // [CK][rows_in][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;  // tile origin in the output grid (the class grid with a.par)
    const int m0 = blockIdx.y * 16 * MT * WM + (WM == 4 ? wave * 16 * MT : 0);
    const int ks = a.ks, ks2 = a.ks2;

    // parity classes (stride-2 dgrad): taps of this class per dimension; otherwise all taps
    const bool par = MODE == MODE_DGRAD && a.par != 0;
    const int cy = par ? (int)blockIdx.z >> 1 : 0, cx = par ? (int)blockIdx.z & 1 : 0;
    const int padF = ks - 1 - a.padT;  // the convolution's own padding (a.padT is that of the flipped, full correlation)
    TapClass cty = {0, ks, 0}, ctx = {0, ks, 0};
    if (par) {
        cty = tap_class(cy, padF, ks);
        ctx = tap_class(cx, padF, ks);
    }
    const int ktaps = cty.nt * ctx.nt;  // taps per channel that this workgroup walks (ks2 without classes)

    for (int k = tid; k < a.Kc; k += 256) {
        int off = 0, aoff = 0;
        if (ktaps > 0 && k < a.CK * ktaps) {
            const int c = k / ktaps, ti = k - ktaps * c;
            if (par) {
                const int iy = ti / ctx.nt, ix = ti - ctx.nt * iy;
                const int ty_ = cty.t0 + 2 * iy, tx_ = ctx.t0 + 2 * ix;
                off = c * a.plane + ((cy + padF - ty_) / 2 - cty.dmin) * a.pitch + ((cx + padF - tx_) / 2 - ctx.dmin);
                aoff = c * a.Cin * ks2 + ty_ * ks + tx_;  // W[c0 + c (= co), ., ty, tx]: the forward tap itself
            } else {
                const int ty_ = ti / ks, tx_ = ti - ks * ty_;
                off = c * a.plane + ty_ * a.pitch + tx_;
                aoff = c * a.Cin * ks2 + (ks2 - 1 - ti);  // W[c0 + c (= co), ., flipped tap]
            }
        }
        ktab[k] = off;
        if (MODE == MODE_DGRAD) atab[k] = aoff;
    }
    if (MODE == MODE_FWD_CODES)
        for (int i = tid; i < a.L; i += 256) qs[i] = a.q[i];

    int pixbase[NT];
    int oyx[NT];  // (oy << 16) | ox in the output tensor, or -1 when outside
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pr = ((WM == 4 ? 0 : wave * NT) + t) * 16 + jn;
        const int py = pr / a.TW, px = pr - py * a.TW;
        pixbase[t] = py * a.rs * a.pitch + px * a.rs;
        int oy = oy0 + py, ox = ox0 + px;
        if (par) {
            oy = 2 * oy + cy;
            ox = 2 * ox + cx;
        }
        oyx[t] = (oy < a.OH && ox < a.OW) ? ((oy << 16) | ox) : -1;
    }

    floatx4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int r0 = par ? oy0 + cty.dmin : oy0 * a.rs - a.padT, c0 = par ? ox0 + ctx.dmin : ox0 * a.rs - a.padT;
    const int dil = par ? 1 : a.dil;
    const long long chs = (long long)a.IH * a.IW;
    const float *img = a.in + (long long)n * a.KC * chs;
    const int nkb = a.Kc >> 4;

    for (int cbase = 0; cbase < a.KC && ktaps > 0; cbase += a.CK) {
        const int nvalid = min(a.CK, a.KC - cbase);
        const int kvalid = nvalid * ktaps;     // k of this chunk that carry weights
        const int kg0 = cbase * ks2;           // first global k of the chunk (forward: k = (channel, tap))
        if (cbase) __syncthreads();            // every wavefront is done reading the previous chunk's image
        // LDS image of the chunk: rows_in x pitch per channel from (r0, c0) in (dilated) input coordinates, zero
        // outside the tensor and for channels beyond the tensor's
        for (int e = tid; e < a.plane; e += 256) {
            const int r = e / a.pitch, cc = e - r * a.pitch;
            int gy = r0 + r, gx = c0 + cc;
            bool ok = gy >= 0 && gx >= 0;
            if (dil == 2) {
                ok = ok && !((gy | gx) & 1);
                gy >>= 1;
                gx >>= 1;
            }
            ok = ok && gy < a.IH && gx < a.IW;
            const float *src = img + (long long)cbase * chs + (ok ? gy * a.IW + gx : 0);
            for (int c = 0; c < a.CK; c += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (ok && c + u < nvalid) ? src[(long long)(c + u) * chs] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + u < a.CK) xs[(c + u) * a.plane + e] = v[u];
            }
        }
        __syncthreads();

        auto load_a = [&](int kb, float (&av)[MT][4]) {
            const int kbase = 16 * kb + 4 * kq;  // inside the chunk
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m0 + 16 * m + jn;
                const bool rok = row < a.M;
                if (MODE == MODE_DGRAD) {
                    const int4 ao = *reinterpret_cast<const int4 *>(&atab[kbase]);
                    const int aof[4] = {ao.x, ao.y, ao.z, ao.w};
                    const float *wb = a.W + (long long)cbase * a.Cin * ks2 + (long long)row * ks2;
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[m][t] = (rok && kbase + t < kvalid) ? wb[aof[t]] : 0.0f;
                } else if (MODE == MODE_FWD_CODES) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        av[m][t] = (rok && kbase + t < kvalid) ? qs[a.idx[(long long)row * a.K + kg0 + kbase + t]] : 0.0f;
                } else {
                    if (a.vecA && kbase + 3 < kvalid) {
                        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (rok) w4 = *reinterpret_cast<const float4 *>(&a.W[(long long)row * a.K + kg0 + kbase]);
                        av[m][0] = w4.x; av[m][1] = w4.y; av[m][2] = w4.z; av[m][3] = w4.w;
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) av[m][t] = (rok && kbase + t < kvalid) ? a.W[(long long)row * a.K + kg0 + kbase + t] : 0.0f;
                    }
                }
            }
        };
        auto load_k = [&](int kb, int (&kof)[4]) {
            const int4 ko = *reinterpret_cast<const int4 *>(&ktab[16 * (kb < nkb ? kb : nkb - 1) + 4 * kq]);
            kof[0] = ko.x; kof[1] = ko.y; kof[2] = ko.z; kof[3] = ko.w;
        };
        auto load_b = [&](const int (&kof)[4], float (&bv)[4][NT]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[t][nt] = xs[pixbase[nt] + kof[t]];
        };
        float av[MT][4], an[MT][4], bv[4][NT], bn[4][NT];
        int kof1[4], kof2[4];
        const int nkb_run = (kvalid + 15) >> 4;  // k-blocks of the chunk that hold any weight
        load_a(0, av);
        load_k(0, kof1);
        load_b(kof1, bv);
        load_k(1, kof1);
        for (int kb = 0; kb < nkb_run; ++kb) {
            if (kb + 1 < nkb_run) {
                load_a(kb + 1, an);
                load_b(kof1, bn);
            }
            load_k(kb + 2, kof2);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][t], bv[t][nt], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                kof1[t] = kof2[t];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m][t] = an[m][t];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[t][nt] = bn[t][nt];
            }
        }
    }

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (oyx[nt] < 0) continue;
            const int oy = oyx[nt] >> 16, ox = oyx[nt] & 0xffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = m0 + 16 * m + 4 * kq + r;
                if (ch < a.M) {
                    const long long o = (((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox;
                    a.out[o] = a.res ? acc[m][nt][r] + a.res[o] : acc[m][nt][r];
                }
            }
        }
}

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), the same bits in each of them
__device__ __forceinline__ float row16_sum(float v)
{
    v = v + bn_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + bn_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + bn_dpp<0x141>(v);  // row_half_mirror
    v = v + bn_dpp<0x140>(v);  // row_mirror
    return v;
}

// Persistent variant of the forward / dgrad GEMM for tiles that span the whole output width
// (all ResNet-20 shapes).  Per workgroup, ONCE: the k -> LDS-offset table and the weight panel of
// its 16*MT output channels, stored in LDS in MFMA fragment order (one conflict-free ds_read_b128
// per k-block and m-tile; hard-mode codes are decoded here).  Then it walks over tiles (image, row
// band) with stride gridDim.x: the input of the NEXT tile is loaded into registers while the MFMAs
// of the current tile run, and the main loop touches only LDS, so those loads stay in flight.
// CH4 = input channels per wavefront, J = plane elements per lane.
// position of a workgroup in the grid its kernel body was written for (a launch that runs several bodies side by
// side hands every workgroup the coordinates it would have had in a launch of its own)
struct BlockCoord {
    int x, y, z, gx;
};

template <int MT, int NT, int CH4, int J, int MODE>
__device__ __forceinline__ void conv_gemm_persist_body(const ConvArgs &a, const int ntiles, const int pitch_mul, const int tw_shift,
                                                       const unsigned k4n_rcp, const BlockCoord bc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *ktab = reinterpret_cast<int *>(smem);       // [Kpad]
    float *As = smem + a.Kpad;                       // [Kpad/16][MT][64 lanes][4]
    float *xs = As + MT * 16 * a.Kpad;               // [KC][rows_in][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;
    const int m0 = bc.y * 16 * MT;

    const bool par = MODE == MODE_DGRAD && a.par != 0;
    for (int k = tid; k < a.Kpad; k += 256) {
        int off = 0;
        if (par) {
            // k = k0(class) + c * nt(class) + ti with k0 = KC * {0, 1, 3, 5}, nt = {1, 2, 2, 4}; class = 2*py + px;
            // within a class the taps run over ky in {1} (py = 0) or {0, 2} (py = 1), then kx likewise;
            // tap ky = 0 reads the dy row below (offset 1), ky = 1, 2 the same row
            const int cls = k < a.KC ? 0 : (k < 3 * a.KC ? 1 : (k < 5 * a.KC ? 2 : 3));
            const int kk = k - a.KC * (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5)));
            const int sh = cls == 0 ? 0 : (cls == 3 ? 2 : 1);
            const int c = kk >> sh, ti = kk & ((1 << sh) - 1);
            const int py = cls >> 1, px = cls & 1;
            const int tiy = px ? ti >> 1 : ti, tix = px ? ti & 1 : 0;
            off = c * a.plane + ((py && tiy == 0) ? a.pitch : 0) + ((px && tix == 0) ? 1 : 0);
        } else if (k < a.K) {
            const int c = k / 9, tap = k - 9 * c;
            off = c * a.plane + (tap / 3) * a.pitch + (tap % 3);
        }
        ktab[k] = off;
    }
    // plane geometry of this lane's elements: row inside the image, input column (or -1)
    int rj[J], gxj[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int e = lane + 64 * j;
        const int r = (e * pitch_mul) >> 16;  // e / pitch, exact for e < 1024, pitch <= 65
        int gx = e - r * a.pitch - a.halo;
        bool ok = gx >= 0;
        if (a.dil == 2) {
            ok = ok && !(gx & 1);
            gx >>= 1;
        }
        ok = ok && gx < a.IW;
        rj[j] = e < a.plane ? r : -1;
        gxj[j] = ok ? gx : -1;
    }
    // output pixels of this lane
    int pixbase[NT], pyx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pr = (wave * NT + t) * 16 + jn;
        const int py = pr >> tw_shift, px = pr & (a.TW - 1);
        pixbase[t] = py * a.rs * a.pitch + px * a.rs;
        pyx[t] = px < (par ? a.OW >> 1 : a.OW) ? ((py << 16) | px) : -1;
    }
    const int chs = a.IH * a.IW;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // wavefront index as a scalar

    float sv[CH4][J];
    unsigned inmask = 0;  // bit j: element j of the tile being fetched lies inside the tensor
    auto fetch_tile = [&](int tile) {
        const int n = tile / a.tiles_y;
        int ty = tile - n * a.tiles_y;
        if (par) ty -= (ty / a.tyc) * a.tyc;  // row band inside its class
        const int r0 = ty * a.TH * a.rs - a.halo;
        const __amdgpu_buffer_rsrc_t src = make_rsrc(a.in + (long long)n * a.KC * chs, (unsigned)(a.KC * chs) * 4u);
        unsigned off[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            int gy = r0 + rj[j];
            bool ok = rj[j] >= 0 && gxj[j] >= 0 && gy >= 0;
            if (a.dil == 2) {
                ok = ok && !(gy & 1);
                gy >>= 1;
            }
            ok = ok && gy < a.IH;
            off[j] = ok ? (unsigned)(gy * a.IW + gxj[j]) * 4u : BUF_OOB;
            inmask = j == 0 ? (unsigned)ok : (inmask | ((unsigned)ok << j));
        }
#pragma unroll
        for (int u = 0; u < CH4; ++u) {
            // channels beyond KC are never committed: clamp so that the scalar offset stays inside the tensor
            const int c = min(wave_s + 4 * u, a.KC - 1);
            const unsigned soff = (unsigned)(c * chs) * 4u;
#pragma unroll
            for (int j = 0; j < J; ++j) sv[u][j] = buf_load(src, off[j], soff);
        }
    };
    float bsc[CH4], bsh[CH4];  // scale / shift of this wavefront's input channels (uniform)
    auto commit_tile = [&]() {
#pragma unroll
        for (int u = 0; u < CH4; ++u) {
            const int c = wave + 4 * u;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float v = sv[u][j];
                if (MODE == MODE_FWD && a.bn_on) v = ((inmask >> j) & 1u) ? fmaxf(__builtin_fmaf(v, bsc[u], bsh[u]), 0.0f) : 0.0f;  // padding stays 0
                if (c < a.KC && rj[j] >= 0) xs[c * a.plane + lane + 64 * j] = v;
            }
        }
    };

    // the first tile's loads are issued before the weight panel's, so both are in flight together
    int tile = bc.x;
    if (tile < ntiles) fetch_tile(tile);

    // weight panel in fragment order: element (row rl, k) lives at
    //   As[(((k>>4)*MT + (rl>>4))*64 + ((k>>2)&3)*16 + (rl&15))*4 + (k&3)]
    // so the 4 consecutive k of one row form one 16-byte LDS store.  Forward: work item = (row, group of
    // 4 k); all items of a thread are loaded before the first store (the fill is latency-bound).
    auto as_index = [&](int rl, int k) { return (((k >> 4) * MT + (rl >> 4)) * 64 + ((k >> 2) & 3) * 16 + (rl & 15)) * 4 + (k & 3); };
    if (MODE == MODE_DGRAD || MODE == MODE_FWD) {
        // forward: A[row = co, k = (ci, tap)] = W[co, ci, tap];  dgrad: A[row = ci, k = (co, tap)] =
        // W[co, ci, 8 - tap].  Work item = (channel c of the k axis, row), whose nine taps are contiguous in
        // memory; all loads of a round are issued before the first LDS store
        const __amdgpu_buffer_rsrc_t wsrc = make_rsrc(a.W, (unsigned)(a.Cout * a.Cin * 9) * 4u);
        const int items = 16 * MT * a.KC;
        for (int base = 0; base < items; base += 256 * 4) {
            float w[4][9];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = base + u * 256 + tid;
                const int c = it / (16 * MT), rl = it - c * (16 * MT);
                const int row = m0 + rl;
                const unsigned o = (it < items && row < a.M) ? (unsigned)((MODE == MODE_DGRAD ? c * a.Cin + row : row * a.Cin + c) * 9) * 4u
                                                             : BUF_OOB;
#pragma unroll
                for (int tw = 0; tw < 9; ++tw) w[u][tw] = buf_load(wsrc, o + 4u * tw, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = base + u * 256 + tid;
                const int c = it / (16 * MT), rl = it - c * (16 * MT);
                if (it < items) {
#pragma unroll
                    for (int tw = 0; tw < 9; ++tw) {
                        int k = c * 9 + (MODE == MODE_DGRAD ? 8 - tw : tw);
                        if (par) {  // tap (ky, kx) = tw of W[co, ci] goes to its parity class's share of the k axis
                            const int ky = tw / 3, kx = tw % 3;
                            const int py = ky != 1, px = kx != 1;
                            const int cls = 2 * py + px, sh = py + px;
                            const int ti = (ky == 2 ? 1 : 0) * (px ? 2 : 1) + (kx == 2 ? 1 : 0);
                            k = a.KC * (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5))) + (c << sh) + ti;
                        }
                        As[as_index(rl, k)] = w[u][tw];
                    }
                }
            }
        }
        const int padk = a.Kpad - a.K;  // zero weights for the padded k
        for (int it = tid; it < 16 * MT * padk; it += 256) {
            const int rl = it / padk, k = a.K + it - rl * padk;
            As[as_index(rl, k)] = 0.0f;
        }
    } else {
        constexpr int PW = 10;  // items per thread and round: one round covers the ResNet-20 panels
        const int k4n = a.Kpad >> 2, items = 16 * MT * k4n;
        const __amdgpu_buffer_rsrc_t wsrc = make_rsrc(a.W, (MODE == MODE_FWD && a.vecA) ? (unsigned)(a.M * a.K) * 4u : 0u);
        for (int base = 0; base < items; base += 256 * PW) {
            float4 w[PW];
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int it = base + u * 256 + tid;
                const int rl = (int)__umulhi((unsigned)it, k4n_rcp), k = (it - rl * k4n) * 4;
                const int row = m0 + rl;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == MODE_FWD && a.vecA) {
                    const unsigned o = (it < items && row < a.M && k < a.K) ? (unsigned)(row * a.K + k) * 4u : BUF_OOB;
                    v = make_float4(buf_load(wsrc, o, 0), buf_load(wsrc, o + 4u, 0), buf_load(wsrc, o + 8u, 0), buf_load(wsrc, o + 12u, 0));
                } else if (it < items && row < a.M && k < a.K) {
                    float e[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk = k + t;
                        float x = 0.0f;
                        if (kk < a.K) x = (MODE == MODE_FWD_CODES) ? a.q[a.idx[(long long)row * a.K + kk]] : a.W[(long long)row * a.K + kk];
                        e[t] = x;
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
                w[u] = v;
            }
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int it = base + u * 256 + tid;
                if (it < items) {
                    const int rl = (int)__umulhi((unsigned)it, k4n_rcp), k = (it - rl * k4n) * 4;
                    *reinterpret_cast<float4 *>(&As[as_index(rl, k)]) = w[u];
                }
            }
        }
    }

    // batch-norm input: every wavefront merges the statistics of the channels it stages (same operation
    // sequence as the stand-alone kernel); the first workgroup also records them
    if (MODE == MODE_FWD && a.bn_on) {
        int cs[CH4];
        float mean[CH4], rstd[CH4];
#pragma unroll
        for (int u = 0; u < CH4; ++u) cs[u] = wave_s + 4 * u < a.KC ? wave_s + 4 * u : -1;
        float gam[CH4], bet[CH4];  // all loads first: one memory round trip, not one per channel
#pragma unroll
        for (int u = 0; u < CH4; ++u) {
            gam[u] = a.bn.gamma[cs[u] >= 0 ? cs[u] : 0];
            bet[u] = a.bn.beta[cs[u] >= 0 ? cs[u] : 0];
        }
        bn_channel_stats_multi<CH4>(a.bn, a.bng, cs, lane, bc.x == 0 && bc.y == 0, mean, rstd);
        float sc[CH4], sh[CH4];
#pragma unroll
        for (int u = 0; u < CH4; ++u) scale_shift(gam[u], bet[u], mean[u], rstd[u], sc[u], sh[u]);
#pragma unroll
        for (int u = 0; u < CH4; ++u) {
            bsc[u] = sc[u];
            bsh[u] = sh[u];
        }
    }

    // k-blocks a tile walks: all of them, or (parity classes) the share of the tile's class; set per tile
    int kb_lo = 0, nkb = a.Kpad >> 4;
    auto load_a = [&](int kb, float (&av)[MT][4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 w4 = *reinterpret_cast<const float4 *>(&As[(((kb_lo + kb) * MT + m) * 64 + lane) * 4]);
            av[m][0] = w4.x; av[m][1] = w4.y; av[m][2] = w4.z; av[m][3] = w4.w;
        }
    };
    auto load_k = [&](int kb, int (&kof)[4]) {
        const int4 ko = *reinterpret_cast<const int4 *>(&ktab[16 * (kb_lo + (kb < nkb ? kb : nkb - 1)) + 4 * kq]);
        kof[0] = ko.x; kof[1] = ko.y; kof[2] = ko.z; kof[3] = ko.w;
    };
    auto load_b = [&](const int (&kof)[4], float (&bv)[4][NT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[t][nt] = xs[pixbase[nt] + kof[t]];
    };

    // channel statistics of the output (stat_part): every lane sums the values it stores, over all tiles
    float ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.0f;
    const bool stats_on = MODE != MODE_DGRAD && a.stat_part != nullptr;
    const bool bnb_on = MODE == MODE_DGRAD && a.bwd_part != nullptr;
    float bsc_o[MT][4], bsh_o[MT][4], bmu_o[MT][4];  // scale, shift, mean of this lane's output channels
    if (bnb_on) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = min(m0 + 16 * m + 4 * kq + r, a.M - 1);
                const float mu = a.bmean[ch];
                scale_shift(a.bgamma[ch], a.bbeta[ch], mu, a.brstd[ch], bsc_o[m][r], bsh_o[m][r]);
                bmu_o[m][r] = mu;
            }
    }

    int tcount_ = 0;
    for (; tile < ntiles; tile += bc.gx) {
        commit_tile();
        __syncthreads();  // image (and, the first time, tables and weight panel) visible
        const int next = tile + bc.gx;
        if (next < ntiles) fetch_tile(next);  // in flight during the MFMAs below
        // the epilogue's second operand (the shortcut added to the output, or the batch-norm input whose
        // backward sums are reduced) is requested now as well, not after the MFMAs
        const int n = tile / a.tiles_y;
        int ty = tile - n * a.tiles_y;
        int os = 1, cpy = 0, cpx = 0;  // output pixel = os * (tile pixel) + (cpy, cpx)
        if (par) {
            const int cls = ty / a.tyc;
            ty -= cls * a.tyc;
            os = 2; cpy = cls >> 1; cpx = cls & 1;
            kb_lo = (a.KC * (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5)))) >> 4;
            nkb = (a.KC << (cpy + cpx)) >> 4;
        }
        const int oy0 = ty * a.TH;
        const int ohw = a.OH * a.OW;
        const long long img = (long long)n * a.M * ohw;
        const bool eop = bnb_on || a.res != nullptr;  // (the host never sets both)
        float ev[NT][MT][4];
        if (eop) {
            const __amdgpu_buffer_rsrc_t esrc = make_rsrc((bnb_on ? a.bx : a.res) + img, (unsigned)(a.M * ohw) * 4u);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int oy = (oy0 + (pyx[nt] >> 16)) * os + cpy, ox = (pyx[nt] & 0xffff) * os + cpx;
                const bool pok = pyx[nt] >= 0 && oy < a.OH;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = m0 + 16 * m + 4 * kq + r;
                        ev[nt][m][r] = buf_load(esrc, (pok && ch < a.M) ? (unsigned)(ch * ohw + oy * a.OW + ox) * 4u : BUF_OOB, 0);
                    }
            }
        }

        floatx4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = floatx4{0.f, 0.f, 0.f, 0.f};
        // Two register sets (0/1) alternate over k-blocks, the loop body has no branches and the number
        // of k-blocks is even (K is padded with zero weights): while the MFMAs of one set run, the LDS
        // reads of the other set are in flight, and the compiler can emit counted lgkmcnt waits.
        float av0[MT][4], av1[MT][4], bv0[4][NT], bv1[4][NT];
        int kofA[4], kofB[4];
        auto mfmas = [&](const float (&av)[MT][4], const float (&bv)[4][NT]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][t], bv[t][nt], acc[m][nt], 0, 0, 0);
        };
        load_k(0, kofA);
        load_k(1, kofB);
        load_a(0, av0);
        load_b(kofA, bv0);
        const int nkb_run = nkb;
        for (int kb = 0; kb < nkb_run; kb += 2) {
            // sched_barrier: keep each group of LDS reads ahead of the other set's MFMAs (the
            // scheduler otherwise sinks the reads next to their consumers and the wave stalls)
            load_a(kb + 1, av1);
            load_b(kofB, bv1);
            load_k(kb + 2, kofA);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(av0, bv0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(kb + 2 < nkb ? kb + 2 : nkb - 1, av0);
            load_b(kofA, bv0);
            load_k(kb + 3, kofB);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t dst = make_rsrc(a.out + img, (unsigned)(a.M * ohw) * 4u);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int oy = (oy0 + (pyx[nt] >> 16)) * os + cpy, ox = (pyx[nt] & 0xffff) * os + cpx;
            const bool pok = pyx[nt] >= 0 && oy < a.OH;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = m0 + 16 * m + 4 * kq + r;
                    const unsigned o = (pok && ch < a.M) ? (unsigned)(ch * ohw + oy * a.OW + ox) * 4u : BUF_OOB;
                    float v = acc[m][nt][r];
                    if (a.res && !bnb_on) v += ev[nt][m][r];
                    buf_store(v, dst, o, 0);
                    if (stats_on) {  // the host enables this for exact tilings only: every pixel is an output
                        ssum[m][r] += v;
                        ssq[m][r] = __builtin_fmaf(v, v, ssq[m][r]);
                    }
                    if (bnb_on) {  // exact tilings only; channels beyond M hold zeros
                        const float xv = ev[nt][m][r];
                        const float gq = __builtin_fmaf(xv, bsc_o[m][r], bsh_o[m][r]) > 0.0f ? v : 0.0f;
                        ssum[m][r] += gq;
                        ssq[m][r] = __builtin_fmaf(gq, xv - bmu_o[m][r], ssq[m][r]);
                    }
                }
        }
        __syncthreads();  // every wavefront is done with this tile's LDS image
        ++tcount_;
    }
    if (stats_on || bnb_on) {
        // lanes of a 16-lane row share their channels: row sums by DPP, the four wavefronts through LDS
        // (the weight panel is no longer needed), then one thread per channel in fixed order
        float *red = As;  // [4 wavefronts][16*MT channels][2]
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row16_sum(ssum[m][r]), s2 = row16_sum(ssq[m][r]);
                if (jn == 0) {
                    const int cl = 16 * m + 4 * kq + r;
                    red[(wave * 16 * MT + cl) * 2] = s1;
                    red[(wave * 16 * MT + cl) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (tid < 16 * MT && m0 + tid < a.M) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s1 += red[(w * 16 * MT + tid) * 2];
                s2 += red[(w * 16 * MT + tid) * 2 + 1];
            }
            if (bnb_on) {
                a.bwd_part[(m0 + tid) * bc.gx + bc.x] = make_float2(s1, s2 * a.brstd[m0 + tid]);
            } else {
                const float cnt = (float)(tcount_ * a.TH * a.TW);
                const float mean = s1 / cnt;
                a.stat_part[(m0 + tid) * bc.gx + bc.x] = make_float2(mean, fmaxf(__builtin_fmaf(-s1, mean, s2), 0.0f));
            }
        }
    }
}

template <int MT, int NT, int CH4, int J, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_persist_kernel(const ConvArgs a, const int ntiles, const int pitch_mul, const int tw_shift,
                                                                const unsigned k4n_rcp)
{
    conv_gemm_persist_body<MT, NT, CH4, J, MODE>(a, ntiles, pitch_mul, tw_shift, k4n_rcp,
                                                 BlockCoord{(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x});
}

// ------------------------------------------------------------------------------------------ wgrad

struct WgradArgs {
    const float *x, *dy;
    float *part;  // [chunks_groups][Cout*Cin*9]
    // when bn_mean is set, x is a raw tensor and the operand is relu(batch_norm(x)) with these statistics
    const float *bn_gamma, *bn_beta, *bn_mean, *bn_rstd;
    int N, Cin, H, Wd, Cout, stride, Ho, Wo;
    int ks, ks2, pad;            // kernel size, taps per channel, padding (3, 9, 1 for the reference's layers)
    int K9;                      // Cin*ks2: columns (ci, tap) of the weight-gradient matrix
    int TW, TH, tiles_x, tiles_y;
    int pitch, rows_in, plane;   // LDS image of x
    int cpitch;                  // LDS pitch of a dy channel (TH*TW + 2)
    int CW;                      // input channels a workgroup stages: those its (ci, tap) columns touch
    int tw_shift, npix_shift;    // TW and TH*TW are powers of two
    int pitch_mul;               // e / pitch == (e * pitch_mul) >> 16 for e < 1024, pitch <= 65
    int IPB;                     // images per workgroup
    int ngroups_img;             // ceil(N / IPB)
};

// Workgroup = 8 wavefronts: wavefronts 0-3 compute (MFMA), wavefronts 4-7 load.  The chunks of a
// workgroup (TH x TW pixels of one image each) flow through two LDS buffers: while the compute
// wavefronts work on chunk i, the loader wavefronts fill the other buffer with chunk i+1; one
// barrier per chunk.  So global-memory latency is hidden behind the matrix work of the same CU.
// CH4 > 0 (tiles spanning the output width): the loader wavefronts keep the next chunk in registers
// (CH4 input channels per wavefront, J image elements per lane) for one phase before writing it to
// LDS, so their loads are in flight during a whole compute phase.  CH4 == 0: generic loader.
template <int MG, int NG, int CH4, int J>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs &a, const BlockCoord bc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npix = a.TH * a.TW;
    const int bufsz = 16 * MG * a.cpitch + a.CW * a.plane;  // floats per buffer: dy tile, then x image

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const bool loader = threadIdx.x >= 256;
    const int jn = lane & 15, kq = lane >> 4;

    int bx = bc.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int ig = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int m0 = bc.y * 16 * MG;   // first output channel of this group
    const int j0 = bc.z * 16 * NG;   // first (ci, tap) column of this group
    const int c_lo = a.ks2 == 9 ? j0 / 9 : j0 / a.ks2;  // first input channel those columns touch
    const int ceff = min(a.Cin - c_lo, a.CW);  // channels staged by this workgroup

    // compute wavefronts: column (ci, tap) owned by this lane in each n-tile -> offset inside the x image
    int joff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int jcol = j0 + 16 * g + jn;
        int off = 0;
        if (jcol < a.K9) {
            if (a.ks2 == 9) {  // (divisions by constants)
                const int ci = jcol / 9, tap = jcol - 9 * ci;
                off = (ci - c_lo) * a.plane + (tap / 3) * a.pitch + (tap % 3);
            } else {
                const int ci = jcol / a.ks2, tap = jcol - a.ks2 * ci;
                off = (ci - c_lo) * a.plane + (tap / a.ks) * a.pitch + (tap % a.ks);
            }
        }
        joff[g] = off;
    }
    // pixel 4t+kq of a 16-pixel block (raster order inside the TH x TW tile)
    int toff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int pp = 4 * t + kq;
        const int py = pp >> a.tw_shift, px = pp & (a.TW - 1);
        toff[t] = py * a.stride * a.pitch + px * a.stride;
    }

    floatx4 acc[MG][NG];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[m][g] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int r0 = oy0 * a.stride - a.pad, c0 = ox0 * a.stride - a.pad;
    const int nblk16 = npix >> 4;
    int nch = a.N - ig * a.IPB;
    nch = nch < a.IPB ? nch : a.IPB;  // chunks (images) of this workgroup

    // loader state (register-staged variant): the tile position is fixed for the workgroup, only the
    // image changes from chunk to chunk, so all offsets are computed once
    constexpr int CH4R = CH4 > 0 ? CH4 : 1, JR = CH4 > 0 ? J : 1;
    unsigned xvo[JR];      // register-staged variant: byte offset inside a channel, or BUF_OOB
    bool xin[JR];          // element belongs to the plane
    unsigned dvo[8 * MG];  // byte offset of this thread's dy elements inside an image, or BUF_OOB
    constexpr bool DEEP_SETS = CH4 > 0 && MG * NG <= 9 && CH4 * J <= 16;
    float svA[CH4R][JR], dvA[8 * MG];                                        // staging set A
    float svB[DEEP_SETS ? CH4R : 1][DEEP_SETS ? JR : 1], dvB[DEEP_SETS ? 8 * MG : 1];  // set B (two chunks in flight)
    const int pr = tid & (npix - 1), cstep = 256 / npix;
    const int chs = a.Ho * a.Wo, xchs = a.H * a.Wd;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    int poff = -1;
    if (loader) {
        const int py = pr >> a.tw_shift, px = pr & (a.TW - 1);
        const int oy = oy0 + py, ox = ox0 + px;
        poff = (oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
        if (CH4 > 0) {
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                const int e = lane + 64 * j;
                const int r = (e * a.pitch_mul) >> 16, cc = e - r * a.pitch;
                const int gy = r0 + r, gx = c0 + cc;
                const bool ok = e < a.plane && gy >= 0 && gx >= 0 && gy < a.H && gx < a.Wd;
                xvo[j] = ok ? (unsigned)(gy * a.Wd + gx) * 4u : BUF_OOB;
                xin[j] = e < a.plane;
            }
#pragma unroll
            for (int u = 0; u < 8 * MG; ++u) {
                const int c = (tid >> a.npix_shift) + cstep * u;
                const int co = m0 + c;
                dvo[u] = (c < 16 * MG && poff >= 0 && co < a.Cout) ? (unsigned)(co * chs + poff) * 4u : BUF_OOB;
            }
        }
    }
    auto fetch_chunk = [&](int im, auto &sv, auto &dvr) {
        const int n = ig * a.IPB + im;
        const __amdgpu_buffer_rsrc_t sdy = make_rsrc(a.dy + (long long)n * a.Cout * chs, (unsigned)(a.Cout * chs) * 4u);
        const __amdgpu_buffer_rsrc_t sx = make_rsrc(a.x + (long long)n * a.Cin * xchs, (unsigned)(a.Cin * xchs) * 4u);
#pragma unroll
        for (int u = 0; u < 8 * MG; ++u) dvr[u] = buf_load(sdy, dvo[u], 0);
#pragma unroll
        for (int u = 0; u < CH4R; ++u) {
            const int c = c_lo + min(wave_s + 4 * u, ceff - 1);  // slots beyond the staged channels are never committed
            const unsigned soff = (unsigned)(c * xchs) * 4u;
#pragma unroll
            for (int j = 0; j < JR; ++j) sv[u][j] = buf_load(sx, xvo[j], soff);
        }
    };
    float wsc[CH4R], wsh[CH4R];
    if (CH4 > 0 && loader && a.bn_mean) {
#pragma unroll
        for (int u = 0; u < CH4R; ++u) {
            const int c = c_lo + min(wave_s + 4 * u, ceff - 1);
            scale_shift(a.bn_gamma[c], a.bn_beta[c], a.bn_mean[c], a.bn_rstd[c], wsc[u], wsh[u]);
        }
    }
    auto commit_chunk = [&](int im, const auto &sv, const auto &dvr) {
        float *dys = smem + (im & 1) * bufsz;
        float *xs = dys + 16 * MG * a.cpitch;
#pragma unroll
        for (int u = 0; u < 8 * MG; ++u) {
            const int c = (tid >> a.npix_shift) + cstep * u;
            if (c < 16 * MG) dys[c * a.cpitch + pr] = dvr[u];
        }
#pragma unroll
        for (int u = 0; u < CH4R; ++u) {
            const int c = wave + 4 * u;
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                float v = sv[u][j];
                if (a.bn_mean) v = xvo[j] != BUF_OOB ? fmaxf(__builtin_fmaf(v, wsc[u], wsh[u]), 0.0f) : 0.0f;  // padding stays 0
                if (c < ceff && xin[j]) xs[c * a.plane + lane + 64 * j] = v;
            }
        }
    };
    // compute wavefronts: one chunk (TH x TW pixels of one image) from LDS buffer (cidx & 1)
    auto compute_chunk = [&](int cidx) {
        const float *dys = smem + (cidx & 1) * bufsz;
        const float *xs = dys + 16 * MG * a.cpitch;
        // steps: (pixel block pb = wave, wave+4, ..) x (t = 0..3); two register sets alternate so that
        // the LDS reads of the next step are in flight during the MFMAs of the current one
        const int nsteps = ((nblk16 - wave + 3) >> 2) * 4;
        float av0[MG], av1[MG], bv0[NG], bv1[NG];
        auto load_step = [&](int sidx, float (&av)[MG], float (&bv)[NG]) {
            const int s2 = sidx < nsteps ? sidx : nsteps - 1;
            const int p16 = (wave + 4 * (s2 >> 2)) * 16, t = s2 & 3;
            const int py = p16 >> a.tw_shift, px = p16 & (a.TW - 1);
            const int bo = py * a.stride * a.pitch + px * a.stride + toff[t];
#pragma unroll
            for (int m = 0; m < MG; ++m) av[m] = dys[(16 * m + jn) * a.cpitch + p16 + 4 * t + kq];
#pragma unroll
            for (int g = 0; g < NG; ++g) bv[g] = xs[bo + joff[g]];
        };
        auto mfmas = [&](const float (&av)[MG], const float (&bv)[NG]) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[g], acc[m][g], 0, 0, 0);
        };
        if (MG * NG <= 9) {
            if (nsteps > 0) {
                load_step(0, av0, bv0);
                for (int sidx = 0; sidx < nsteps; sidx += 2) {  // nsteps is a multiple of 4
                    load_step(sidx + 1, av1, bv1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(av0, bv0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_step(sidx + 2, av0, bv0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(av1, bv1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (nsteps > 0) {
            // large tile groups: one operand set (register budget), pipelined by halves: while the
            // MFMAs of one half of the n-tiles run, the LDS reads of the other half are in flight
            constexpr int H1 = NG / 2;
            auto step_offsets = [&](int sidx, int &ao, int &bo) {
                const int s2 = sidx < nsteps ? sidx : nsteps - 1;
                const int p16 = (wave + 4 * (s2 >> 2)) * 16, t = s2 & 3;
                const int py = p16 >> a.tw_shift, px = p16 & (a.TW - 1);
                bo = py * a.stride * a.pitch + px * a.stride + toff[t];
                ao = p16 + 4 * t + kq;
            };
            auto load_av = [&](int sidx, float (&av)[MG]) {
                int ao, bo;
                step_offsets(sidx, ao, bo);
#pragma unroll
                for (int m = 0; m < MG; ++m) av[m] = dys[(16 * m + jn) * a.cpitch + ao];
            };
            auto load_bv = [&](int sidx, int g0, int g1) {
                int ao, bo;
                step_offsets(sidx, ao, bo);
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if (g >= g0 && g < g1) bv0[g] = xs[bo + joff[g]];
            };
            auto mfma_part = [&](const float (&av)[MG], int g0, int g1) {
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        if (g >= g0 && g < g1) acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv0[g], acc[m][g], 0, 0, 0);
            };
            load_av(0, av0);
            load_bv(0, 0, H1);
            for (int sidx = 0; sidx < nsteps; sidx += 2) {  // nsteps is a multiple of 4
                load_bv(sidx, H1, NG);
                load_av(sidx + 1, av1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_part(av0, 0, H1);
                __builtin_amdgcn_sched_barrier(0);
                load_bv(sidx + 1, 0, H1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_part(av0, H1, NG);
                __builtin_amdgcn_sched_barrier(0);
                load_bv(sidx + 1, H1, NG);
                load_av(sidx + 2, av0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_part(av1, 0, H1);
                __builtin_amdgcn_sched_barrier(0);
                load_bv(sidx + 2, 0, H1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_part(av1, H1, NG);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // loader wavefronts, generic variant: chunk im straight into LDS buffer (im & 1)
    auto load_generic = [&](int im) {
        const int n = ig * a.IPB + im;
        float *dys = smem + (im & 1) * bufsz;
        float *xs = dys + 16 * MG * a.cpitch;
        // dy tile: [16*MG channels][npix], zero outside the output / beyond Cout; the dy values
        // wait in registers while the x image loads are issued behind them (one round trip)
        const float *src = a.dy + (long long)n * a.Cout * a.Ho * a.Wo;
        const int chs = a.Ho * a.Wo;
#pragma unroll
        for (int u = 0; u < 8 * MG; ++u) {
            const int c = (tid >> a.npix_shift) + cstep * u;
            const int co = m0 + c;
            dvA[u] = (c < 16 * MG && poff >= 0 && co < a.Cout) ? src[(long long)co * chs + poff] : 0.0f;
        }
        stage_image(xs, a.x + ((long long)n * a.Cin + c_lo) * a.H * a.Wd, ceff, a.H, a.Wd, a.plane, a.pitch, r0, c0, 1, tid);
#pragma unroll
        for (int u = 0; u < 8 * MG; ++u) {
            const int c = (tid >> a.npix_shift) + cstep * u;
            if (c < 16 * MG) dys[c * a.cpitch + pr] = dvA[u];
        }
    };
    // Pipeline (register-staged variant).  DEEP (small tile groups, where registers allow a second staging
    // set): chunk p is fetched into registers during phase p-2, written to LDS buffer (p & 1) in phase p
    // and consumed in phase p+1, so a load has two phases to land (one phase is shorter than the memory
    // latency); phases come in pairs so that the register set of a chunk (A: even, B: odd) is a
    // compile-time choice.  Otherwise chunk p is fetched during phase p-1.
    constexpr bool DEEP = CH4 > 0 && MG * NG <= 9 && CH4 * J <= 16;
    if (CH4 > 0 && loader) {
        if (nch > 0) fetch_chunk(0, svA, dvA);
        if (DEEP && nch > 1) fetch_chunk(1, svB, dvB);
    }
    if constexpr (DEEP) {
        for (int im = 0; im <= nch; im += 2) {
            if (loader) {
                if (im < nch) commit_chunk(im, svA, dvA);
                if (im + 2 < nch) fetch_chunk(im + 2, svA, dvA);
            } else if (im >= 1) {
                compute_chunk(im - 1);
            }
            __syncthreads();  // buffer (im & 1) filled, buffer ((im-1) & 1) consumed
            if (im + 1 > nch) break;
            if (loader) {
                if (im + 1 < nch) commit_chunk(im + 1, svB, dvB);
                if (im + 3 < nch) fetch_chunk(im + 3, svB, dvB);
            } else {
                compute_chunk(im);
            }
            __syncthreads();
        }
    } else {
        for (int im = 0; im <= nch; ++im) {
            if (loader) {
                if (CH4 > 0) {
                    if (im < nch) commit_chunk(im, svA, dvA);          // loaded during the previous phase
                    if (im + 1 < nch) fetch_chunk(im + 1, svA, dvA);   // lands during the next compute phase
                } else if (im < nch) {
                    load_generic(im);
                }
            } else if (im >= 1) {
                compute_chunk(im - 1);
            }
            __syncthreads();  // buffer (im & 1) filled, buffer ((im-1) & 1) consumed
        }
    }

    // ---- sum the 4 compute wavefronts in LDS (fixed order 0+1+2+3), write the workgroup's slab ------
    float *part = a.part + (long long)bc.x * a.Cout * a.K9;
    constexpr int TPR = 8;  // tiles per round: 8 tiles * 4 waves * 256 floats = 32 KiB
    float *red = smem;
    for (int base = 0; base < MG * NG; base += TPR) {
        __syncthreads();
        if (!loader) {
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
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TPR * 64; e += 512) {
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

template <int MG, int NG, int CH4, int J>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradArgs a)
{
    conv_wgrad_body<MG, NG, CH4, J>(a, BlockCoord{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x});
}

template <int MG, int NG, int CH4, int J>
__global__ __launch_bounds__(512) void conv_wgrad_group_kernel(const WgradArgs a, const WgradGroup g)
{
    const int l = (int)blockIdx.x / g.gx;
    WgradArgs b = a;
    b.x = g.x[l];
    b.dy = g.dy[l];
    b.part = g.part[l];
    conv_wgrad_body<MG, NG, CH4, J>(b, BlockCoord{(int)blockIdx.x - l * g.gx, (int)blockIdx.y, (int)blockIdx.z, g.gx});
}

// Weight-gradient slabs for layers with many channels: a workgroup of four wavefronts owns 64 output channels x
// 16*NG (ci, tap) columns (for 3x3 kernels: 16 input channels exactly), wavefront w the 16 channels 16w..16w+15 with all
// NG column tiles in its accumulators, and walks over its share of the pixel chunks (TH x TW pixels of one image): the
// chunk's dy tile [64][pixels] and x image [channels touched][rows_in x pitch] are staged in LDS and every staged value
// feeds 9 (dy) or 4 (x) MFMAs, four times the reuse of the small-channel kernel.  Several workgroups share a CU, so
// one's staging overlaps another's matrix work.  Output: one slab per blockIdx.x, reduced by wgrad_reduce like the
// other kernel's.
template <int NG>
__global__ __launch_bounds__(256) void conv_wgrad_big_kernel(const WgradArgs a, const int nchunks)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npix = a.TH * a.TW;
    float *dys = smem;                      // [64][cpitch]
    float *xs = smem + 64 * a.cpitch;       // [CW][plane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 64;
    const int j0 = blockIdx.z * 16 * NG;
    const int c_lo = j0 / a.ks2;
    const int ceff = min(a.Cin - c_lo, a.CW);

    int joff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int jcol = j0 + 16 * g + jn;
        int off = 0;
        if (jcol < a.K9) {
            const int ci = jcol / a.ks2, tap = jcol - a.ks2 * ci;
            off = (ci - c_lo) * a.plane + (tap / a.ks) * a.pitch + (tap % a.ks);
        }
        joff[g] = off;
    }
    int toff[4];  // pixel 4t + kq of a 16-pixel block
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int pp = 4 * t + kq;
        const int py = pp >> a.tw_shift, px = pp & (a.TW - 1);
        toff[t] = py * a.stride * a.pitch + px * a.stride;
    }
    floatx4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};

    const long long chs = (long long)a.Ho * a.Wo, xchs = (long long)a.H * a.Wd;
    const int tiles = a.tiles_x * a.tiles_y;
    for (int q = blockIdx.x; q < nchunks; q += gridDim.x) {
        const int n = q / tiles;
        int tr = q - n * tiles;
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const int r0 = oy0 * a.stride - a.pad, c0 = ox0 * a.stride - a.pad;
        if (q != (int)blockIdx.x) __syncthreads();  // the previous chunk has been consumed
        // dy tile: [64 channels][npix], zero outside the output / beyond Cout
        const float *dsrc = a.dy + (long long)n * a.Cout * chs;
        for (int e0 = tid; e0 < 64 * npix; e0 += 256 * 8) {  // eight loads in flight per thread, then the LDS stores
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                const int c = e >> a.npix_shift, pr = e & (npix - 1);
                const int py = pr >> a.tw_shift, px = pr & (a.TW - 1);
                const int oy = oy0 + py, ox = ox0 + px, co = m0 + c;
                v[u] = (e < 64 * npix && oy < a.Ho && ox < a.Wo && co < a.Cout) ? dsrc[(long long)co * chs + oy * a.Wo + ox] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                if (e < 64 * npix) dys[(e >> a.npix_shift) * a.cpitch + (e & (npix - 1))] = v[u];
            }
        }
        // x image of the channels this workgroup's columns touch
        const float *xsrc = a.x + ((long long)n * a.Cin + c_lo) * xchs;
        for (int e = tid; e < a.plane; e += 256) {
            const int r = e / a.pitch, cc = e - r * a.pitch;
            const int gy = r0 + r, gx = c0 + cc;
            const bool ok = gy >= 0 && gx >= 0 && gy < a.H && gx < a.Wd;
            const float *src = xsrc + (ok ? gy * a.Wd + gx : 0);
            for (int c = 0; c < ceff; c += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (ok && c + u < ceff) ? src[(long long)(c + u) * xchs] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + u < ceff) xs[(c + u) * a.plane + e] = v[u];
            }
        }
        __syncthreads();
        // k = pixels: 16-pixel blocks, four MFMA steps each; two operand sets alternate
        const float *arow = dys + (16 * wave + jn) * a.cpitch + kq;
        const int nsteps = npix >> 2;  // multiple of 4
        float av0, av1, bv0[NG], bv1[NG];
        auto load_step = [&](int sidx, float &av, float (&bv)[NG]) {
            const int s2 = sidx < nsteps ? sidx : nsteps - 1;
            const int p16 = (s2 >> 2) * 16, t = s2 & 3;
            const int py = p16 >> a.tw_shift, px = p16 & (a.TW - 1);
            const int bo = py * a.stride * a.pitch + px * a.stride + toff[t];
            av = arow[p16 + 4 * t];
#pragma unroll
            for (int g = 0; g < NG; ++g) bv[g] = xs[bo + joff[g]];
        };
        load_step(0, av0, bv0);
        for (int sidx = 0; sidx < nsteps; sidx += 2) {
            load_step(sidx + 1, av1, bv1);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0[g], acc[g], 0, 0, 0);
            load_step(sidx + 2, av0, bv0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1[g], acc[g], 0, 0, 0);
        }
    }
    // D[row = 4 * kq + r][col = jn] of tile g -> slab
    float *part = a.part + (long long)blockIdx.x * a.Cout * a.K9;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int col = j0 + 16 * g + jn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wave + 4 * kq + r;
            if (row < a.Cout && col < a.K9) part[(long long)row * a.K9 + col] = acc[g][r];
        }
    }
}

// Both gradients of one convolution in ONE launch.  The input gradient (persistent GEMM, 4 wavefronts per
// workgroup) and the weight-gradient slabs (8 wavefronts) read the same dy and do not depend on each other;
// each of them alone leaves most of a CU idle (latency-bound at this problem size), so their workgroups run
// side by side on the same CUs.  Workgroups have 512 threads; in the
// input-gradient role the upper four wavefronts leave at once (a barrier counts live wavefronts only).
struct FusedGrid {
    int d_gx, d_gy;        // grid of the input-gradient body
    int w_gx, w_gy, w_gz;  // grid of the weight-gradient body
};
template <int MT, int NT, int CH4, int J, int MG, int NG, int WCH4, int WJ>
__global__ __launch_bounds__(512) void conv_bwd_fused_kernel(const ConvArgs a, const int ntiles, const int pitch_mul, const int tw_shift,
                                                             const unsigned k4n_rcp, const WgradArgs w, const FusedGrid fg)
{
    // Workgroups are handed out in index order, round-robin over the XCDs and their CUs, so a role must own a
    // contiguous run of about one workgroup per CU to land on every CU: the weight-gradient workgroups (the
    // longer ones) come first, the input-gradient workgroups fill in beside them.
    const int n2 = fg.w_gx * fg.w_gy * fg.w_gz, b = blockIdx.x;
    const int role = b < n2 ? 1 : 0, idx = b < n2 ? b : b - n2;
    if (role == 0) {
        if (threadIdx.x >= 256) return;
        const int by = idx / fg.d_gx;
        conv_gemm_persist_body<MT, NT, CH4, J, MODE_DGRAD>(a, ntiles, pitch_mul, tw_shift, k4n_rcp, BlockCoord{idx - by * fg.d_gx, by, 0, fg.d_gx});
    } else {
        const int bz = idx / (fg.w_gx * fg.w_gy), rem = idx - bz * (fg.w_gx * fg.w_gy);
        const int by = rem / fg.w_gx;
        conv_wgrad_body<MG, NG, WCH4, WJ>(w, BlockCoord{rem - by * fg.w_gx, by, bz, fg.w_gx});
    }
}

// dW[i] = sum over the workgroup slabs.  8 thread rows each add every 8th slab in ascending order, then the
// 8 running sums are added in order 0..7.  A workgroup owns 128 consecutive elements, four per thread as
// one 16-byte load per slab (a wavefront reads two 512-byte runs), when rows of n elements keep that
// alignment; otherwise 32 elements, one per thread.  Same order of additions either way.
__host__ __device__ inline bool wgrad_reduce_vec(const float *part, const float *dW, int n)
{
    return (n & 3) == 0 && (((uintptr_t)part | (uintptr_t)dW) & 15) == 0;
}
__host__ __device__ inline int wgrad_reduce_blocks(const float *part, const float *dW, int n)
{
    return wgrad_reduce_vec(part, dW, n) ? (n + 127) / 128 : (n + 31) / 32;
}

template <int V>
__device__ __forceinline__ void wgrad_reduce_cols(const float *__restrict__ part, float *__restrict__ dW, int n, int nslab, int blk, float (*sh)[32 * 4 + 1])
{
    const int c = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = (blk * 32 + c) * V;
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.0f;
    auto load = [&](int b, float (&v)[V]) {
        if constexpr (V == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(part + (long long)b * n + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            v[0] = part[(long long)b * n + i];
        }
    };
    if (i < n) {
        int b = sg;
        for (; b + 56 < nslab; b += 64) {  // eight independent loads in flight, added in ascending order
            float v[8][V];
#pragma unroll
            for (int u = 0; u < 8; ++u) load(b + 8 * u, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] += v[u][k];
        }
        for (; b < nslab; b += 8) {
            float v[V];
            load(b, v);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) sh[sg][c * V + k] = s[k];
    __syncthreads();
    if (sg == 0 && i < n) {
        float t[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            t[k] = sh[0][c * V + k];
#pragma unroll
            for (int g = 1; g < 8; ++g) t[k] += sh[g][c * V + k];
        }
        if constexpr (V == 4) *reinterpret_cast<float4 *>(dW + i) = make_float4(t[0], t[1], t[2], t[3]);
        else dW[i] = t[0];
    }
}

__device__ __forceinline__ void wgrad_reduce_block(const float *__restrict__ part, float *__restrict__ dW, int n, int nslab, int blk)
{
    __shared__ float sh[8][32 * 4 + 1];
    if (wgrad_reduce_vec(part, dW, n)) wgrad_reduce_cols<4>(part, dW, n, nslab, blk, sh);
    else wgrad_reduce_cols<1>(part, dW, n, nslab, blk, sh);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int n, int nslab)
{
    wgrad_reduce_block(part, dW, n, nslab, blockIdx.x);
}

// the slab reductions of several layers in one launch: job j owns blocks [first[j], first[j+1])
constexpr int WGRAD_MAX_JOBS = 32;
struct WgradJobTab {
    const float *part[WGRAD_MAX_JOBS];
    float *dW[WGRAD_MAX_JOBS];
    int n[WGRAD_MAX_JOBS], nslab[WGRAD_MAX_JOBS], first[WGRAD_MAX_JOBS + 1];
    int njobs;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgradJobTab t)
{
    int j = 0;
    while (j + 1 < t.njobs && (int)blockIdx.x >= t.first[j + 1]) ++j;
    wgrad_reduce_block(t.part[j], t.dW[j], t.n[j], t.nslab[j], blockIdx.x - t.first[j]);
}

// ------------------------------------------------------------------------------------- host side

// 3x3 / padding 1 is the reference's only configuration (adp_qtz_fan.py:574-601 is written for any ksize / stride /
// padding) and has the tuned kernels; other odd sizes up to 7x7 and any padding below the kernel size run on the
// general kernels.
static int check_conv(const char *who, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad)
{
    AQ_REQUIRE(ks >= 1 && ks <= 7 && (ks & 1) && pad >= 0 && pad < ks, -ENOTSUP,
               "%s: kernels of 1x1, 3x3, 5x5, 7x7 with padding below the kernel size are implemented (got k=%d pad=%d)", who, ks, pad);
    AQ_REQUIRE(stride == 1 || stride == 2, -ENOTSUP, "%s: stride must be 1 or 2 (got %d)", who, stride);
    AQ_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && Wd >= 1, -EINVAL, "%s: bad sizes", who);
    AQ_REQUIRE(H + 2 * pad >= ks && Wd + 2 * pad >= ks, -EINVAL, "%s: input %dx%d smaller than the kernel", who, H, Wd);
    AQ_REQUIRE(H < 32768 && Wd < 32768 && (long long)Cin * ks * ks < (1 << 20) && (long long)Cout * ks * ks < (1 << 20), -E2BIG,
               "%s: dimension too large", who);
    return 0;
}

static void pick_tile(int OW, int npix, int &TW, int &TH)
{
    TW = OW > 16 ? 32 : (OW > 8 ? 16 : 8);
    TH = npix / TW;
}

// The general kernel: any kernel size / padding / channel count (see conv_gemm_chunked_kernel).
template <int MODE>
static int launch_chunked(ConvArgs &a, int N, hipStream_t st, const char *who)
{
    const int mtiles = (a.M + 15) / 16;
    // stride-2 dgrad: by output parity classes (a quarter of the work); the class grid is half the output each way
    const bool par = MODE == MODE_DGRAD && a.dil == 2;
    const int gh = par ? (a.OH + 1) / 2 : a.OH, gw = par ? (a.OW + 1) / 2 : a.OW;
    const int nclass = par ? 4 : 1;
    a.par = par ? 1 : 0;
    a.halo = a.padT;
    if (par) a.rs = 1;
    // wavefronts along the output channels once there are 64 of them (weight fragments loaded once per workgroup)
    const int WM = mtiles >= 4 ? 4 : 1;
    int MT, NT;  // per wavefront
    if (WM == 4) {
        MT = mtiles >= 8 ? 2 : 1;
        NT = 8;
    } else {
        MT = mtiles >= 2 ? 2 : 1;
        NT = 2;
    }
    auto npix = [&](int nt) { return 16 * nt * (WM == 4 ? 1 : 4); };
    auto blocks = [&](int mt, int nt) {
        int TW, TH;
        pick_tile(gw, npix(nt), TW, TH);
        return (long long)N * ((gh + TH - 1) / TH) * ((gw + TW - 1) / TW) * nclass * ((mtiles + mt * WM - 1) / (mt * WM));
    };
    while (blocks(MT, NT) < 512) {  // fill the chip: smaller pixel tiles first, then fewer channels per workgroup
        if (WM == 4 && NT > 4) NT = 4;
        else if (WM == 1 && NT > 1) NT = 1;
        else if (MT > 1) MT = 1;
        else break;
    }
    {
        int tw, th;
        pick_tile(gw, npix(NT), tw, th);
        if (th > gh && ((WM == 4 && NT > 4) || (WM == 1 && NT > 1))) NT >>= 1;  // a tile taller than the grid would be half empty
    }
    pick_tile(gw, npix(NT), a.TW, a.TH);
    a.tiles_x = (gw + a.TW - 1) / a.TW;
    a.tiles_y = (gh + a.TH - 1) / a.TH;
    if (par) {  // undilated dy tile: one input pixel per output pixel of the class plus the taps' reach
        const int reach = (a.ks - 1) / 2 + 1;  // at most ceil(ks / 2) taps per dimension -> that many distinct offsets
        a.pitch = a.TW + reach;
        a.rows_in = a.TH + reach;
    } else {
        a.pitch = (a.TW - 1) * a.rs + a.ks;
        a.rows_in = (a.TH - 1) * a.rs + a.ks;
    }
    a.plane = a.rows_in * a.pitch;
    const int taps = par ? ((a.ks + 1) / 2) * ((a.ks + 1) / 2) : a.ks2;  // most taps any class walks per channel
    // channels per chunk: about 144 k per chunk (nine k-blocks between two barriers), within 48 KiB of LDS
    int CK = a.KC;
    if (a.KC >= 16) {
        const int want = std::max(16, ((144 / taps) + 15) & ~15);
        const int fit = (int)((48 * 1024) / ((size_t)a.plane * 4)) & ~15;
        CK = std::max(16, std::min(want, fit));
        CK = std::min(CK, (a.KC + 15) & ~15);
    }
    a.CK = CK;
    a.Kc = (CK * taps + 15) & ~15;
    a.Kpad = a.Kc;
    a.vecA = (a.K % 4 == 0) && (((uintptr_t)a.W & 15) == 0) && ((CK * a.ks2) % 4 == 0);
    const size_t lds = (size_t)a.Kc * 4 * (MODE == MODE_DGRAD ? 2 : 1) + (MODE == MODE_FWD_CODES ? ((a.L + 3) & ~3) * 4 : 0) + (size_t)CK * a.plane * 4;
    AQ_REQUIRE(lds <= 64 * 1024, -E2BIG, "%s: input tile needs %zu bytes of LDS (limit 65536)", who, lds);
    const long long gx = (long long)N * a.tiles_x * a.tiles_y;
    AQ_REQUIRE(gx < (1LL << 31), -E2BIG, "%s: grid too large", who);
    const unsigned gy = (unsigned)((mtiles + MT * WM - 1) / (MT * WM));
    AQ_REQUIRE(gy < 65536, -E2BIG, "%s: too many output channels", who);
    dim3 grid((unsigned)gx, gy, (unsigned)nclass);
#define AQ_CH(MM, NN, WW) hipLaunchKernelGGL((conv_gemm_chunked_kernel<MM, NN, MODE, WW>), grid, dim3(256), lds, st, a)
    if (WM == 4) {
        if (MT == 2 && NT == 8) AQ_CH(2, 8, 4);
        else if (MT == 2) AQ_CH(2, 4, 4);
        else if (NT == 8) AQ_CH(1, 8, 4);
        else AQ_CH(1, 4, 4);
    } else {
        if (MT == 2 && NT == 2) AQ_CH(2, 2, 1);
        else if (MT == 2) AQ_CH(2, 1, 1);
        else if (NT == 2) AQ_CH(1, 2, 1);
        else AQ_CH(1, 1, 1);
    }
#undef AQ_CH
    return check_launch(who);
}

// the many-channel 3x3 kernels live in aq_conv_c16.hip
template <int MODE>
static int launch_c3(const ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken)
{
    if constexpr (MODE == MODE_FWD) return ::aq_c16_launch_fwd(a, N, st, who, taken);
    else if constexpr (MODE == MODE_DGRAD) return ::aq_c16_launch_dgrad(a, N, st, who, taken);
    else {
        *taken = false;
        return 0;
    }
}

// dry: no launch; returns 1 when the persistent kernel (the one that can normalise its input) would run
// par: stride-2 dgrad by output parity classes (persistent kernel only); *fell = true when that kernel does not
// take the shape, nothing launched (the caller then plans again without par)
template <int MODE>
static int launch_gemm_impl(ConvArgs &a, int N, hipStream_t st, const char *who, bool dry, ConvStatsPlan *plan, bool par, bool *fell)
{
    if (plan) *plan = ConvStatsPlan{};
    // tile choice: NT = 2 (128 pixels) and MT up to 4, shrunk until the grid fills the chip
    const int mtiles = (a.M + 15) / 16;
    int MT = std::min(4, mtiles), NT = 2;
    if (MT == 3) MT = 2;
    // grid the tiles are cut from: the output, or (par) one parity class of it, four of them per image
    const int gw = par ? a.OW / 2 : a.OW, gh = par ? a.OH / 2 : a.OH, gmul = par ? 4 : 1;
    if (par) {
        a.par = 1; a.halo = 0; a.dil = 1; a.rs = 1;
    } else {
        a.par = 0; a.halo = 1;
    }
    auto blocks = [&](int mt, int nt) {
        int TW, TH;
        pick_tile(gw, 64 * nt, TW, TH);
        return (long long)N * ((gh + TH - 1) / TH) * ((gw + TW - 1) / TW) * gmul * ((mtiles + mt - 1) / mt);
    };
    while (blocks(MT, NT) < 512 && (NT > 1 || MT > 1)) {
        if (NT > 1) NT = 1;
        else MT >>= 1;
    }
    if (NT > 1) {  // a 128-pixel tile taller than the grid would be half empty
        int tw, th;
        pick_tile(gw, 128, tw, th);
        if (th > gh) NT = 1;
    }
    pick_tile(gw, 64 * NT, a.TW, a.TH);
    a.tiles_x = (gw + a.TW - 1) / a.TW;
    a.tyc = (gh + a.TH - 1) / a.TH;
    a.tiles_y = a.tyc * gmul;
    a.pitch = par ? a.TW + 1 : (a.TW - 1) * a.rs + 3;
    a.rows_in = par ? a.TH + 1 : (a.TH - 1) * a.rs + 3;
    a.plane = a.rows_in * a.pitch;
    a.Kpad = (a.K + 31) & ~31;  // even number of 16-wide k-blocks (main loops are unrolled by two)
    a.vecA = (a.K % 4 == 0) && (((uintptr_t)a.W & 15) == 0);
    size_t lds = (size_t)a.Kpad * 4 * (MODE == MODE_DGRAD ? 2 : 1) + (MODE == MODE_FWD_CODES ? ((a.L + 3) & ~3) * 4 : 0) +
                 (size_t)a.KC * a.plane * 4;
    const long long gx = (long long)N * a.tiles_x * a.tiles_y;
    AQ_REQUIRE(gx < (1LL << 31), -E2BIG, "%s: grid too large", who);
    const unsigned gy = (unsigned)((mtiles + MT - 1) / MT);
    // persistent kernel: full-width tiles, channels and plane within the register-staging budget
    const int CH4 = (a.KC + 3) / 4, J = (a.plane + 63) / 64;
    const size_t plds = ((size_t)a.Kpad + (size_t)MT * 16 * a.Kpad + (size_t)a.KC * a.plane) * 4;
    const bool fits32 = (long long)a.KC * a.IH * a.IW * 4 < (1LL << 31) && (long long)a.M * a.OH * a.OW * 4 < (1LL << 31) &&
                        (long long)a.Cout * a.Cin * 36 < (1LL << 31);  // buffer offsets
    if (a.tiles_x == 1 && CH4 <= 16 && J <= 9 && CH4 * J <= 48 && a.pitch <= 65 && plds <= 160 * 1024 && fits32) {
        const int pitch_mul = (65536 + a.pitch - 1) / a.pitch;
        const unsigned k4n_rcp = (unsigned)(((1ULL << 32) + (a.Kpad >> 2) - 1) / (a.Kpad >> 2));  // it / (Kpad/4) == umulhi(it, rcp) for it < 2^16
        int tw_shift = 0;
        while ((1 << tw_shift) < a.TW) ++tw_shift;
        // workgroups: as many as fit on the chip at once (LDS-limited, at most 2 per CU), each walking
        // over its share of the tiles so that the weight panel is staged once per workgroup
        const long long per_cu = std::max<long long>(1, std::min<long long>(2, (160 * 1024) / (long long)plds));
        const long long want = std::max<long long>(1, 256 * per_cu / gy);
        dim3 pgrid((unsigned)std::min<long long>(gx, want), gy);
        const int c4 = CH4 <= 4 ? 4 : (CH4 <= 8 ? 8 : 16);
        const int jj = J <= 2 ? 2 : (J <= 3 ? 3 : (J <= 5 ? 5 : 9));
        bool done = true;
        const bool have = !((jj == 5 && c4 > 8) || (jj == 9 && c4 > 4));
        // statistics epilogue: every pixel of every tile is an output pixel, and few enough partials
        const bool stats_ok = have && a.TW == gw && gh % a.TH == 0 && pgrid.x <= (unsigned)CONV_STATS_MAX_PARTS;
        if (plan && have) {
            *plan = ConvStatsPlan{stats_ok ? (int)pgrid.x : 0, (int)gx, a.TH * a.TW, 1, MT, NT, c4, jj, pitch_mul, tw_shift, pgrid.x, pgrid.y, k4n_rcp, plds};
        }
        if (dry && !(par && !have)) return have ? 1 : 0;
        AQ_REQUIRE(!(a.stat_part || a.bwd_part) || stats_ok, -ENOTSUP, "%s: no statistics epilogue for this shape", who);
#define AQ_P(MM, NN, CC, JJ)                                                                                                      \
    {                                                                                                                             \
        auto kern = conv_gemm_persist_kernel<MM, NN, CC, JJ, MODE>;                                                               \
        if (plds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds); \
        hipLaunchKernelGGL(kern, pgrid, dim3(256), plds, st, a, (int)gx, pitch_mul, tw_shift, k4n_rcp);                          \
    }
#define AQ_PJ(MM, NN, CC)                                 \
        if (jj == 2) AQ_P(MM, NN, CC, 2)                  \
        else if (jj == 3) AQ_P(MM, NN, CC, 3)             \
        else if (jj == 5 && CC <= 8) AQ_P(MM, NN, CC, 5)  \
        else if (jj == 9 && CC <= 4) AQ_P(MM, NN, CC, 9)  \
        else done = false;
#define AQ_PC(MM, NN)                          \
        if (c4 == 4) { AQ_PJ(MM, NN, 4) }      \
        else if (c4 == 8) { AQ_PJ(MM, NN, 8) } \
        else { AQ_PJ(MM, NN, 16) }
        if (MT == 4 && NT == 2) { AQ_PC(4, 2) }
        else if (MT == 4) { AQ_PC(4, 1) }
        else if (MT == 2 && NT == 2) { AQ_PC(2, 2) }
        else if (MT == 2) { AQ_PC(2, 1) }
        else if (NT == 2) { AQ_PC(1, 2) }
        else { AQ_PC(1, 1) }
#undef AQ_PC
#undef AQ_PJ
#undef AQ_P
        if (done) return check_launch(who);
    }
    if (par) {
        *fell = true;
        return 0;
    }
    if (dry) return 0;
    AQ_REQUIRE(!a.bn_on, -ENOTSUP, "%s: no kernel with fused input normalisation for this shape", who);
    AQ_REQUIRE(!a.stat_part && !a.bwd_part, -ENOTSUP, "%s: no statistics epilogue for this shape", who);
    if (MODE != MODE_FWD_CODES) {  // stride 1, at least 64 channels either side: the minimal-filtering form (2.25 times less matrix work)
        bool taken = false;
        const int rc = MODE == MODE_DGRAD ? aq_wino_launch_dgrad(a, N, st, who, &taken) : aq_wino_launch_fwd(a, N, st, who, &taken);
        if (taken) return rc;
    }
    {
        bool taken = false;
        const int rc = launch_c3<MODE>(a, N, st, who, &taken);
        if (taken) return rc;
    }
    if (lds > 64 * 1024) return launch_chunked<MODE>(a, N, st, who);  // too many channels for one LDS image: chunks
    dim3 grid((unsigned)gx, gy);
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

template <int MODE>
static int launch_gemm(ConvArgs &a, int N, hipStream_t st, const char *who, bool dry = false, ConvStatsPlan *plan = nullptr)
{
    if (a.ks != 3 || a.padT != 1) {  // anything but 3x3 / padding 1: the general kernel (no epilogues)
        if (plan) *plan = ConvStatsPlan{};
        if (dry) return 0;
        AQ_REQUIRE(!a.bn_on && !a.stat_part && !a.bwd_part, -ENOTSUP, "%s: no fused normalisation / statistics epilogue for this kernel size", who);
        bool taken = false;
        if (a.ks == 1 && MODE != MODE_FWD_CODES) {  // 1x1: a plain GEMM, nothing to stage
            const int rc1 = MODE == MODE_DGRAD ? aq_c1_launch_dgrad(a, N, st, who, &taken) : aq_c1_launch_fwd(a, N, st, who, &taken);
            if (taken) return rc1;
        }
        if (MODE == MODE_FWD) {  // the image stem (RGB, 7x7, stride 2)
            const int rcs = aq_stem_launch_fwd(a, N, st, who, &taken);
            if (taken) return rcs;
        }
        const int rc = launch_c3<MODE>(a, N, st, who, &taken);
        if (taken) return rc;
        return launch_chunked<MODE>(a, N, st, who);
    }
    // stride-2 dgrad: by parity classes when the class shares of the k axis are whole pairs of k-blocks
    if (MODE == MODE_DGRAD && a.dil == 2 && a.KC % 32 == 0 && !(a.OH & 1) && !(a.OW & 1) && a.IH * 2 == a.OH && a.IW * 2 == a.OW) {
        ConvArgs b = a;
        bool fell = false;
        const int rc = launch_gemm_impl<MODE>(b, N, st, who, dry, plan, true, &fell);
        if (!fell) {
            a = b;
            return rc;
        }
    }
    return launch_gemm_impl<MODE>(a, N, st, who, dry, plan, false, nullptr);
}

static void conv_fwd_args(ConvArgs &a, int N, int Cin, int H, int Wd, int Cout, int stride, int ks = 3, int pad = 1)
{
    a.N = N; a.Cin = Cin; a.Cout = Cout;
    a.ks = ks; a.ks2 = ks * ks; a.padT = pad;
    a.M = Cout; a.KC = Cin; a.K = Cin * ks * ks;
    a.OH = conv_out(H, ks, stride, pad); a.OW = conv_out(Wd, ks, stride, pad);
    a.IH = H; a.IW = Wd; a.rs = stride; a.dil = 1;
}

static int conv_fwd_common(const float *x, const float *W, const uint8_t *idx, const float *q, int L, const float *res, float *y, int N, int Cin,
                           int H, int Wd, int Cout, int ks, int stride, int pad, hipStream_t st, const char *who, float2 *stat_part = nullptr,
                           ConvStatsPlan *plan = nullptr)
{
    if (int rc = check_conv(who, N, Cin, H, Wd, Cout, ks, stride, pad)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(x && y && (W || (idx && q)), -EFAULT, "%s: NULL pointer", who);
    ConvArgs a = {};
    a.in = x; a.W = W; a.idx = idx; a.q = q; a.L = L; a.out = y; a.res = res;
    a.stat_part = stat_part;
    conv_fwd_args(a, N, Cin, H, Wd, Cout, stride, ks, pad);
    return idx ? launch_gemm<MODE_FWD_CODES>(a, N, st, who, false, plan) : launch_gemm<MODE_FWD>(a, N, st, who, false, plan);
}

} // namespace aq

extern "C" {


int aq_conv_fwd(const float *x, const float *W, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *stream)
{
    return aq::conv_fwd_common(x, W, nullptr, nullptr, 0, nullptr, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream), "aq_conv_fwd");
}

int aq_conv_fwd_add(const float *x, const float *W, const float *res, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride,
                    int pad, void *stream)
{
    AQ_REQUIRE(res || N == 0, -EFAULT, "aq_conv_fwd_add: NULL res");
    return aq::conv_fwd_common(x, W, nullptr, nullptr, 0, res, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream),
                               "aq_conv_fwd_add");
}

size_t aq_conv_stats_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    using namespace aq;
    if (check_conv("aq_conv_stats_bytes", N, Cin, H, Wd, Cout, ksize, stride, pad) || N == 0) return 0;
    ConvArgs a = {};
    conv_fwd_args(a, N, Cin, H, Wd, Cout, stride, ksize, pad);
    ConvStatsPlan plan;
    launch_gemm<MODE_FWD>(a, N, nullptr, "aq_conv_stats_bytes", true, &plan);
    return (size_t)plan.P * Cout * sizeof(float2);
}

int aq_conv_fwd_stats(const float *x, const float *W, const float *res, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride,
                      int pad, void *stats, size_t stats_bytes, aq_conv_stats *desc, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(desc, -EFAULT, "aq_conv_fwd_stats: NULL desc");
    *desc = aq_conv_stats{0, 0, 0};
    AQ_REQUIRE(N >= 1, -EINVAL, "aq_conv_fwd_stats: statistics of an empty batch");
    const size_t need = aq_conv_stats_bytes(N, Cin, H, Wd, Cout, ksize, stride, pad);
    AQ_REQUIRE(need > 0, -ENOTSUP, "aq_conv_fwd_stats: no statistics epilogue for this shape (aq_conv_stats_bytes is 0): run the separate statistics pass");
    AQ_REQUIRE(stats && stats_bytes >= need, -ENOSPC, "aq_conv_fwd_stats: statistics buffer of %zu bytes needed, %zu given", need, stats_bytes);
    ConvStatsPlan plan;
    if (int rc = conv_fwd_common(x, W, nullptr, nullptr, 0, res, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream),
                                 "aq_conv_fwd_stats", static_cast<float2 *>(stats), &plan))
        return rc;
    *desc = aq_conv_stats{plan.P, plan.ntiles, plan.tile_px};
    return 0;
}

int aq_conv_fwd_codes(const float *x, const uint8_t *idx, const float *q, int L, float *y, int N, int Cin, int H, int Wd, int Cout, int ksize,
                      int stride, int pad, void *stream)
{
    AQ_REQUIRE(L >= 1 && L <= AQ_MAX_LEVELS, -EINVAL, "aq_conv_fwd_codes: L=%d outside 1..%d", L, AQ_MAX_LEVELS);
    AQ_REQUIRE(idx && q, -EFAULT, "aq_conv_fwd_codes: NULL codes / levels");
    return aq::conv_fwd_common(x, nullptr, idx, q, L, nullptr, y, N, Cin, H, Wd, Cout, ksize, stride, pad, static_cast<hipStream_t>(stream), "aq_conv_fwd_codes");
}

static void conv_dgrad_args(aq::ConvArgs &a, int N, int Cin, int H, int Wd, int Cout, int stride, int ks = 3, int pad = 1)
{
    a.N = N; a.Cin = Cin; a.Cout = Cout;
    a.ks = ks; a.ks2 = ks * ks; a.padT = ks - 1 - pad;  // dx = full correlation of the (dilated) dy with the flipped kernel
    a.M = Cin; a.KC = Cout; a.K = Cout * ks * ks;
    a.OH = H; a.OW = Wd;
    a.IH = aq::conv_out(H, ks, stride, pad); a.IW = aq::conv_out(Wd, ks, stride, pad);
    a.rs = 1; a.dil = stride;
}

static int conv_dgrad_common(const char *who, const float *dy, const float *W, const float *res, float *dx, int N, int Cin, int H, int Wd, int Cout,
                             int ksize, int stride, int pad, void *stream)
{
    using namespace aq;
    if (int rc = check_conv(who, N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    if (N == 0) return 0;
    AQ_REQUIRE(dy && W && dx, -EFAULT, "%s: NULL pointer", who);
    ConvArgs a = {};
    a.in = dy; a.W = W; a.out = dx; a.res = res;
    conv_dgrad_args(a, N, Cin, H, Wd, Cout, stride, ksize, pad);
    return launch_gemm<MODE_DGRAD>(a, N, static_cast<hipStream_t>(stream), who);
}

int aq_conv_dgrad(const float *dy, const float *W, float *dx, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *stream)
{
    return conv_dgrad_common("aq_conv_dgrad", dy, W, nullptr, dx, N, Cin, H, Wd, Cout, ksize, stride, pad, stream);
}

int aq_conv_dgrad_add(const float *dy, const float *W, const float *res, float *dx, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride,
                      int pad, void *stream)
{
    AQ_REQUIRE(res || N == 0, -EFAULT, "aq_conv_dgrad_add: NULL res");
    return conv_dgrad_common("aq_conv_dgrad_add", dy, W, res, dx, N, Cin, H, Wd, Cout, ksize, stride, pad, stream);
}

size_t aq_conv_dgrad_bn_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    using namespace aq;
    if (check_conv("aq_conv_dgrad_bn_bytes", N, Cin, H, Wd, Cout, ksize, stride, pad) || N == 0) return 0;
    ConvArgs a = {};
    conv_dgrad_args(a, N, Cin, H, Wd, Cout, stride, ksize, pad);
    ConvStatsPlan plan;
    launch_gemm<MODE_DGRAD>(a, N, nullptr, "aq_conv_dgrad_bn_bytes", true, &plan);
    return (size_t)plan.P * Cin * sizeof(float2);
}

int aq_conv_dgrad_bn(const float *dy, const float *W, float *dz, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad,
                     const float *x, const float *gamma, const float *beta, const float *save_mean, const float *save_rstd, void *sums,
                     size_t sums_bytes, aq_conv_stats *desc, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(desc, -EFAULT, "aq_conv_dgrad_bn: NULL desc");
    *desc = aq_conv_stats{0, 0, 0};
    if (int rc = check_conv("aq_conv_dgrad_bn", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(N >= 1, -EINVAL, "aq_conv_dgrad_bn: empty batch");
    AQ_REQUIRE(dy && W && dz && x && gamma && beta && save_mean && save_rstd, -EFAULT, "aq_conv_dgrad_bn: NULL pointer");
    const size_t need = aq_conv_dgrad_bn_bytes(N, Cin, H, Wd, Cout, ksize, stride, pad);
    AQ_REQUIRE(need > 0, -ENOTSUP, "aq_conv_dgrad_bn: no reduction epilogue for this shape (aq_conv_dgrad_bn_bytes is 0): use aq_conv_dgrad and aq_bn_relu_bwd");
    AQ_REQUIRE(sums && sums_bytes >= need, -ENOSPC, "aq_conv_dgrad_bn: buffer of %zu bytes needed, %zu given", need, sums_bytes);
    ConvArgs a = {};
    a.in = dy; a.W = W; a.out = dz;
    a.bx = x; a.bgamma = gamma; a.bbeta = beta; a.bmean = save_mean; a.brstd = save_rstd;
    a.bwd_part = static_cast<float2 *>(sums);
    conv_dgrad_args(a, N, Cin, H, Wd, Cout, stride, ksize, pad);
    ConvStatsPlan plan;
    if (int rc = launch_gemm<MODE_DGRAD>(a, N, static_cast<hipStream_t>(stream), "aq_conv_dgrad_bn", false, &plan)) return rc;
    *desc = aq_conv_stats{plan.P, plan.ntiles, plan.tile_px};
    return 0;
}

// LDS bank conflicts of the wgrad operand reads: the 16 lanes jn of a wavefront read 16 consecutive
// (ci, tap) columns of the x image and the lane groups kq read neighbouring pixels, 32 lanes (kq pair)
// per LDS cycle over 32 banks.  Returns the number of extra LDS cycles summed over the column phases.
static int wgrad_bank_cost(int pitch, int plane, int stride, int K9)
{
    int cost = 0;
    for (int j0 = 0; j0 < K9 && j0 < 18 * 16; j0 += 16)
        for (int half = 0; half < 2; ++half) {
            int addr[32][32], n[32] = {0};
            for (int jn = 0; jn < 16; ++jn) {
                const int jcol = j0 + jn;
                if (jcol >= K9) continue;
                const int ci = jcol / 9, tap = jcol % 9;
                for (int kq = 2 * half; kq < 2 * half + 2; ++kq) {
                    const int ad = ci * plane + (tap / 3) * pitch + tap % 3 + kq * stride, b = ad & 31;
                    bool seen = false;
                    for (int i = 0; i < n[b]; ++i) seen = seen || addr[b][i] == ad;
                    if (!seen) addr[b][n[b]++] = ad;
                }
            }
            int worst = 1;
            for (int b = 0; b < 32; ++b) worst = n[b] > worst ? n[b] : worst;
            cost += worst - 1;
        }
    return cost;
}

// pitch (<= 65) and plane of the LDS x image, padded so that the operand reads spread over the banks
static void wgrad_pad(int stride, int K9, int rows_in, int &pitch, int &plane)
{
    // the search costs about a millisecond of host time: remembered per geometry (plans are made on every call)
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> memo;
    const auto key = std::make_tuple(stride, K9, rows_in, pitch);
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = memo.find(key);
        if (it != memo.end()) {
            pitch = it->second.first;
            plane = it->second.second;
            return;
        }
    }
    int best = 1 << 30, bp = pitch, bpl = rows_in * pitch;
    const int limit = ((rows_in * pitch + 63) / 64) * 64;  // same number of 64-element slices as unpadded
    for (int p = pitch; p <= pitch + 8 && p <= 65; ++p)
        for (int extra = 0; extra < 32; ++extra) {
            const int pl = rows_in * p + extra;
            if (pl > limit) continue;
            const int score = wgrad_bank_cost(p, pl, stride, K9) * 4096 + pl;  // conflicts first, then size
            if (score < best) { best = score; bp = p; bpl = pl; }
        }
    pitch = bp;
    plane = bpl;
    std::lock_guard<std::mutex> lock(mu);
    memo[key] = std::make_pair(bp, bpl);
}

static int wgrad_plan_with(aq::WgradArgs &a, int &MG, int &NG, dim3 &grid, size_t &lds, size_t &ws_bytes, int ng_cap, int npix_cap);

// Plan of the slab kernel.  The preferred tile group (as many (ci, tap) columns per workgroup as the registers hold)
// may need more LDS than a CU has once many input channels, a large kernel or stride 2 widen the staged image:
// then fewer columns per workgroup (fewer input channels staged), then smaller pixel tiles.
static int wgrad_plan(aq::WgradArgs &a, int &MG, int &NG, dim3 &grid, size_t &lds, size_t &ws_bytes)
{
    const int caps[4][2] = {{18, 128}, {9, 128}, {3, 128}, {3, 64}};
    int rc = 0;
    for (int i = 0; i < 4; ++i) {
        aq::WgradArgs t = a;
        rc = wgrad_plan_with(t, MG, NG, grid, lds, ws_bytes, caps[i][0], caps[i][1]);
        if (rc != -E2BIG || i == 3) {
            a = t;
            return rc;
        }
    }
    return rc;
}

static int wgrad_plan_with(aq::WgradArgs &a, int &MG, int &NG, dim3 &grid, size_t &lds, size_t &ws_bytes, int ng_cap, int npix_cap)
{
    using namespace aq;
    if (a.ks == 0) { a.ks = 3; a.pad = 1; }
    a.ks2 = a.ks * a.ks;
    a.Ho = conv_out(a.H, a.ks, a.stride, a.pad);
    a.Wo = conv_out(a.Wd, a.ks, a.stride, a.pad);
    a.K9 = a.Cin * a.ks2;
    const int mtiles = (a.Cout + 15) / 16, ntiles = (a.K9 + 15) / 16;
    // tile group kept in registers by every compute wavefront: MG x NG <= 18 tiles (72 accumulator
    // registers + two operand sets fit the 256-register budget of an 8-wavefront workgroup)
    if (ntiles <= 3 || ng_cap <= 3) NG = 3;
    else if (ntiles <= 9 || ng_cap <= 9) NG = 9;
    else NG = 18;
    MG = (NG == 9 && mtiles >= 2) ? 2 : 1;
    pick_tile(a.Wo, npix_cap, a.TW, a.TH);
    if ((long long)a.Ho * a.Wo <= 64) pick_tile(a.Wo, 64, a.TW, a.TH);
    a.tiles_x = (a.Wo + a.TW - 1) / a.TW;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    a.pitch = (a.TW - 1) * a.stride + a.ks;
    a.rows_in = (a.TH - 1) * a.stride + a.ks;
    a.plane = a.rows_in * a.pitch;
    {   // channels touched by the 16*NG consecutive (ci, tap) columns of one workgroup
        const int gzs = (ntiles + NG - 1) / NG;
        a.CW = 1;
        for (int z = 0; z < gzs; ++z) {
            const int j0 = z * 16 * NG, j1 = std::min(a.K9, j0 + 16 * NG) - 1;
            a.CW = std::max(a.CW, j1 / a.ks2 - j0 / a.ks2 + 1);
        }
    }
    if (a.ks == 3) {  // pad the LDS image against bank conflicts when that keeps the per-lane element count (the kernel
        // variants are instantiated per count) and the LDS budget
        int pp = a.pitch, pl = a.plane;
        wgrad_pad(a.stride, a.K9, a.rows_in, pp, pl);
        const size_t need = 2 * ((size_t)16 * MG * (a.TH * a.TW + 2) + (size_t)a.CW * pl) * 4;
        if ((pl + 63) / 64 == (a.plane + 63) / 64 && need <= 160 * 1024) { a.pitch = pp; a.plane = pl; }
    }
    a.cpitch = a.TH * a.TW + 2;
    a.tw_shift = a.npix_shift = 0;
    while ((1 << a.tw_shift) < a.TW) ++a.tw_shift;
    while ((1 << a.npix_shift) < a.TH * a.TW) ++a.npix_shift;
    a.pitch_mul = (65536 + a.pitch - 1) / a.pitch;
    const int gy = (mtiles + MG - 1) / MG, gz = (ntiles + NG - 1) / NG;
    // images per workgroup: keep >= ~256 workgroups in flight, at most 8 images each
    long long chunks = (long long)a.N * a.tiles_x * a.tiles_y;
    int ipb = 1;
    while (ipb < 8 && chunks * gy * gz / (ipb * 2) >= 256 && ipb * 2 <= a.N) ipb *= 2;  // >= 256 workgroups
    a.IPB = ipb;
    a.ngroups_img = (a.N + ipb - 1) / ipb;
    const long long gx = (long long)a.ngroups_img * a.tiles_x * a.tiles_y;
    AQ_REQUIRE(gx < (1LL << 31) && gy < 65536 && gz < 65536, -E2BIG, "aq_conv_wgrad: grid too large");
    grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)gz);
    lds = 2 * ((size_t)16 * MG * a.cpitch + (size_t)a.CW * a.plane) * 4;  // two chunk buffers
    lds = std::max(lds, (size_t)8 * 4 * 256 * 4);
    AQ_REQUIRE(lds <= 160 * 1024, -E2BIG, "aq_conv_wgrad: tiles need %zu bytes of LDS (limit 163840): too many channels", lds);
    ws_bytes = (size_t)gx * a.Cout * a.K9 * 4;
    return 0;
}

static int wgrad_variant(const aq::WgradArgs &a, int MG, int NG);

// Plan of conv_wgrad_big_kernel, used for layers with at least 64 output channels and 144 (ci, tap) columns that
// have no tuned variant of the small-channel kernel (the reference's ResNet-20 shapes keep theirs).
struct WgradBigPlan {
    bool use;
    int nchunks, nslab;
    dim3 grid;
    size_t lds, ws_bytes;
};
static WgradBigPlan wgrad_big_plan(aq::WgradArgs &a)
{
    using namespace aq;
    WgradBigPlan p = {};
    if (a.ks == 0) { a.ks = 3; a.pad = 1; }
    a.ks2 = a.ks * a.ks;
    a.Ho = conv_out(a.H, a.ks, a.stride, a.pad);
    a.Wo = conv_out(a.Wd, a.ks, a.stride, a.pad);
    a.K9 = a.Cin * a.ks2;
    if (a.Cout < 64 || a.K9 < 144) return p;
    {
        WgradArgs t = a;
        int MGs, NGs;
        dim3 g;
        size_t l, w;
        if (wgrad_plan(t, MGs, NGs, g, l, w) == 0 && wgrad_variant(t, MGs, NGs) != 0) return p;
    }
    constexpr int NG = 9;
    pick_tile(a.Wo, (long long)a.Ho * a.Wo <= 64 ? 64 : 128, a.TW, a.TH);
    a.tiles_x = (a.Wo + a.TW - 1) / a.TW;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    a.pitch = (a.TW - 1) * a.stride + a.ks;
    a.rows_in = (a.TH - 1) * a.stride + a.ks;
    a.plane = a.rows_in * a.pitch;
    a.cpitch = a.TH * a.TW + 4;  // row pitch = 4 mod 64 banks: the 16 rows x 4 pixels of an operand read hit 64 banks
    a.tw_shift = a.npix_shift = 0;
    while ((1 << a.tw_shift) < a.TW) ++a.tw_shift;
    while ((1 << a.npix_shift) < a.TH * a.TW) ++a.npix_shift;
    const int ncol_tiles = (a.K9 + 15) / 16, gz = (ncol_tiles + NG - 1) / NG;
    a.CW = 1;
    for (int z = 0; z < gz; ++z) {
        const int j0 = z * 16 * NG, j1 = std::min(a.K9, j0 + 16 * NG) - 1;
        a.CW = std::max(a.CW, j1 / a.ks2 - j0 / a.ks2 + 1);
    }
    p.lds = ((size_t)64 * a.cpitch + (size_t)a.CW * a.plane) * 4;
    if (p.lds > 64 * 1024) return p;
    const int gy = (a.Cout + 63) / 64;
    p.nchunks = a.N * a.tiles_x * a.tiles_y;
    // slabs: enough workgroups to fill the chip a few times over, at least ~4 chunks each, at most 256 slabs
    long long want = (1024 + (long long)gy * gz - 1) / ((long long)gy * gz);
    want = std::max<long long>(1, std::min<long long>(std::min<long long>(want, 256), (p.nchunks + 3) / 4));
    p.nslab = (int)std::min<long long>(want, p.nchunks);
    if (gy >= 65536 || gz >= 65536) return p;
    p.grid = dim3((unsigned)p.nslab, (unsigned)gy, (unsigned)gz);
    p.ws_bytes = (size_t)p.nslab * a.Cout * a.K9 * 4;
    p.use = true;
    return p;
}

size_t aq_conv_wgrad_workspace_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    if (aq::check_conv("aq_conv_wgrad_workspace_bytes", N, Cin, H, Wd, Cout, ksize, stride, pad) || N == 0) return 0;
    aq::WgradArgs a = {};
    a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Cout = Cout; a.stride = stride; a.ks = ksize; a.pad = pad;
    // the largest over the kernels that may take the shape (the variant that normalises x while staging it runs on the
    // small-channel kernel whatever the plain weight gradient runs on)
    size_t need = 0;
    {
        const WgC16Plan cp = wgrad_c16_plan(N, Cin, H, Wd, Cout, ksize, stride, pad);
        if (cp.use) need = std::max(need, cp.ws_bytes);
        const WgStemPlan sp = wgrad_stem_plan(nullptr, N, Cin, H, Wd, Cout, ksize, stride, pad);
        if (sp.use) need = std::max(need, sp.ws_bytes);
        aq::WgradArgs b = a;
        const WgradBigPlan bp = wgrad_big_plan(b);
        if (bp.use) need = std::max(need, bp.ws_bytes);
    }
    int MG, NG;
    dim3 grid;
    size_t lds, ws;
    if (wgrad_plan(a, MG, NG, grid, lds, ws) == 0) need = std::max(need, ws);
    return need;
}

// register-staged loader variant (0: generic) for a planned launch: the tile must span the output width
// and the image must fit the staging budget
static int wgrad_variant(const aq::WgradArgs &a, int MG, int NG)
{
    const int c4 = (a.CW + 3) / 4, jx = (a.plane + 63) / 64;
    const bool fits32 = (long long)a.Cin * a.H * a.Wd * 4 < (1LL << 31) && (long long)a.Cout * a.Ho * a.Wo * 4 < (1LL << 31);  // buffer offsets
    if (a.tiles_x != 1 || !fits32 || a.ks != 3 || a.pad != 1) return 0;
    if (NG == 3 && MG == 1 && c4 <= 1 && jx <= 4) return 6;  // RGB stem
    if (NG == 9 && MG == 1 && c4 <= 4 && jx <= 4) return 1;
    if (NG == 9 && MG == 2 && c4 <= 4 && jx <= 9) return 2;
    if (NG == 18 && c4 <= 8 && jx <= 2) return 5;  // 18 n-tiles = 288 columns = at most 32 channels
    if (NG == 18 && c4 <= 8 && jx <= 3) return 3;
    if (NG == 18 && c4 <= 8 && jx <= 5) return 4;
    return 0;
}

struct BnOperand {  // statistics of the batch normalisation applied to x while it is staged (all NULL: none)
    const float *gamma, *beta, *mean, *rstd;
};

// slab kernel of aq_conv_wgrad; *nslab_out = number of slabs written to workspace ([nslab][Cout*Cin*9])
static int wgrad_slabs(const float *x, const float *dy, int N, int Cin, int H, int Wd, int Cout, int stride, void *workspace,
                       size_t workspace_bytes, hipStream_t st, int *nslab_out, const BnOperand *bn = nullptr, int ks = 3, int pad = 1,
                       const aq::WgradGroup *grp = nullptr);

int aq_conv_wgrad(const float *x, const float *dy, float *dW, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_wgrad", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(dW, -EFAULT, "aq_conv_wgrad: NULL dW");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nW = Cout * Cin * ksize * ksize;
    int nslab = 0;
    if (N > 0)
        if (int rc = wgrad_slabs(x, dy, N, Cin, H, Wd, Cout, stride, workspace, workspace_bytes, st, &nslab, nullptr, ksize, pad)) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(static_cast<const float *>(workspace), dW, nW)), dim3(256), 0, st,
                       static_cast<const float *>(workspace), dW, nW, nslab);
    return check_launch("aq_conv_wgrad(reduce)");
}

int aq_conv_wgrad_partial(const float *x, const float *dy, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad,
                          void *workspace, size_t workspace_bytes, int *nslab, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_wgrad_partial", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(nslab, -EFAULT, "aq_conv_wgrad_partial: NULL nslab");
    *nslab = 0;
    if (N == 0) return 0;
    return wgrad_slabs(x, dy, N, Cin, H, Wd, Cout, stride, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nslab, nullptr, ksize, pad);
}

int aq_conv_wgrad_partial_group(const float *const *x, const float *const *dy, void *const *workspace, int nlayers, int N, int Cin, int H, int Wd,
                                int Cout, int ksize, int stride, int pad, size_t workspace_bytes, int *nslab, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_wgrad_partial_group", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(nslab, -EFAULT, "aq_conv_wgrad_partial_group: NULL nslab");
    *nslab = 0;
    AQ_REQUIRE(nlayers >= 0 && (nlayers == 0 || (x && dy && workspace)), -EINVAL, "aq_conv_wgrad_partial_group: bad layer list");
    if (N == 0 || nlayers == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < nlayers; base += WGRAD_GROUP_MAX) {
        WgradGroup g = {};
        g.n = std::min(WGRAD_GROUP_MAX, nlayers - base);
        g.total = nlayers;
        for (int l = 0; l < g.n; ++l) {
            g.x[l] = x[base + l]; g.dy[l] = dy[base + l]; g.part[l] = static_cast<float *>(workspace[base + l]);
        }
        int rc = wgrad_slabs(nullptr, nullptr, N, Cin, H, Wd, Cout, stride, nullptr, workspace_bytes, st, nslab, nullptr, ksize, pad, &g);
        if (rc == -ENOTSUP) {  // a shape of the many-channel kernels: one launch per layer
            for (int l = 0; l < g.n; ++l)
                if (int r2 = wgrad_slabs(g.x[l], g.dy[l], N, Cin, H, Wd, Cout, stride, g.part[l], workspace_bytes, st, nslab, nullptr, ksize, pad)) return r2;
            rc = 0;
        }
        if (rc) return rc;
    }
    return 0;
}

int aq_conv_wgrad_reduce(const aq_wgrad_job *jobs, int njobs, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(njobs >= 0 && (njobs == 0 || jobs), -EINVAL, "aq_conv_wgrad_reduce: bad job list");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < njobs; base += WGRAD_MAX_JOBS) {
        WgradJobTab t = {};
        t.njobs = std::min(WGRAD_MAX_JOBS, njobs - base);
        int blocks = 0;
        for (int j = 0; j < t.njobs; ++j) {
            const aq_wgrad_job &q = jobs[base + j];
            AQ_REQUIRE(q.dW && q.n >= 1 && q.nslab >= 0 && (q.nslab == 0 || q.part), -EINVAL, "aq_conv_wgrad_reduce: bad job %d", base + j);
            t.part[j] = q.part; t.dW[j] = q.dW; t.n[j] = q.n; t.nslab[j] = q.nslab;
            t.first[j] = blocks;
            blocks += wgrad_reduce_blocks(q.part, q.dW, q.n);
        }
        t.first[t.njobs] = blocks;
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, st, t);
        if (int rc = check_launch("aq_conv_wgrad_reduce")) return rc;
    }
    return 0;
}

int aq_bnconv_supported(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    using namespace aq;
    if (check_conv("aq_bnconv_supported", N, Cin, H, Wd, Cout, ksize, stride, pad) || N < 1 || ksize != 3 || pad != 1) return 0;
    ConvArgs a = {};
    conv_fwd_args(a, N, Cin, H, Wd, Cout, stride);
    if (launch_gemm<MODE_FWD>(a, N, nullptr, "aq_bnconv_supported", true) != 1) return 0;
    WgradArgs w = {};
    w.N = N; w.Cin = Cin; w.H = H; w.Wd = Wd; w.Cout = Cout; w.stride = stride;
    int MG, NG;
    dim3 grid;
    size_t lds, ws;
    if (wgrad_plan(w, MG, NG, grid, lds, ws)) return 0;
    return wgrad_variant(w, MG, NG) != 0 && (long long)N * H * Wd >= 2;
}

int aq_bnconv_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                  long long *num_batches_tracked, float eps, float momentum, const float *W, const float *res, float *y, float *save_mean,
                  float *save_rstd, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *workspace,
                  size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_bnconv_fwd", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(N >= 1 && (long long)N * H * Wd >= 2, -EINVAL, "aq_bnconv_fwd: batch statistics need more than one value per channel");
    AQ_REQUIRE(x && gamma && beta && W && y && save_mean && save_rstd, -EFAULT, "aq_bnconv_fwd: NULL pointer");
    AQ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), -EINVAL, "aq_bnconv_fwd: running_mean and running_var go together");
    AQ_REQUIRE(eps >= 0.0f, -EINVAL, "aq_bnconv_fwd: negative eps");
    const BnGeom g = bn_geom(N, Cin, H * Wd);
    const size_t need = (size_t)Cin * g.S * sizeof(float2);
    AQ_REQUIRE(workspace && workspace_bytes >= need, -ENOSPC, "aq_bnconv_fwd: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    AQ_REQUIRE(aq_bnconv_supported(N, Cin, H, Wd, Cout, ksize, stride, pad), -ENOTSUP,
               "aq_bnconv_fwd: no kernel with fused input normalisation for this shape (see aq_bnconv_supported)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = launch_bn_stats(x, static_cast<float2 *>(workspace), g, st)) return rc;
    ConvArgs a = {};
    a.in = x; a.W = W; a.out = y; a.res = res;
    conv_fwd_args(a, N, Cin, H, Wd, Cout, stride);
    a.bn_on = 1;
    a.bn = BnStatsIn{static_cast<const float2 *>(workspace), gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd,
                     eps, momentum};
    a.bng = g;
    return launch_gemm<MODE_FWD>(a, N, st, "aq_bnconv_fwd");
}

int aq_bnconv_wgrad_partial(const float *x, const float *gamma, const float *beta, const float *save_mean, const float *save_rstd,
                            const float *dy, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *workspace,
                            size_t workspace_bytes, int *nslab, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_bnconv_wgrad_partial", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(nslab && gamma && beta && save_mean && save_rstd, -EFAULT, "aq_bnconv_wgrad_partial: NULL pointer");
    *nslab = 0;
    if (N == 0) return 0;
    const BnOperand bn{gamma, beta, save_mean, save_rstd};
    return wgrad_slabs(x, dy, N, Cin, H, Wd, Cout, stride, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nslab, &bn);
}

// grp: several layers of this shape at once (x, dy, workspace then come from it, each workspace of workspace_bytes);
// shapes that run on the many-channel kernels return -ENOTSUP for a group (the caller launches them one by one)
static int wgrad_slabs(const float *x, const float *dy, int N, int Cin, int H, int Wd, int Cout, int stride, void *workspace,
                       size_t workspace_bytes, hipStream_t st, int *nslab_out, const BnOperand *bn, int ks, int pad, const aq::WgradGroup *grp)
{
    using namespace aq;
    if (grp) {
        x = grp->x[0]; dy = grp->dy[0]; workspace = grp->part[0];
    }
    AQ_REQUIRE(x && dy, -EFAULT, "aq_conv_wgrad: NULL pointer");
    WgradArgs a = {};
    a.x = x; a.dy = dy; a.part = static_cast<float *>(workspace);
    if (bn) { a.bn_gamma = bn->gamma; a.bn_beta = bn->beta; a.bn_mean = bn->mean; a.bn_rstd = bn->rstd; }
    a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Cout = Cout; a.stride = stride; a.ks = ks; a.pad = pad;
    if (!bn) {
        const WgStemPlan sp = wgrad_stem_plan(dy, N, Cin, H, Wd, Cout, ks, stride, pad);
        if (sp.use && grp) return -ENOTSUP;
        if (sp.use) {
            AQ_REQUIRE(workspace && workspace_bytes >= sp.ws_bytes, -ENOSPC, "aq_conv_wgrad: workspace too small: need %zu bytes, got %zu", sp.ws_bytes,
                       workspace_bytes);
            if (int rc = wgrad_stem_launch(x, dy, static_cast<float *>(workspace), N, Cin, H, Wd, Cout, ks, stride, pad, st)) return rc;
            *nslab_out = sp.nslab;
            return check_launch("aq_conv_wgrad");
        }
        WgC16Plan cp = wgrad_c16_plan(N, Cin, H, Wd, Cout, ks, stride, pad, grp ? std::max(grp->n, grp->total) : 1);
        if (cp.use) {
            AQ_REQUIRE(workspace && workspace_bytes >= cp.ws_bytes, -ENOSPC, "aq_conv_wgrad: workspace too small: need %zu bytes, got %zu", cp.ws_bytes,
                       workspace_bytes);
            WgradGroup g1 = {};
            if (grp) {
                g1 = *grp;
            } else {
                g1.n = 1; g1.x[0] = x; g1.dy[0] = dy; g1.part[0] = static_cast<float *>(workspace);
            }
            for (int l = 0; l < g1.n; ++l) AQ_REQUIRE(g1.x[l] && g1.dy[l] && g1.part[l], -EFAULT, "aq_conv_wgrad: NULL pointer (layer %d)", l);
            AQ_REQUIRE((long long)cp.grid.x * g1.n < (1LL << 31), -E2BIG, "aq_conv_wgrad: grid too large");
            if (int rc = wgrad_c16_launch(cp, g1, ks, stride, st)) return rc;
            *nslab_out = cp.a.nslab;
            return check_launch("aq_conv_wgrad");
        }
        WgradArgs b = a;
        const WgradBigPlan bp = wgrad_big_plan(b);
        if (bp.use && grp) return -ENOTSUP;
        if (bp.use) {
            AQ_REQUIRE(workspace && workspace_bytes >= bp.ws_bytes, -ENOSPC, "aq_conv_wgrad: workspace too small: need %zu bytes, got %zu", bp.ws_bytes,
                       workspace_bytes);
            hipLaunchKernelGGL(conv_wgrad_big_kernel<9>, bp.grid, dim3(256), bp.lds, st, b, bp.nchunks);
            *nslab_out = bp.nslab;
            return check_launch("aq_conv_wgrad");
        }
    }
    int MG, NG;
    dim3 grid;
    size_t lds, ws;
    if (int rc = wgrad_plan(a, MG, NG, grid, lds, ws)) return rc;
    AQ_REQUIRE(workspace && workspace_bytes >= ws, -ENOSPC, "aq_conv_wgrad: workspace too small: need %zu bytes, got %zu", ws, workspace_bytes);
    const int variant = wgrad_variant(a, MG, NG);
    AQ_REQUIRE(variant != 0 || !a.bn_mean, -ENOTSUP, "aq_conv_wgrad: no kernel with fused input normalisation for this shape");
    WgradGroup gg = {};
    dim3 ggrid = grid;
    if (grp) {
        for (int l = 0; l < grp->n; ++l) AQ_REQUIRE(grp->x[l] && grp->dy[l] && grp->part[l], -EFAULT, "aq_conv_wgrad_partial_group: NULL pointer (layer %d)", l);
        AQ_REQUIRE((long long)grid.x * grp->n < (1LL << 31), -E2BIG, "aq_conv_wgrad_partial_group: grid too large");
        gg = *grp;
        gg.gx = (int)grid.x;
        ggrid.x = grid.x * (unsigned)grp->n;
    }
#define AQ_WGV(MM, NN, CC, JJ)                                                                                                 \
    if (grp) {                                                                                                                 \
        auto kern = conv_wgrad_group_kernel<MM, NN, CC, JJ>;                                                                   \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, ggrid, dim3(512), lds, st, a, gg);                                                            \
    } else {                                                                                                                   \
        auto kern = conv_wgrad_kernel<MM, NN, CC, JJ>;                                                                         \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                                                                 \
    }
#define AQ_WG(MM, NN) AQ_WGV(MM, NN, 0, 1)
    if (variant == 1) AQ_WGV(1, 9, 4, 4)
    else if (variant == 2) AQ_WGV(2, 9, 4, 9)
    else if (variant == 3) AQ_WGV(1, 18, 8, 3)
    else if (variant == 4) AQ_WGV(1, 18, 8, 5)
    else if (variant == 5) AQ_WGV(1, 18, 8, 2)
    else if (variant == 6) AQ_WGV(1, 3, 1, 4)
    else if (NG == 3) AQ_WG(1, 3)
    else if (NG == 9 && MG == 1) AQ_WG(1, 9)
    else if (NG == 9) AQ_WG(2, 9)
    else AQ_WG(1, 18)
#undef AQ_WGV
#undef AQ_WG
    *nslab_out = (int)grid.x;
    return check_launch("aq_conv_wgrad");
}

size_t aq_conv_bwd_sums_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad)
{
    return aq_conv_dgrad_bn_bytes(N, Cin, H, Wd, Cout, ksize, stride, pad);
}

int aq_conv_bwd(const float *dy, const float *W, const float *x, float *dx, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad,
                const float *xb, const float *gamma, const float *beta, const float *save_mean, const float *save_rstd, void *sums,
                size_t sums_bytes, aq_conv_stats *desc, void *workspace, size_t workspace_bytes, int *nslab, void *stream)
{
    using namespace aq;
    if (int rc = check_conv("aq_conv_bwd", N, Cin, H, Wd, Cout, ksize, stride, pad)) return rc;
    AQ_REQUIRE(nslab, -EFAULT, "aq_conv_bwd: NULL nslab");
    *nslab = 0;
    if (desc) *desc = aq_conv_stats{0, 0, 0};
    if (N == 0) return 0;
    AQ_REQUIRE(dy && W && x && dx, -EFAULT, "aq_conv_bwd: NULL pointer");
    const bool bn = xb != nullptr;
    AQ_REQUIRE(!bn || (gamma && beta && save_mean && save_rstd && sums && desc), -EFAULT, "aq_conv_bwd: the batch-norm epilogue needs all of its operands");
    hipStream_t st = static_cast<hipStream_t>(stream);

    // plans of the two halves
    ConvArgs a = {};
    a.in = dy; a.W = W; a.out = dx;
    conv_dgrad_args(a, N, Cin, H, Wd, Cout, stride, ksize, pad);
    ConvStatsPlan plan;
    launch_gemm<MODE_DGRAD>(a, N, nullptr, "aq_conv_bwd", true, &plan);
    WgradArgs w = {};
    w.x = x; w.dy = dy; w.part = static_cast<float *>(workspace);
    w.N = N; w.Cin = Cin; w.H = H; w.Wd = Wd; w.Cout = Cout; w.stride = stride; w.ks = ksize; w.pad = pad;
    int MG, NG;
    dim3 wgrid;
    size_t wlds, ws;
    if (int rc = wgrad_plan(w, MG, NG, wgrid, wlds, ws)) return rc;
    const int variant = wgrad_variant(w, MG, NG);
    const size_t lds = std::max(plan.lds, wlds);
    // one launch when a kernel pair is instantiated for the shape and two workgroups fit on a CU side by side
    int pair = 0;
    if (plan.persist && (!bn || plan.P > 0) && 2 * lds <= 160 * 1024) {
        if (plan.MT == 1 && plan.NT == 2 && plan.c4 == 4 && plan.jj == 5 && variant == 1) pair = 1;
        else if (plan.MT == 2 && plan.NT == 1 && plan.c4 == 8 && plan.jj == 2 && variant == 3) pair = 2;
        else if (plan.MT == 1 && plan.NT == 1 && plan.c4 == 16 && plan.jj == 2 && variant == 5) pair = 3;
    }
    if (!pair) {
        if (int rc = aq_conv_wgrad_partial(x, dy, N, Cin, H, Wd, Cout, ksize, stride, pad, workspace, workspace_bytes, nslab, stream)) return rc;
        if (bn && aq_conv_dgrad_bn_bytes(N, Cin, H, Wd, Cout, ksize, stride, pad) > 0)
            return aq_conv_dgrad_bn(dy, W, dx, N, Cin, H, Wd, Cout, ksize, stride, pad, xb, gamma, beta, save_mean, save_rstd, sums, sums_bytes, desc,
                                    stream);
        return aq_conv_dgrad(dy, W, dx, N, Cin, H, Wd, Cout, ksize, stride, pad, stream);  // desc->P stays 0: no sums
    }
    AQ_REQUIRE(workspace && workspace_bytes >= ws, -ENOSPC, "aq_conv_bwd: workspace too small: need %zu bytes, got %zu", ws, workspace_bytes);
    // one input-gradient workgroup per CU (a launch of its own uses two where LDS allows): the weight-gradient
    // workgroup beside it is the second one
    const int d_gx = std::min<int>((int)plan.grid_x, std::max(1, 256 / (int)plan.grid_y));
    if (bn) {
        const size_t need = (size_t)d_gx * Cin * sizeof(float2);
        AQ_REQUIRE(sums_bytes >= need, -ENOSPC, "aq_conv_bwd: buffer of %zu bytes needed for the sums, %zu given", need, sums_bytes);
        a.bx = xb; a.bgamma = gamma; a.bbeta = beta; a.bmean = save_mean; a.brstd = save_rstd;
        a.bwd_part = static_cast<float2 *>(sums);
    }
    const FusedGrid fg{d_gx, (int)plan.grid_y, (int)wgrid.x, (int)wgrid.y, (int)wgrid.z};
    const long long blocks = (long long)fg.d_gx * fg.d_gy + (long long)fg.w_gx * fg.w_gy * fg.w_gz;
    AQ_REQUIRE(blocks < (1LL << 31), -E2BIG, "aq_conv_bwd: grid too large");
#define AQ_FUSED(MT, NT, C4, JJ, MGG, NGG, WC, WJ)                                                                                   \
    {                                                                                                                                \
        auto kern = conv_bwd_fused_kernel<MT, NT, C4, JJ, MGG, NGG, WC, WJ>;                                                         \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, a, plan.ntiles, plan.pitch_mul, plan.tw_shift, plan.k4n_rcp, w, fg); \
    }
    if (pair == 1) AQ_FUSED(1, 2, 4, 5, 1, 9, 4, 4)
    else if (pair == 2) AQ_FUSED(2, 1, 8, 2, 1, 18, 8, 3)
    else AQ_FUSED(1, 1, 16, 2, 1, 18, 8, 2)
#undef AQ_FUSED
    *nslab = (int)wgrid.x;
    if (bn) *desc = aq_conv_stats{d_gx, plan.ntiles, plan.tile_px};
    return check_launch("aq_conv_bwd");
}

} // extern "C"
