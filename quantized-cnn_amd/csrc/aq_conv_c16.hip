// The many-channel convolution kernels: 3x3 forward / input gradient and 3x3 / 1x1 weight gradient for layers whose
// channel counts are multiples of 16 (adp_qtz_fan.py:574-601 is written for any channel count; the reference's own
// ResNet-20 layers keep their tuned kernels in aq_conv.hip for forward / input gradient and use the weight-gradient
// kernel of this file).  fp32 matrix cores (v_mfma_f32_16x16x4_f32), NCHW, no atomics: bit-reproducible.
#include "aq_conv_shared.h"

#include <algorithm>

namespace aq {

// 3x3 convolutions with many channels: input channels a multiple of 16, 48 or more output channels; forward with
// stride 1 | 2 and the stride-1 input gradient (MODE_DGRAD, the flipped kernel over dy).  Built for the matrix pipes:
//  * k order inside a chunk of 16 input channels is (tap, channel): one k-block of 16 = one tap x 16 channels, and the four
//    MFMAs of a k-block take channels 4*kq + t, t = 0..3, from lane group kq.
//  * LDS image of the chunk: [pixel][channel], C3_CS = 20 floats per pixel (16 + 4 of padding), so the B fragments of the
//    four MFMAs of a k-block are ONE 16-byte LDS read per lane, bank-conflict free over 16 neighbouring pixels
//    (20 * x mod 64 are 16 distinct multiples of 4).  Two images: chunk c+1 is fetched into registers before the
//    matrix work of chunk c and written to the other image after it; one barrier per chunk.
//  * A fragments never touch LDS: for its output channel a lane needs W[row][4*kq + t][tap], t < 4, tap < 9, which is
//    a contiguous run of 36 floats of the weight tensor (forward; 16-byte aligned), kept in registers for the whole
//    chunk and fetched one chunk ahead.  The four wavefronts of a workgroup take 16 output channels each and share the
//    image, so a workgroup covers 64 channels x 16*NT pixels.
constexpr int C3_CS = 20;
constexpr int C3_MAXIT = 6;  // staging items (pixel, group of 4 channels) per thread: 4 * plane <= 256 * C3_MAXIT

template <int NT, int MODE>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int m0 = blockIdx.y * 64 + wave * 16;
    const int bufsz = a.plane * C3_CS;
    const int chs = a.IH * a.IW;
    const float *img = a.in + (long long)n * a.KC * chs;

    // staging: item i = (group of 4 channels cg, plane element e), cg-major so that neighbouring lanes read neighbouring
    // pixels of one channel; goff = offset inside the chunk's 16 channel planes (-1: padding, -2: no item)
    int goff[C3_MAXIT], loff[C3_MAXIT];
    {
        const int r0 = oy0 * a.rs - a.padT, c0 = ox0 * a.rs - a.padT;
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it) {
            const int i = tid + 256 * it;
            goff[it] = -2;
            loff[it] = 0;
            if (i < 4 * a.plane) {
                const int cg = i / a.plane, e = i - cg * a.plane;
                const int r = e / a.pitch, cc = e - r * a.pitch;
                const int gy = r0 + r, gx = c0 + cc;
                const bool ok = gy >= 0 && gx >= 0 && gy < a.IH && gx < a.IW;
                goff[it] = ok ? 4 * cg * chs + gy * a.IW + gx : -1;
                loff[it] = e * C3_CS + 4 * cg;
            }
        }
    }
    auto stage_load = [&](int cbase, f4 (&xr)[C3_MAXIT]) {
        const float *src = img + (long long)cbase * chs;
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it) {
            xr[it] = f4{0.f, 0.f, 0.f, 0.f};
            if (goff[it] >= 0) {
                const float *s0 = src + goff[it];
                xr[it] = f4{s0[0], s0[chs], s0[2 * chs], s0[3 * chs]};
            }
        }
    };
    auto stage_store = [&](float *buf, const f4 (&xr)[C3_MAXIT]) {
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it)
            if (goff[it] != -2) *reinterpret_cast<f4 *>(buf + loff[it]) = xr[it];
    };

    // A: 36 floats per lane and chunk.  Forward: W[row][cbase + 4*kq ..+3][0..8] is one aligned run; element (t, tap)
    // sits at 9*t + tap.  Input gradient: row = input channel, W[cbase + 4*kq + t][row][0..8], four runs of 9, used
    // with the tap flipped.
    const int row = m0 + jn;
    const bool rok = row < a.M;
    auto load_a = [&](int cbase, float (&A)[36]) {
        if (MODE == MODE_DGRAD) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *w = a.W + ((long long)(cbase + 4 * kq + t) * a.Cin + (rok ? row : 0)) * 9;
                __builtin_memcpy(&A[9 * t], w, 16);
                __builtin_memcpy(&A[9 * t + 4], w + 4, 16);
                A[9 * t + 8] = w[8];
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

    int pb[NT], oyx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pr = 16 * t + jn;  // (a tile of TH x TW pixels may leave the last of the 16 * NT slots empty)
        const bool used = pr < a.TW * a.TH;
        const int py = used ? pr / a.TW : 0, px = used ? pr - py * a.TW : 0;
        pb[t] = (py * a.rs * a.pitch + px * a.rs) * C3_CS + 4 * kq;
        const int oy = oy0 + py, ox = ox0 + px;
        oyx[t] = (used && oy < a.OH && ox < a.OW) ? ((oy << 16) | ox) : -1;
    }
    auto tap_off = [&](int i) { return ((i / 3) * a.pitch + i % 3) * C3_CS; };  // LDS offset of tap i

    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = a.KC >> 4;
    float A[36], An[36];
    f4 xr[C3_MAXIT];
    stage_load(0, xr);
    load_a(0, A);
    stage_store(smem, xr);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) {
            stage_load(16 * (c + 1), xr);
            load_a(16 * (c + 1), An);
        }
        const float *buf = smem + (c & 1) * bufsz;
        f4 b[NT], bn[NT];
        {
            const int toff = tap_off(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f4 *>(buf + pb[nt] + toff);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                const int toff = tap_off(tap + 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bn[nt] = *reinterpret_cast<const f4 *>(buf + pb[nt] + toff);
            }
            // index of the tap's weight inside a lane's run of 9
            const int ta = MODE == MODE_DGRAD ? 8 - tap : tap;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[9 * t + ta], b[nt][t], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
        }
        if (more) stage_store(smem + ((c + 1) & 1) * bufsz, xr);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 36; ++i) A[i] = An[i];
    }

    // D[row = 4*kq + r][col = jn]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (oyx[nt] < 0) continue;
        const int oy = oyx[nt] >> 16, ox = oyx[nt] & 0xffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = m0 + 4 * kq + r;
            if (ch < a.M) {
                const long long o = (((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox;
                a.out[o] = a.res ? acc[nt][r] + a.res[o] : acc[nt][r];
            }
        }
    }
}

// Stride-2 input gradient, all four output parity classes from one workgroup: the tile is TH x TW positions (a, b) of the
// class grid, the LDS image holds the undilated dy around them ((TH + 1) x (TW + 1) pixels), and every tap (ty, tx) of the
// nine feeds the class (cy, cx) = ((ty + pad) & 1, (tx + pad) & 1) of dx[2a + cy][2b + cx], reading dy at row
// a + (cy + pad - ty) / 2.  Nine taps per staged chunk as in the forward kernel (the one-class form above stages a chunk
// for one to four taps), at the price of four accumulator sets: NT <= 4.  PP = pad & 1.
template <int NT, int PP>
__global__ __launch_bounds__(256) void conv3x3_c16_dgrad2_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;

    int bx = blockIdx.x;
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int m0 = blockIdx.y * 64 + wave * 16;
    const int bufsz = a.plane * C3_CS;
    const int chs = a.IH * a.IW;
    const float *img = a.in + (long long)n * a.KC * chs;
    const int padF = 2 - a.padT;
    // input offset of tap t (either dimension): d(t) = (c + pad - t) / 2 with c = (t + pad) & 1; smallest over the taps
    auto dtap = [&](int t) { return (((t + PP) & 1) + padF - t) / 2; };
    const int dmin = min(dtap(0), min(dtap(1), dtap(2)));

    int goff[C3_MAXIT], loff[C3_MAXIT];
    {
        const int r0 = oy0 + dmin, c0 = ox0 + dmin;
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it) {
            const int i = tid + 256 * it;
            goff[it] = -2;
            loff[it] = 0;
            if (i < 4 * a.plane) {
                const int cg = i / a.plane, e = i - cg * a.plane;
                const int r = e / a.pitch, cc = e - r * a.pitch;
                const int gy = r0 + r, gx = c0 + cc;
                const bool ok = gy >= 0 && gx >= 0 && gy < a.IH && gx < a.IW;
                goff[it] = ok ? 4 * cg * chs + gy * a.IW + gx : -1;
                loff[it] = e * C3_CS + 4 * cg;
            }
        }
    }
    auto stage_load = [&](int cbase, f4 (&xr)[C3_MAXIT]) {
        const float *src = img + (long long)cbase * chs;
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it) {
            xr[it] = f4{0.f, 0.f, 0.f, 0.f};
            if (goff[it] >= 0) {
                const float *s0 = src + goff[it];
                xr[it] = f4{s0[0], s0[chs], s0[2 * chs], s0[3 * chs]};
            }
        }
    };
    auto stage_store = [&](float *buf, const f4 (&xr)[C3_MAXIT]) {
#pragma unroll
        for (int it = 0; it < C3_MAXIT; ++it)
            if (goff[it] != -2) *reinterpret_cast<f4 *>(buf + loff[it]) = xr[it];
    };
    const int row = m0 + jn;  // input channel
    const bool rok = row < a.M;
    auto load_a = [&](int cbase, float (&A)[36]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float *w = a.W + ((long long)(cbase + 4 * kq + t) * a.Cin + (rok ? row : 0)) * 9;
            __builtin_memcpy(&A[9 * t], w, 16);
            __builtin_memcpy(&A[9 * t + 4], w + 4, 16);
            A[9 * t + 8] = w[8];
        }
        if (!rok) {
#pragma unroll
            for (int i = 0; i < 36; ++i) A[i] = 0.0f;
        }
    };

    int pb[NT], pyx[NT];  // pyx = (py << 16) | px of the slot inside the tile, -1: empty slot
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pr = 16 * t + jn;
        const bool used = pr < a.TW * a.TH;
        const int py = used ? pr / a.TW : 0, px = used ? pr - py * a.TW : 0;
        pb[t] = (py * a.pitch + px) * C3_CS + 4 * kq;
        pyx[t] = used ? ((py << 16) | px) : -1;
    }
    floatx4 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[c][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = a.KC >> 4;
    float A[36], An[36];
    f4 xr[C3_MAXIT];
    stage_load(0, xr);
    load_a(0, A);
    stage_store(smem, xr);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) {
            stage_load(16 * (c + 1), xr);
            load_a(16 * (c + 1), An);
        }
        const float *buf = smem + (c & 1) * bufsz;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t_y = tap / 3, t_x = tap % 3;
            const int cls = 2 * ((t_y + PP) & 1) + ((t_x + PP) & 1);
            const int toff = ((dtap(t_y) - dmin) * a.pitch + (dtap(t_x) - dmin)) * C3_CS;
            f4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f4 *>(buf + pb[nt] + toff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[cls][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[9 * t + tap], b[nt][t], acc[cls][nt], 0, 0, 0);
        }
        if (more) stage_store(smem + ((c + 1) & 1) * bufsz, xr);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 36; ++i) A[i] = An[i];
    }

#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (pyx[nt] < 0) continue;
            const int oy = 2 * (oy0 + (pyx[nt] >> 16)) + (cls >> 1), ox = 2 * (ox0 + (pyx[nt] & 0xffff)) + (cls & 1);
            if (oy >= a.OH || ox >= a.OW) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = m0 + 4 * kq + r;
                if (ch < a.M) {
                    const long long o = (((long long)n * a.M + ch) * a.OH + oy) * a.OW + ox;
                    a.out[o] = a.res ? acc[cls][nt][r] + a.res[o] : acc[cls][nt][r];
                }
            }
        }
}

// Weight gradient of 3x3 convolutions with many channels (input channels a multiple of 32, output channels a multiple of
// 16 and at least 48), stride S.  GEMM view: dW[co][(ci, tap)] = sum over pixels of dy[co][pixel] * x[ci][pixel shifted by
// the tap]; k = pixels.  A workgroup owns 64 output channels x 32 input channels x 9 taps (wavefront w: channels 16w ..
// 16w+15, 18 accumulator tiles) and a slab of the pixel tiles (TH full rows of one image each); slabs are summed by
// wgrad_reduce.  Both operands sit in LDS as [pixel][channel] with pitches of 68 (dy) and 36 (x) floats: the lanes of an
// MFMA operand read (channel jn, pixel group kq) hit banks jn + 16 * kq, no conflicts, with plain 4-byte reads at any
// tap shift.  Rows of the dy tile are padded to a multiple of 4 pixels (zeros), so the 4 pixels of a lane's k-group
// stay in one row and the x image is addressed as base + tap + t * S pixels.  The next tile is fetched into registers
// while the matrix work of the current one runs.

// S = stride, KS = 3 | 1 (kernel size), NG = groups of 16 input channels per workgroup (KS = 3: 1 or 2, KS = 1: 4).
// KS = 1 (the projections of dimension-changing blocks): no taps, so x is staged like dy, one pixel per slot (only the
// pixels the stride keeps), and the workgroup takes 64 input channels instead of 32.
// CW = groups of 16 output channels per workgroup (4, 2 or 1 for 64+, 32, 16 output channels): the four wavefronts are
// CW channel groups x 4 / CW shares of the k-blocks (wavefront w: group w % CW, k-blocks w / CW, w / CW + 4 / CW, ..);
// the shares are summed in ascending order on the way out, so a layer with 16 output channels still uses all four
// wavefronts' matrix pipes.
template <int S, int KS, int NG, int CW>
__global__ __launch_bounds__(256) void conv_wgrad_c16_kernel(const WgC16Args a, const WgradGroup grp)
{
    constexpr int T2 = KS * KS, CSX = 16 * NG + 4, CGX = 4 * NG, ROW = 16 * NG * T2, TP = ROW + 4;
    constexpr int PW = 4 / CW, WC_CSA = 16 * CW + 4, WC_MAXA = 2 * CW;  // dy staging items per thread: 4 * CW * slots <= 256 * WC_MAXA
    constexpr int MAXX = KS == 1 ? 8 : 5 * NG;  // x staging items per thread: 4 * NG * image pixels <= 256 * MAXX (the plan keeps the image within 320 pixels)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int nslots = a.nkb * 16;
    float *dyS = smem;                       // [nslots][68]
    float *xS = smem + nslots * WC_CSA;      // KS = 3: [XR * XP][CSX]; KS = 1: [nslots][CSX]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;
    // layer l of the launch owns the blocks [l * nslab, (l + 1) * nslab) of the grid's x axis
    const int layer = (int)blockIdx.x / a.nslab;
    const int slab = (int)blockIdx.x - layer * a.nslab, co0 = blockIdx.y * 16 * CW, ci0 = blockIdx.z * 16 * NG;
    const float *__restrict__ gx = grp.x[layer];
    const float *__restrict__ gdy = grp.dy[layer];
    const int cw = wave % CW, pw = wave / CW;
    const int HoWo = a.Ho * a.Wo, HW = a.H * a.Wd, npx = KS == 1 ? nslots : a.XR * a.XP;

    // staging items.  dy: (slot p, group of 4 channels), p-major inside a group; x: (image pixel or slot, group of 4 channels).
    // loc = (row << 16) | [bit 15: column outside the image] | LDS offset, -1 = no item; rel = offset from the tile's first row
    int relA[WC_MAXA], locA[WC_MAXA];
    int relX[MAXX], locX[MAXX];
#pragma unroll
    for (int it = 0; it < WC_MAXA; ++it) {
        const int i = tid + 256 * it;
        relA[it] = -1;
        locA[it] = -1;
        if (i < 4 * CW * nslots) {
            const int cg = i / nslots, p = i - cg * nslots;
            const int py = p / a.PW4, px = p - py * a.PW4;
            const bool ok = px < a.Wo && py < a.TH && co0 + 4 * cg < a.Cout;
            relA[it] = ok ? 4 * cg * HoWo + py * a.Wo + px : -1;
            locA[it] = (py << 16) | (p * WC_CSA + 4 * cg);
        }
    }
#pragma unroll
    for (int it = 0; it < MAXX; ++it) {
        const int i = tid + 256 * it;
        relX[it] = 0;
        locX[it] = -1;
        if (i < CGX * npx) {
            const int cg = i / npx, e = i - cg * npx;
            if (KS == 1) {
                const int py = e / a.PW4, px = e - py * a.PW4;
                const bool ok = px < a.Wo && py < a.TH;
                relX[it] = 4 * cg * HW + py * S * a.Wd + px * S;
                locX[it] = (py << 16) | (ok ? 0 : 0x8000) | (e * CSX + 4 * cg);
            } else {
                const int r = e / a.XP, c = e - r * a.XP;
                const int gx = c - a.pad;
                const bool colok = gx >= 0 && gx < a.Wd;
                relX[it] = 4 * cg * HW + r * a.Wd + gx;
                locX[it] = (r << 16) | (colok ? 0 : 0x8000) | (e * CSX + 4 * cg);
            }
        }
    }
    auto stage_load = [&](int tile, f4 (&ra)[WC_MAXA], f4 (&rx)[MAXX]) {
        const int n = tile / a.tiles_y, ty = tile - n * a.tiles_y;
        const int rows_left = a.Ho - ty * a.TH;
        const float *da = gdy + ((long long)n * a.Cout + co0) * HoWo + (long long)ty * a.TH * a.Wo;
#pragma unroll
        for (int it = 0; it < WC_MAXA; ++it) {
            ra[it] = f4{0.f, 0.f, 0.f, 0.f};
            if (relA[it] >= 0 && (locA[it] >> 16) < rows_left) {
                const float *s0 = da + relA[it];
                ra[it] = f4{s0[0], s0[HoWo], s0[2 * HoWo], s0[3 * HoWo]};
            }
        }
        // KS = 3: image row of the x image's first row (may lie above the image); KS = 1: the tile's first kept row
        const int gy0 = ty * a.TH * S - (KS == 1 ? 0 : a.pad);
        const float *dx = gx + ((long long)n * a.Cin + ci0) * HW + (long long)gy0 * a.Wd;
#pragma unroll
        for (int it = 0; it < MAXX; ++it) {
            rx[it] = f4{0.f, 0.f, 0.f, 0.f};
            const int row = locX[it] >> 16;
            const bool rowok = KS == 1 ? row < rows_left : (gy0 + row >= 0 && gy0 + row < a.H);
            if (locX[it] >= 0 && !(locX[it] & 0x8000) && rowok) {
                const float *s0 = dx + relX[it];
                rx[it] = f4{s0[0], s0[HW], s0[2 * HW], s0[3 * HW]};
            }
        }
    };
    auto stage_store = [&](const f4 (&ra)[WC_MAXA], const f4 (&rx)[MAXX]) {
#pragma unroll
        for (int it = 0; it < WC_MAXA; ++it)
            if (locA[it] >= 0) *reinterpret_cast<f4 *>(dyS + (locA[it] & 0x7fff)) = ra[it];
#pragma unroll
        for (int it = 0; it < MAXX; ++it)
            if (locX[it] >= 0) *reinterpret_cast<f4 *>(xS + (locX[it] & 0x7fff)) = rx[it];
    };

    floatx4 acc[NG][T2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < T2; ++t) acc[g][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    // tiles of this slab: [t0, t1)
    const int t0 = (int)((long long)slab * a.ntiles / a.nslab), t1 = (int)((long long)(slab + 1) * a.ntiles / a.nslab);
    f4 ra[WC_MAXA], rx[MAXX];
    if (t0 < t1) {
        stage_load(t0, ra, rx);
        stage_store(ra, rx);
    }
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const bool more = tile + 1 < t1;
        if (more) stage_load(tile + 1, ra, rx);
        for (int kb = pw; kb < a.nkb; kb += PW) {
            const int p = 16 * kb + 4 * kq;
            const float *ap = dyS + p * WC_CSA + cw * 16 + jn;
            const float *bp;
            if (KS == 1) {
                bp = xS + p * CSX + jn;
            } else {
                const int py = p / a.PW4, px = p - py * a.PW4;
                bp = xS + (py < a.TH ? (py * S * a.XP + px * S) * CSX : 0) + jn;
            }
            float av[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t] = ap[t * WC_CSA];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tap = 0; tap < T2; ++tap) {
                    const float *bt = KS == 1 ? bp + t * CSX : bp + ((tap / 3) * a.XP + tap % 3) * CSX + t * S * CSX;
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bt[16 * g], acc[g][tap], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            stage_store(ra, rx);
            __syncthreads();
        }
    }

    // D[row = 4*kq + r -> co][col = jn -> ci]: through LDS, so that the slab is written as whole rows (output channel
    // co: the workgroup's input channels x taps are contiguous in dW) with 16-byte stores, the k-block shares of a
    // channel group (wavefronts cw, cw + CW, ..) summed in that order
    float *tileS = smem;  // [64][TP], rows 16 * wave ..
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int tap = 0; tap < T2; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) tileS[(wave * 16 + 4 * kq + r) * TP + (16 * g + jn) * T2 + tap] = acc[g][tap][r];
    __syncthreads();
    float *out = grp.part[layer] + (long long)slab * a.nW + (long long)ci0 * T2;
    for (int i = tid; i < 16 * CW * (ROW / 4); i += 256) {
        const int r = i / (ROW / 4), c = i - (ROW / 4) * r;  // r = 16 * (channel group) + row
        if (co0 + r < a.Cout) {
            f4 v = *reinterpret_cast<const f4 *>(tileS + r * TP + 4 * c);
#pragma unroll
            for (int q = 1; q < PW; ++q) v += *reinterpret_cast<const f4 *>(tileS + (q * CW * 16 + r) * TP + 4 * c);
            *reinterpret_cast<f4 *>(out + (long long)(co0 + r) * a.Cin * T2 + 4 * c) = v;
        }
    }
}

// The many-channel 3x3 kernel (conv3x3_c16_kernel).  *taken = false: the shape is not one of its, nothing launched.
template <int MODE>
static int launch_c3(ConvArgs a, int N, hipStream_t st, const char *who, bool *taken)
{
    *taken = false;
    if constexpr (MODE == MODE_FWD_CODES) return 0;
    else {
    if (a.ks != 3 || a.KC % 16 != 0 || a.M < 48 || a.bn_on || a.stat_part || a.bwd_part) return 0;
    const bool par = MODE == MODE_DGRAD && a.dil == 2;  // stride-2 input gradient: by output parity classes
    if (a.dil != 1 && !par) return 0;
    if (MODE == MODE_FWD && (((uintptr_t)a.W & 15) != 0)) return 0;
    if ((long long)a.IH * a.IW * 16 >= (1LL << 31)) return 0;
    const int mgroups = (a.M + 63) / 64;
    // grid the tiles are cut from: the output, or (par) one parity class of it, four of them per image
    const int gw = par ? (a.OW + 1) / 2 : a.OW, gh = par ? (a.OH + 1) / 2 : a.OH;
    const int rs = par ? 1 : a.rs, span = par ? 2 : 3;  // step between neighbouring pixels / extra rows and columns of the LDS image
    // tile = TH x TW output pixels in 16 * NT slots.  Candidates: full-width rows (no ragged last column) or 32 / 16 / 8
    // columns, NT = 2 .. 8; the choice maximises (share of slots that hold a pixel) x (share of the chip's 512
    // workgroup places that the last round of workgroups fills), with a slight preference for larger tiles.
    int NT = 0;
    double best = 0.0;
    const int tws[4] = {gw <= 64 ? gw : 0, 32, 16, 8};
    for (int ti = 0; ti < 4; ++ti) {
        const int tw = tws[ti];
        if (tw <= 0 || (ti > 0 && tw >= 2 * gw)) continue;
        for (int nt = 8; nt >= 2; --nt) {
            if (par && nt != 4 && nt != 2) continue;  // (four accumulator sets per slot)
            int th = std::min(16 * nt / tw, gh);
            if (th < 1) continue;
            const int plane = ((th - 1) * rs + span) * ((tw - 1) * rs + span);
            if (4 * plane > 256 * C3_MAXIT) continue;
            const long long tiles = (long long)((gw + tw - 1) / tw) * ((gh + th - 1) / th);
            const long long blocks = (long long)N * tiles * mgroups;
            const double util = (double)gw * gh / ((double)tiles * 16 * nt);
            const double fill = (double)blocks / (double)(((blocks + 511) / 512) * 512);
            const double score = util * fill * (1.0 - 0.015 * (8 - nt));
            if (score > best) {
                best = score;
                NT = nt;
                a.TW = tw;
                a.TH = th;
            }
        }
    }
    if (NT == 0) return 0;
    a.tiles_x = (gw + a.TW - 1) / a.TW;
    a.tiles_y = (gh + a.TH - 1) / a.TH;
    a.pitch = (a.TW - 1) * rs + span;
    a.rows_in = (a.TH - 1) * rs + span;
    a.plane = a.rows_in * a.pitch;
    a.par = par ? 1 : 0;
    const long long gx = (long long)N * a.tiles_x * a.tiles_y;
    if (gx >= (1LL << 31) || mgroups >= 65536) return 0;
    const size_t lds = (size_t)2 * a.plane * C3_CS * 4;
    dim3 grid((unsigned)gx, (unsigned)mgroups);
    if (par) {
        if constexpr (MODE == MODE_DGRAD) {
            const int pp = (2 - a.padT) & 1;
            if (NT == 4 && pp) hipLaunchKernelGGL((conv3x3_c16_dgrad2_kernel<4, 1>), grid, dim3(256), lds, st, a);
            else if (NT == 4) hipLaunchKernelGGL((conv3x3_c16_dgrad2_kernel<4, 0>), grid, dim3(256), lds, st, a);
            else if (pp) hipLaunchKernelGGL((conv3x3_c16_dgrad2_kernel<2, 1>), grid, dim3(256), lds, st, a);
            else hipLaunchKernelGGL((conv3x3_c16_dgrad2_kernel<2, 0>), grid, dim3(256), lds, st, a);
        }
    } else {
        switch (NT) {
#define AQ_C3(NN) case NN: hipLaunchKernelGGL((conv3x3_c16_kernel<NN, MODE>), grid, dim3(256), lds, st, a); break;
            AQ_C3(2) AQ_C3(3) AQ_C3(4) AQ_C3(5) AQ_C3(6) AQ_C3(7) AQ_C3(8)
#undef AQ_C3
        }
    }
    *taken = true;
    return check_launch(who);
    }
}

}  // namespace aq

int aq_c16_launch_fwd(aq::ConvArgs a, int N, hipStream_t st, const char *who, bool *taken)
{
    return aq::launch_c3<aq::MODE_FWD>(a, N, st, who, taken);
}
int aq_c16_launch_dgrad(aq::ConvArgs a, int N, hipStream_t st, const char *who, bool *taken)
{
    return aq::launch_c3<aq::MODE_DGRAD>(a, N, st, who, taken);
}

// Plan of conv_wgrad_c16_kernel: 3x3 with input channels a multiple of 16, or 1x1 / padding 0 with input channels a
// multiple of 64; output channels a multiple of 16.  nlayers: layers of this shape that share the launch (the slab
// count is chosen so that all of them together make about one round of workgroups over the chip).
WgC16Plan wgrad_c16_plan(int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad, int nlayers)
{
    using namespace aq;
    WgC16Plan p = {};
    if (!(ks == 3 && Cin % 16 == 0) && !(ks == 1 && Cin % 64 == 0 && pad == 0)) return p;
    if (Cout % 16 != 0 || N < 1) return p;
    const int NG = ks == 3 ? (Cin % 32 == 0 ? 2 : 1) : 4, CW = Cout >= 48 ? 4 : (Cout >= 32 ? 2 : 1);
    const int CSX = 16 * NG + 4, CSA = 16 * CW + 4, row = 16 * NG * ks * ks;
    p.NG = NG; p.CW = CW;
    WgC16Args &a = p.a;
    a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Cout = Cout; a.pad = pad;
    a.Ho = conv_out(H, ks, stride, pad);
    a.Wo = conv_out(Wd, ks, stride, pad);
    if ((long long)Cin * H * Wd >= (1LL << 29) || (long long)Cout * a.Ho * a.Wo >= (1LL << 29) || H >= 32768) return p;
    a.PW4 = (a.Wo + 3) & ~3;
    a.XP = (a.PW4 - 1) * stride + ks;
    // rows per tile: at most 128 pixel slots, the x image within its staging and LDS budget; best slot utilisation wins
    double best = 0.0;
    for (int th = 1; th <= a.Ho && th * a.PW4 <= 128; ++th) {
        const int xr = (th - 1) * stride + ks;
        if (ks == 3 && (xr * a.XP > 320 || xr * a.XP * CSX > 300 * 36)) break;  // staging: 4 * NG * pixels <= 256 * (5 * NG) items; LDS
        const int nkb = (th * a.PW4 + 15) / 16, ty = (a.Ho + th - 1) / th;
        const double util = (double)a.Ho * a.Wo / ((double)ty * nkb * 16) * (1.0 - 0.2 / nkb);  // (fixed cost per tile: barriers, staging)
        if (util > best) {
            best = util;
            a.TH = th;
        }
    }
    if (a.TH == 0) return p;
    a.XR = (a.TH - 1) * stride + ks;
    a.nkb = (a.TH * a.PW4 + 15) / 16;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    a.ntiles = N * a.tiles_y;
    a.nW = Cout * Cin * ks * ks;
    const int gy = (Cout + 16 * CW - 1) / (16 * CW), gz = Cin / (16 * NG);
    if (gy >= 65536 || gz >= 65536) return p;
    // slabs: one round of workgroups over the chip's places for all layers of the launch together (every slab costs
    // a pass over dW and a prologue), at least 2 tiles per slab, at most 256 slabs
    const long long per = (long long)gy * gz * std::max(1, nlayers);
    const long long places = CW == 4 ? 512 : 768;  // (the 16- and 32-channel forms need fewer registers: three workgroups per CU)
    long long want = (places + per - 1) / per;
    want = std::max<long long>(1, std::min<long long>(std::min<long long>(want, 256), (a.ntiles + 1) / 2));
    a.nslab = (int)std::min<long long>(want, a.ntiles);
    p.grid = dim3((unsigned)a.nslab, (unsigned)gy, (unsigned)gz);
    const size_t ximg = ks == 1 ? (size_t)a.nkb * 16 : (size_t)a.XR * a.XP;
    p.lds = std::max(((size_t)a.nkb * 16 * CSA + ximg * CSX) * 4, (size_t)64 * (row + 4) * 4);
    p.ws_bytes = (size_t)a.nslab * a.nW * 4;
    p.use = true;
    return p;
}

// launch of a planned conv_wgrad_c16_kernel; 0 or a negative error code.  A tile of more than 64 KB of dynamic LDS (the
// epilogue tile of the forms with two input-channel groups: 74 752 bytes) needs the kernel's limit raised first: done once
// per instantiation and size (not per launch), and a refusal - a device without that much LDS per workgroup - is reported
// with the size asked for instead of surfacing as a generic launch failure.
int wgrad_c16_launch(const WgC16Plan &cp, const aq::WgradGroup &g, int ks, int stride, hipStream_t st)
{
    using namespace aq;
    dim3 grid = cp.grid;
    grid.x *= (unsigned)g.n;
#define AQ_WC(SS, KK, GG, CC)                                                                                                      \
    {                                                                                                                              \
        auto kern = conv_wgrad_c16_kernel<SS, KK, GG, CC>;                                                                         \
        static size_t granted = 64 * 1024; /* (per instantiation; one device per process) */                                      \
        if (cp.lds > granted) {                                                                                                    \
            const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp.lds); \
            AQ_REQUIRE(e == hipSuccess, -ENOMEM, "aq_conv_wgrad: the weight-gradient kernel needs %zu bytes of LDS per workgroup: %s", \
                       cp.lds, hipGetErrorString(e));                                                                              \
            granted = cp.lds;                                                                                                      \
        }                                                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), cp.lds, st, cp.a, g);                                                            \
    }
#define AQ_WCC(SS, KK, GG)                        \
    if (cp.CW == 4) AQ_WC(SS, KK, GG, 4)          \
    else if (cp.CW == 2) AQ_WC(SS, KK, GG, 2)     \
    else AQ_WC(SS, KK, GG, 1)
    if (ks == 1) {
        if (stride == 1) { AQ_WCC(1, 1, 4) } else { AQ_WCC(2, 1, 4) }
    } else if (cp.NG == 2) {
        if (stride == 1) { AQ_WCC(1, 3, 2) } else { AQ_WCC(2, 3, 2) }
    } else {
        if (stride == 1) { AQ_WCC(1, 3, 1) } else { AQ_WCC(2, 3, 1) }
    }
#undef AQ_WCC
#undef AQ_WC
    return 0;
}

