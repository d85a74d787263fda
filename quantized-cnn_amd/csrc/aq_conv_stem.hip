// The image stem: a KS x KS convolution over very few input channels with many output channels (7x7, stride 2, padding 3
// over RGB in the torchvision ResNet layout of resnet18.py; adp_qtz_fan.py:574-601 is written for any kernel size, the
// reference's own models start with a 3x3 layer that keeps its kernel in aq_conv.hip).  Forward and weight gradient; the
// input of a stem needs no gradient.  fp32 matrix cores (v_mfma_f32_16x16x4_f32), NCHW, no atomics: bit-reproducible.
//
// Both kernels work on one output row at a time: the KS input rows it touches (all channels) are the LDS image,
// [channel][row][column], fetched one tile ahead into registers by coalesced row reads.  A workgroup walks over the rows
// gridDim.x apart, so whatever depends on the layer only (the weights of the forward pass, the tap offsets) is set up
// once per workgroup and lives in registers.
//  * forward: the GEMM is 64 output channels x 16 * NT pixels x K = CIN * KS * KS per workgroup; the four wavefronts take
//    16 output channels each.  A fragments (weights) stay in registers for the whole launch (one float per k-step and
//    lane), B fragments are 4-byte LDS reads at tap offset + stride * pixel.
//  * weight gradient: dW[co][tap] = sum over pixels dy[co][pixel] * x[tap, pixel]: M = 64 output channels (16 per
//    wavefront), N = the K taps in tiles of 16, k = the pixels of the row, four per MFMA.  dy rows sit in LDS with a
//    pitch of 4 * odd floats, so the 16 channels x 4 pixels of an A read hit 64 different banks.  Every workgroup
//    accumulates its rows in registers and writes one slab; the slabs are summed by the usual reduction.
#include "aq_conv_shared.h"

#include <algorithm>

namespace aq {

struct StemArgs {
    const float *x, *W, *dy;
    float *out;          // y (forward) or the slabs [gridDim.x][Cout][K] (weight gradient)
    int N, H, Wd, Cout, OH, OW, pad;
    int pitch, plane;    // LDS image: KS rows of pitch floats per channel, plane = KS * pitch
    int rowlen;          // columns of the image that are staged
    int zoff;            // start of an area of zeros that the k slots beyond K read
    int ntiles;          // N * OH
    int dpitch, ow4, doff;  // weight gradient: pitch of the dy rows, float4 per dy row, start of the dy tile
};

// LDS offset of tap t = (c * KS + ky) * KS + kx
template <int KS>
__device__ __forceinline__ int stem_tap_offset(int t, int plane, int pitch)
{
    const int c = t / (KS * KS), r = t - c * KS * KS, ky = r / KS, kx = r - ky * KS;
    return c * plane + ky * pitch + kx;
}

// 16 bytes through a buffer resource (range-checked like buf_load)
typedef unsigned stem_u4 __attribute__((ext_vector_type(4)));
typedef float stem_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ stem_f4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(stem_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int CIN, int KS, int S>
struct StemImage {
    static constexpr int ROWS = CIN * KS;
    // row rr = c * KS + r of the tile (n, oy): thread tid reads column tid (image column tid - pad); what lies outside the
    // image comes back as 0 from the range check (no branches: the rows stay independent loads)
    __device__ static __forceinline__ void fetch(const StemArgs &a, __amdgpu_buffer_rsrc_t rx, int tile, int tid, float (&xr)[ROWS])
    {
        const int n = tile / a.OH, oy = tile - n * a.OH;
        const int gx = tid - a.pad;
        const unsigned col = (tid < a.rowlen && gx >= 0 && gx < a.Wd) ? 4u * (unsigned)gx : BUF_OOB;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int c = rr / KS, r = rr - c * KS;
            const int iy = oy * S - a.pad + r;
            const bool rowok = iy >= 0 && iy < a.H;
            const unsigned soff = rowok ? (unsigned)((((n * CIN + c) * a.H + iy) * a.Wd) * 4) : 0u;
            xr[rr] = buf_load(rx, rowok ? col : BUF_OOB, soff);
        }
    }
    __device__ static __forceinline__ void store(const StemArgs &a, float *smem, int tid, const float (&xr)[ROWS])
    {
        if (tid < a.rowlen) {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int c = rr / KS, r = rr - c * KS;
                smem[c * a.plane + r * a.pitch + tid] = xr[rr];
            }
        }
    }
};

template <int CIN, int KS, int S, int NT>
__global__ __launch_bounds__(256, 2) void conv_stem_fwd_kernel(const StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = CIN * KS * KS, KSTEPS = (K + 3) / 4;
    typedef StemImage<CIN, KS, S> Img;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;
    const int cob = blockIdx.y * 64 + wave * 16;

    // k slot (ks, kq) = tap 4 * ks + kq; slots beyond K multiply zeros (weight 0, the zero area of the image)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)a.N * CIN * a.H * a.Wd * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.W, (unsigned)a.Cout * K * 4u);
    float A[KSTEPS];
    int boff[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k = 4 * ks + kq;
        const bool ok = k < K;
        A[ks] = buf_load(rw, ok ? (unsigned)(((cob + jn) * K + k) * 4) : BUF_OOB, 0);
        boff[ks] = (ok ? stem_tap_offset<KS>(k, a.plane, a.pitch) : a.zoff) + S * jn;
    }
    for (int i = tid; i < S * 16 * NT; i += 256) smem[a.zoff + i] = 0.0f;

    float xr[Img::ROWS];
    int tile = blockIdx.x;
    if (tile < a.ntiles) Img::fetch(a, rx, tile, tid, xr);
    for (; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();  // the previous row's reads are done
        Img::store(a, smem, tid, xr);
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) Img::fetch(a, rx, tile + gridDim.x, tid, xr);

        floatx4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        // the B fragments of k-step ks + 1 are requested before the matrix work of k-step ks (pinned: left alone, the
        // scheduler sinks every read to its use and each pair of MFMAs waits for an LDS round trip)
        float b[NT], bn[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = smem[boff[0] + S * 16 * nt];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + 1 < KSTEPS) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bn[nt] = smem[boff[ks + 1] + S * 16 * nt];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ks], b[nt], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
        }

        // D[row = 4 * kq + r][col = jn]
        const int n = tile / a.OH, oy = tile - n * a.OH;
        float *orow = a.out + (((long long)n * a.Cout + cob + 4 * kq) * a.OH + oy) * a.OW;
        const long long chs = (long long)a.OH * a.OW;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ox = 16 * nt + jn;
            if (ox < a.OW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) orow[r * chs + ox] = acc[nt][r];
            }
        }
    }
}

template <int CIN, int KS, int S>
__global__ __launch_bounds__(256, 2) void conv_stem_wgrad_kernel(const StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = CIN * KS * KS, NK = (K + 15) / 16, DIT = 7;  // DIT: float4 of dy per thread and row, 64 * ow4 <= 256 * DIT
    typedef StemImage<CIN, KS, S> Img;
    typedef stem_f4 f4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jn = lane & 15, kq = lane >> 4;
    const int cob = blockIdx.y * 64;

    // B: tap 16 * nk + jn at pixel 4 * ps + kq; taps beyond K read the zero area
    int boff[NK];
#pragma unroll
    for (int nk = 0; nk < NK; ++nk) {
        const int t = 16 * nk + jn;
        boff[nk] = (t < K ? stem_tap_offset<KS>(t, a.plane, a.pitch) : a.zoff) + S * kq;
    }
    const int aoff = a.doff + (wave * 16 + jn) * a.dpitch + kq;
    for (int i = tid; i < S * 4 * a.ow4; i += 256) smem[a.zoff + i] = 0.0f;

    // dy staging: item i = (channel, float4 of the row)
    int dgo[DIT], dlo[DIT];
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
        const int i = tid + 256 * it;
        const int ch = i / a.ow4, q = i - ch * a.ow4;
        const bool ok = i < 64 * a.ow4;
        dgo[it] = ok ? (ch * a.OH * a.OW + 4 * q) * 4 : (int)BUF_OOB;
        dlo[it] = ok ? a.doff + ch * a.dpitch + 4 * q : -1;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)a.N * CIN * a.H * a.Wd * 4u);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, (unsigned)a.N * a.Cout * a.OH * a.OW * 4u);
    auto fetch_dy = [&](int tile, f4 (&dr)[DIT]) {
        const int n = tile / a.OH, oy = tile - n * a.OH;
        const unsigned soff = (unsigned)(((n * a.Cout + cob) * a.OH + oy) * a.OW) * 4u;
#pragma unroll
        for (int it = 0; it < DIT; ++it) dr[it] = buf_load4(rdy, (unsigned)dgo[it], soff);
    };

    floatx4 acc[NK];
#pragma unroll
    for (int nk = 0; nk < NK; ++nk) acc[nk] = floatx4{0.f, 0.f, 0.f, 0.f};

    float xr[Img::ROWS];
    f4 dr[DIT];
    int tile = blockIdx.x;
    if (tile < a.ntiles) {
        Img::fetch(a, rx, tile, tid, xr);
        fetch_dy(tile, dr);
    }
    for (; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();
        Img::store(a, smem, tid, xr);
#pragma unroll
        for (int it = 0; it < DIT; ++it)
            if (dlo[it] >= 0) *reinterpret_cast<f4 *>(smem + dlo[it]) = dr[it];
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) {
            Img::fetch(a, rx, tile + gridDim.x, tid, xr);
            fetch_dy(tile + gridDim.x, dr);
        }
        // operands of pixel group ps + 1 are requested before the matrix work of group ps (the last round reads its own again)
        float av = smem[aoff], b[NK];
#pragma unroll
        for (int nk = 0; nk < NK; ++nk) b[nk] = smem[boff[nk]];
        for (int ps = 0; ps < a.ow4; ++ps) {
            const int pn = min(ps + 1, a.ow4 - 1);
            const float avn = smem[aoff + 4 * pn];
            float bn[NK];
#pragma unroll
            for (int nk = 0; nk < NK; ++nk) bn[nk] = smem[boff[nk] + S * 4 * pn];
            __builtin_amdgcn_sched_barrier(0);  // (pinned: the scheduler would sink the reads to their uses)
#pragma unroll
            for (int nk = 0; nk < NK; ++nk) acc[nk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nk], acc[nk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            av = avn;
#pragma unroll
            for (int nk = 0; nk < NK; ++nk) b[nk] = bn[nk];
        }
    }

    // D[row = 4 * kq + r][col = jn]: dW[cob + 16 * wave + 4 * kq + r][16 * nk + jn]
    float *slab = a.out + ((long long)blockIdx.x * a.Cout + cob + wave * 16 + 4 * kq) * K;
#pragma unroll
    for (int nk = 0; nk < NK; ++nk) {
        const int t = 16 * nk + jn;
        if (t < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[r * K + t] = acc[nk][r];
        }
    }
}

struct StemPlan {
    bool use;
    int NT;
    StemArgs a;
    dim3 grid;
    size_t lds;
};

// shapes of these kernels: RGB, 7x7, stride 2, output channels a multiple of 64, rows of at most 112 output pixels
// (weight gradient: a multiple of 4 pixels)
static StemPlan stem_plan(int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad, bool wgrad)
{
    StemPlan p = {};
    if (Cin != 3 || ks != 7 || stride != 2 || Cout % 64 != 0 || N < 1) return p;
    StemArgs &a = p.a;
    a.N = N; a.H = H; a.Wd = Wd; a.Cout = Cout; a.pad = pad;
    a.OH = conv_out(H, ks, stride, pad);
    a.OW = conv_out(Wd, ks, stride, pad);
    if (a.OW > 112 || a.OH < 1 || a.OW < 1) return p;
    if (wgrad && (a.OW & 3)) return p;
    // (32-bit byte offsets into x, dy / y and W)
    if ((long long)N * Cin * H * Wd >= (1LL << 29) || (long long)N * Cout * a.OH * a.OW >= (1LL << 29) || (long long)Cout * 147 >= (1LL << 29)) return p;
    p.NT = a.OW <= 32 ? 2 : (a.OW <= 64 ? 4 : 7);
    const int slots = wgrad ? a.OW : 16 * p.NT;
    a.rowlen = (slots - 1) * stride + ks;  // <= 229
    a.pitch = (a.rowlen + 3) & ~3;
    a.plane = ks * a.pitch;
    a.zoff = Cin * a.plane;
    a.ntiles = N * a.OH;
    const int zeros = stride * slots;
    size_t floats = (size_t)a.zoff + zeros;
    if (wgrad) {
        a.ow4 = a.OW / 4;
        a.dpitch = a.OW + ((a.ow4 & 1) ? 8 : 4);  // 4 * odd
        a.doff = (int)((floats + 3) & ~(size_t)3);
        floats = (size_t)a.doff + (size_t)64 * a.dpitch;
    }
    p.lds = floats * 4;
    const int gy = Cout / 64;
    if (gy >= 65536) return p;
    // two workgroups per CU; the rows divide evenly over them for the usual image sizes
    // (measured at 32 x 224 x 224: 512 workgroups 83 us, 768 86 us, 1024 86 us, 256 86 us, counts that leave a ragged last round 91-101 us)
    p.grid = dim3((unsigned)std::max(1, std::min(a.ntiles, 512 / std::min(gy, 512))), (unsigned)gy);
    p.use = true;
    return p;
}

}  // namespace aq

int aq_stem_launch_fwd(const aq::ConvArgs &c, int N, hipStream_t st, const char *who, bool *taken)
{
    using namespace aq;
    *taken = false;
    if (c.res || c.bn_on || c.stat_part || c.dil != 1) return 0;
    StemPlan p = stem_plan(N, c.Cin, c.IH, c.IW, c.Cout, c.ks, c.rs, c.padT, false);
    if (!p.use) return 0;
    p.a.x = c.in; p.a.W = c.W; p.a.out = c.out;
    if (p.NT == 2) hipLaunchKernelGGL((conv_stem_fwd_kernel<3, 7, 2, 2>), p.grid, dim3(256), p.lds, st, p.a);
    else if (p.NT == 4) hipLaunchKernelGGL((conv_stem_fwd_kernel<3, 7, 2, 4>), p.grid, dim3(256), p.lds, st, p.a);
    else hipLaunchKernelGGL((conv_stem_fwd_kernel<3, 7, 2, 7>), p.grid, dim3(256), p.lds, st, p.a);
    *taken = true;
    return check_launch(who);
}

WgStemPlan wgrad_stem_plan(const float *dy, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad)
{
    using namespace aq;
    WgStemPlan w = {};
    if (((uintptr_t)dy & 15) != 0) return w;
    const StemPlan p = stem_plan(N, Cin, H, Wd, Cout, ks, stride, pad, true);
    if (!p.use) return w;
    w.use = true;
    w.nslab = (int)p.grid.x;
    w.ws_bytes = (size_t)w.nslab * Cout * Cin * ks * ks * 4;
    return w;
}

int wgrad_stem_launch(const float *x, const float *dy, float *part, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad,
                      hipStream_t st)
{
    using namespace aq;
    StemPlan p = stem_plan(N, Cin, H, Wd, Cout, ks, stride, pad, true);
    AQ_REQUIRE(p.use && ((uintptr_t)dy & 15) == 0, -ENOTSUP, "aq_conv_wgrad: not a shape of the stem kernel");
    p.a.x = x; p.a.dy = dy; p.a.out = part;
    hipLaunchKernelGGL((conv_stem_wgrad_kernel<3, 7, 2>), p.grid, dim3(256), p.lds, st, p.a);
    return 0;
}
