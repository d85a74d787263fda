// Ternary / power-of-two weight quantisers of the papers the reference cites (README.md:3-29): Ternary Weight
// Networks (Li, Zhang, Liu 2016), Trained Ternary Quantization (Zhu, Han, Mao, Dally 2017) and Incremental Network
// Quantization (Zhou et al. 2017).  The reference's scripts contain none of them (only the citations), so these
// kernels restate the published algorithms; their CPU checker (oracle/aq_oracle.c, aqo_tern_* / aqo_inq_*) is
// written from the same papers: parity unpinned - no reference implementation.
//
//   forward (all layers of a model in one call, five small launches):
//     A  per-wavefront partials of (sum |w|, max |w|)                     [HBM pass 1 over w]
//     B  one workgroup per layer: partials -> delta = f * mean|w|  (or f * max|w|)
//     C  codes: 0 where |w| <= delta, 1 where w > delta, 2 where w < -delta; optionally the same codes packed
//        2 bits each, and the decoded weights q[code]; partials of (sum of |w| above delta, their count,
//        positives, negatives)                                            [HBM pass 2 over w]
//     D  one workgroup per layer: TWN alpha = mean of |w| above delta, codebook q = {0, +pos, -neg}
//     E  (TWN only, when decoded weights are wanted) wt = q[code]
//   backward (one launch + one reduction launch):
//     straight-through gradient of the latent weights, scaled as TTQ prescribes, with the two scale gradients
//     as masked sums; or the gradient masked by INQ's partition.
// All cross-element sums are fixed-order (per-lane chain, wavefront butterfly, per-wavefront partial, then a
// fixed-order second stage): run-to-run deterministic, no atomics.
#include "aq_common.h"
#include "../../include/aq_hip.h"

#include <algorithm>

namespace aq {

constexpr int TERN_MAX_SEGS = 32;
constexpr int TERN_ELEMS_PER_WAVE = 64 * 4 * 8;  // 8 coalesced 16-byte loads per lane
constexpr int TERN_ELEMS_PER_WAVE_2IN = 64 * 4 * 4;  // kernels with two input streams: 4 loads per lane and stream (8 in flight)

struct TernSegDev {
    const float *w;
    const float *dwt;
    const uint8_t *mask;   // INQ: 1 = still trainable, 0 = frozen
    uint8_t *codes, *packed;
    float *wt, *dw;
    float *stats;          // [4] delta, pos, neg, nonzero count
    float *q;              // [3] codebook
    const float *wp, *wn;
    float *dwp, *dwn;
    float *part;           // per-workgroup partials of this layer
    float *part2;          // second level: one partial per LEVEL_CHUNK first-level partials (layers with many of them)
    long long n;
    int blk0, nwave;
    int n2;                // number of second-level partials (0: the last stage reads the first level)
};
struct TernTab {
    int nseg;
    int nt;  // tensors beyond the last-level cache: the host launches the streaming (nontemporal) instantiations
    TernSegDev s[TERN_MAX_SEGS];
};

__device__ __forceinline__ int tern_segment(const TernTab &tab)
{
    int si = 0;
    for (int i = 1; i < tab.nseg; ++i)
        if ((int)blockIdx.x >= tab.s[i].blk0) si = i;
    return si;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// One partial per wavefront and quantity, written by lane 0: no barrier, so a wavefront retires as soon as its own
// loads are consumed (with a workgroup-level sum the streaming kernels measured 3-4 % slower).  Layers with
// thousands of partials get a middle reduction stage (tern_level_kernel).
template <int F>
__device__ __forceinline__ void wave_partials(const float (&vals)[F], float *part, int wt, int nwave)
{
    if ((threadIdx.x & 63) == 0 && wt < nwave) {
#pragma unroll
        for (int f = 0; f < F; ++f) part[wt * F + f] = vals[f];
    }
}

// element range [e0, e1) of wavefront `wt` of a layer, and this lane's share: elements e0 + 256*u + 4*lane + k.
// NT is a template parameter on purpose: with a run-time flag the two branches hold the same loads, the compiler merges
// them and the merged load loses its nontemporal marking (plain loads: 6.1 instead of 6.9 TB/s on a read-only pass).
typedef float tern_f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__device__ __forceinline__ void load_run(const float *__restrict__ w, long long e0, long long n, int lane, float (&v)[U][4], bool vec)
{
    if (vec && e0 + 256 * U <= n) {  // whole wavefront inside the tensor (uniform): U loads issued back to back
        const tern_f4 *src = reinterpret_cast<const tern_f4 *>(w + e0) + lane;
        tern_f4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(src + 64 * u) : src[64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u][0] = t[u].x; v[u][1] = t[u].y; v[u][2] = t[u].z; v[u][3] = t[u].w;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long e = e0 + 256 * u + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = e + k < n ? w[e + k] : 0.0f;
    }
}

// A: partials (sum |w|, max |w|) per wavefront
template <bool NT>
__global__ __launch_bounds__(256) void tern_absstat_kernel(const TernTab tab)
{
    const int si = tern_segment(tab);
    const TernSegDev &s = tab.s[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x - s.blk0;
    const int wt = blk * 4 + wave;
    const long long e0 = (long long)wt * TERN_ELEMS_PER_WAVE;  // (wavefronts beyond the layer load nothing: all zeros)
    float v[8][4];
    load_run<8, NT>(s.w, e0, s.n, lane, v, ((uintptr_t)s.w & 15) == 0);
    float sum = 0.0f, mx = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = fabsf(v[u][k]);
            sum += a;
            mx = fmaxf(mx, a);
        }
    const float vals[2] = {wave_sum(sum), wave_max(mx)};
    wave_partials<2>(vals, s.part, wt, s.nwave);
}

// fixed-order sum of `cnt` strided values: thread t adds t, t+256, ..; then a halving tree over the threads
__device__ __forceinline__ float block_sum_strided(const float *p, int cnt, int stride, float *sh)
{
    float a = 0.0f;
    const int T = blockDim.x;  // 256 or 1024 threads (a power of two)
    int i = threadIdx.x;
    for (; i + T * 7 < cnt; i += T * 8) {  // eight independent loads in flight, added in ascending order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(i + T * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; i < cnt; i += T) a += p[(long long)i * stride];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = T >> 1; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max_strided(const float *p, int cnt, int stride, float *sh)
{
    float a = 0.0f;
    const int T = blockDim.x;
    for (int i = threadIdx.x; i < cnt; i += T) a = fmaxf(a, p[(long long)i * stride]);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = T >> 1; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// Middle stage for layers with thousands of first-level partials: workgroup b adds (or maximises) the partials
// [b * LEVEL_CHUNK, (b + 1) * LEVEL_CHUNK) of every quantity, in a fixed order, into one second-level partial.
constexpr int LEVEL_CHUNK = 8192;
template <int F, int MAXF>
__global__ __launch_bounds__(1024) void tern_level_kernel(const float *__restrict__ src, int cnt, float *__restrict__ dst)
{
    __shared__ float sh[1024];
    const int lo = blockIdx.x * LEVEL_CHUNK, c = min(LEVEL_CHUNK, cnt - lo);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float r = (f == MAXF) ? block_max_strided(src + (long long)lo * F + f, c, F, sh) : block_sum_strided(src + (long long)lo * F + f, c, F, sh);
        if (threadIdx.x == 0) dst[blockIdx.x * F + f] = r;
    }
}
__device__ __forceinline__ const float *last_level(const TernSegDev &s, int &cnt)
{
    cnt = s.n2 ? s.n2 : s.nwave;
    return s.n2 ? s.part2 : s.part;
}

// B: delta of every layer (one workgroup per layer)
__global__ __launch_bounds__(1024) void tern_delta_kernel(const TernTab tab, const float factor, const int thr_max)
{
    __shared__ float sh[1024];
    const TernSegDev &s = tab.s[blockIdx.x];
    int nblk;
    const float *pp = last_level(s, nblk);
    const float sum = block_sum_strided(pp, nblk, 2, sh);
    const float mx = block_max_strided(pp + 1, nblk, 2, sh);
    if (threadIdx.x == 0) {
        const float mean = sum / (float)s.n;
        s.stats[0] = factor * (thr_max ? mx : mean);
    }
}

// C: codes (+ packed codes, decoded weights when the codebook is already known) and partials of the second reduction
template <bool TTQ, bool NT>
__global__ __launch_bounds__(256) void tern_emit_kernel(const TernTab tab)
{
    const int si = tern_segment(tab);
    const TernSegDev &s = tab.s[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x - s.blk0;
    const int wt = blk * 4 + wave;
    const long long e0 = (long long)wt * TERN_ELEMS_PER_WAVE;
    const float delta = s.stats[0];
    float pos = 0.0f, neg = 0.0f;
    if (TTQ) {
        pos = s.wp[0];
        neg = s.wn[0];
    }
    float v[8][4];
    const bool vec = ((uintptr_t)s.w & 15) == 0;
    load_run<8, NT>(s.w, e0, s.n, lane, v, vec);
    float sabs = 0.0f, npos = 0.0f, nneg = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long e = e0 + 256 * u + 4 * lane;
        unsigned c4 = 0, p1 = 0;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = v[u][k];
            const unsigned c = x > delta ? 1u : (x < -delta ? 2u : 0u);
            c4 |= c << (8 * k);
            p1 |= c << (2 * k);
            sabs += c ? fabsf(x) : 0.0f;
            npos += c == 1u ? 1.0f : 0.0f;
            nneg += c == 2u ? 1.0f : 0.0f;
            o[k] = c == 1u ? pos : (c == 2u ? -neg : 0.0f);
        }
        if (s.codes) {
            if (e + 3 < s.n && ((uintptr_t)s.codes & 3) == 0) {
                if (NT) __builtin_nontemporal_store(c4, reinterpret_cast<unsigned *>(s.codes + e));
                else *reinterpret_cast<unsigned *>(s.codes + e) = c4;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k < s.n) s.codes[e + k] = (uint8_t)(c4 >> (8 * k));
            }
        }
        if (s.packed) {
            // the four bytes of a lane quad form one 32-bit word (16 codes): gathered with quad permutes
            const int b0 = __builtin_amdgcn_update_dpp(0, (int)p1, 0x00, 0xF, 0xF, false);  // quad_perm [0,0,0,0]
            const int b1 = __builtin_amdgcn_update_dpp(0, (int)p1, 0x55, 0xF, 0xF, false);  // [1,1,1,1]
            const int b2 = __builtin_amdgcn_update_dpp(0, (int)p1, 0xAA, 0xF, 0xF, false);  // [2,2,2,2]
            const int b3 = __builtin_amdgcn_update_dpp(0, (int)p1, 0xFF, 0xF, 0xF, false);  // [3,3,3,3]
            const unsigned word = (unsigned)b0 | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
            if ((lane & 3) == 0 && e < s.n) {
                const long long byte0 = e >> 2;  // e is a multiple of 16 here
                const long long nbytes = (s.n + 3) >> 2;
                if (byte0 + 3 < nbytes && ((uintptr_t)s.packed & 3) == 0) *reinterpret_cast<unsigned *>(s.packed + byte0) = word;
                else
                    for (int b = 0; b < 4; ++b)
                        if (byte0 + b < nbytes) s.packed[byte0 + b] = (uint8_t)(word >> (8 * b));
            }
        }
        if (TTQ && s.wt) {
            if (vec && ((uintptr_t)s.wt & 15) == 0 && e + 3 < s.n) {
                const tern_f4 t = {o[0], o[1], o[2], o[3]};
                if (NT) __builtin_nontemporal_store(t, reinterpret_cast<tern_f4 *>(s.wt + e));
                else *reinterpret_cast<tern_f4 *>(s.wt + e) = t;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k < s.n) s.wt[e + k] = o[k];
            }
        }
    }
    const float vals[3] = {wave_sum(sabs), wave_sum(npos), wave_sum(nneg)};
    wave_partials<3>(vals, s.part, wt, s.nwave);
}

// D: codebook of every layer.  TWN: alpha = mean of |w| above delta for both signs; TTQ: the learned scales.
template <bool TTQ>
__global__ __launch_bounds__(1024) void tern_codebook_kernel(const TernTab tab)
{
    __shared__ float sh[1024];
    const TernSegDev &s = tab.s[blockIdx.x];
    int nblk;
    const float *pp = last_level(s, nblk);
    const float sabs = block_sum_strided(pp, nblk, 3, sh);
    const float npos = block_sum_strided(pp + 1, nblk, 3, sh);
    const float nneg = block_sum_strided(pp + 2, nblk, 3, sh);
    if (threadIdx.x == 0) {
        const float cnt = npos + nneg;
        float pos, neg;
        if (TTQ) {
            pos = s.wp[0];
            neg = s.wn[0];
        } else {
            pos = neg = cnt > 0.0f ? sabs / cnt : 0.0f;
        }
        s.stats[1] = pos;
        s.stats[2] = neg;
        s.stats[3] = cnt;
        s.q[0] = 0.0f;
        s.q[1] = pos;
        s.q[2] = -neg;
    }
}

// E: wt = q[code]
__global__ __launch_bounds__(256) void tern_decode_kernel(const TernTab tab)
{
    const int si = tern_segment(tab);
    const TernSegDev &s = tab.s[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = ((int)blockIdx.x - s.blk0) * 4 + wave;
    if (wt >= s.nwave || !s.wt) return;
    const long long e0 = (long long)wt * TERN_ELEMS_PER_WAVE;
    const float q1 = s.q[1], q2 = s.q[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long e = e0 + 256 * u + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e + k < s.n) {
                const unsigned c = s.codes[e + k];
                s.wt[e + k] = c == 1u ? q1 : (c == 2u ? q2 : 0.0f);
            }
    }
}

// backward.  MODE 0 (TWN): dw = dwt (straight through).  MODE 1 (TTQ): dw = pos * dwt where w > delta, neg * dwt where
// w < -delta, dwt in between; partials of sum dwt over the positives and over the negatives.  MODE 2 (INQ):
// dw = dwt where mask != 0, else 0.
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void tern_bwd_kernel(const TernTab tab)
{
    constexpr int U = MODE == 1 ? 4 : 8;  // loads per lane and input stream: eight in flight either way
    const int si = tern_segment(tab);
    const TernSegDev &s = tab.s[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x - s.blk0;
    const int wt = blk * 4 + wave;
    const long long e0 = (long long)wt * (256 * U);
    float delta = 0.0f, pos = 1.0f, neg = 1.0f;
    if (MODE == 1) {
        delta = s.stats[0];
        pos = s.wp[0];
        neg = s.wn[0];
    }
    float g[U][4], v[U][4];
    const bool vec = (((uintptr_t)s.dwt | (uintptr_t)s.dw | (uintptr_t)s.w) & 15) == 0;
    load_run<U, NT>(s.dwt, e0, s.n, lane, g, vec);
    if (MODE == 1) load_run<U, NT>(s.w, e0, s.n, lane, v, vec);
    float sp = 0.0f, sn = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long e = e0 + 256 * u + 4 * lane;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = g[u][k];
            if (MODE == 1) {
                const float x = v[u][k];
                if (x > delta) {
                    sp += d;
                    d *= pos;
                } else if (x < -delta) {
                    sn += d;
                    d *= neg;
                }
            } else if (MODE == 2) {
                d = (e + k < s.n && s.mask[e + k]) ? d : 0.0f;
            }
            o[k] = d;
        }
        if (vec && e + 3 < s.n) {
            const tern_f4 t = {o[0], o[1], o[2], o[3]};
            if (NT) __builtin_nontemporal_store(t, reinterpret_cast<tern_f4 *>(s.dw + e));
            else *reinterpret_cast<tern_f4 *>(s.dw + e) = t;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < s.n) s.dw[e + k] = o[k];
        }
    }
    if (MODE == 1) {
        const float vals[2] = {wave_sum(sp), wave_sum(sn)};
        wave_partials<2>(vals, s.part, wt, s.nwave);
    }
}

// d loss / d Wp = sum over the positives of dwt;  d loss / d Wn = - sum over the negatives (w~ = -Wn there)
__global__ __launch_bounds__(1024) void tern_scale_grad_kernel(const TernTab tab)
{
    __shared__ float sh[1024];
    const TernSegDev &s = tab.s[blockIdx.x];
    int nblk;
    const float *pp = last_level(s, nblk);
    const float sp = block_sum_strided(pp, nblk, 2, sh);
    const float sn = block_sum_strided(pp + 1, nblk, 2, sh);
    if (threadIdx.x == 0) {
        if (s.dwp) s.dwp[0] = sp;
        if (s.dwn) s.dwn[0] = -sn;
    }
}

// INQ (Zhou et al. 2017, section 2): weights still trainable (mask 1) whose magnitude reaches tau join the frozen
// partition: w <- sign(w) * 2^k with k chosen by the paper's rule (the power of two beta with (alpha + beta) / 2
// <= |w| < 3 beta / 2, alpha the next smaller level; 0 below 2^(n2-1); at most 2^n1), mask <- 0.  codes: for frozen
// weights 0 = zero, 1 + 2 * (k - n2) + (w < 0) otherwise; 255 for weights still trainable.
__global__ __launch_bounds__(256) void inq_partition_kernel(float *__restrict__ w, uint8_t *__restrict__ mask, uint8_t *__restrict__ codes, long long n,
                                                            float tau, int n1, int n2)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = w[i];
    unsigned m = mask[i];
    if (m && fabsf(x) >= tau) {
        const float a = fabsf(x);
        const unsigned b = __builtin_bit_cast(unsigned, a);
        int k = (int)((b >> 23) & 0xff) - 127 + ((b & 0x7fffffu) >= 0x400000u ? 1 : 0);  // 0.75 * 2^(k+1) = 1.5 * 2^k
        float qv = 0.0f;
        if ((b >> 23) == 0) k = -200;  // zero / subnormal
        if (k < n2) k = a >= ldexpf(1.0f, n2 - 1) ? n2 : -200;
        if (k > n1) k = n1;
        if (k > -200) qv = ldexpf(1.0f, k);
        x = x < 0.0f ? -qv : qv;
        w[i] = x;
        mask[i] = 0;
        m = 0;
    }
    if (codes) {
        unsigned c = 255;
        if (!m) {
            const float a = fabsf(x);
            if (a == 0.0f) c = 0;
            else {
                const int k = (int)((__builtin_bit_cast(unsigned, a) >> 23) & 0xff) - 127;
                c = (k >= n2 && k <= n1) ? 1u + 2u * (unsigned)(k - n2) + (x < 0.0f ? 1u : 0u) : 254u;  // 254: frozen value outside the level set
            }
        }
        codes[i] = (uint8_t)c;
    }
}

// Effective weights of INQ layers: the latent parameter where it is still trainable, the frozen level elsewhere
// (wt = mask ? w : frozen).  The frozen values come decoded (the caller refreshes them once per INQ stage).
// SS: also the sum of squares of the latent weights (the layer's term of the l2 regulariser, which reads the same
// tensor), one partial per wavefront; inq_sumsq_finish_kernel adds a layer's partials in double, in ascending order.
template <bool SS>
__global__ __launch_bounds__(256) void inq_effective_kernel(const TernTab tab)
{
    const int si = tern_segment(tab);
    const TernSegDev &s = tab.s[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = ((int)blockIdx.x - s.blk0) * 4 + wave;
    if (wt >= s.nwave) return;
    const long long e0 = (long long)wt * TERN_ELEMS_PER_WAVE;
    const float *frozen = s.dwt;  // (the table's second input stream)
    const bool vec = ((((uintptr_t)s.w | (uintptr_t)frozen | (uintptr_t)s.wt) & 15) == 0) && (((uintptr_t)s.mask & 3) == 0);
    float acc = 0.0f;
    if (vec && e0 + TERN_ELEMS_PER_WAVE <= s.n) {  // whole wavefront inside the tensor: 8 + 8 + 8 loads in flight per lane
        const tern_f4 *pw = reinterpret_cast<const tern_f4 *>(s.w + e0) + lane;
        const tern_f4 *pf = reinterpret_cast<const tern_f4 *>(frozen + e0) + lane;
        const unsigned *pm = reinterpret_cast<const unsigned *>(s.mask + e0) + lane;
        tern_f4 *po = reinterpret_cast<tern_f4 *>(s.wt + e0) + lane;
        tern_f4 a[8], f[8];
        unsigned m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = pw[64 * u];
            f[u] = pf[64 * u];
            m[u] = pm[64 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            tern_f4 o;
            o.x = (m[u] & 0xffu) ? a[u].x : f[u].x;
            o.y = (m[u] & 0xff00u) ? a[u].y : f[u].y;
            o.z = (m[u] & 0xff0000u) ? a[u].z : f[u].z;
            o.w = (m[u] & 0xff000000u) ? a[u].w : f[u].w;
            po[64 * u] = o;
            if (SS) {
                acc = __builtin_fmaf(a[u].x, a[u].x, acc); acc = __builtin_fmaf(a[u].y, a[u].y, acc);
                acc = __builtin_fmaf(a[u].z, a[u].z, acc); acc = __builtin_fmaf(a[u].w, a[u].w, acc);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long e = e0 + 256 * u + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < s.n) {
                    const float w = s.w[e + k];
                    s.wt[e + k] = s.mask[e + k] ? w : frozen[e + k];
                    if (SS) acc = __builtin_fmaf(w, w, acc);
                }
        }
    }
    if (SS) {
        acc = wave_sum(acc);
        if (lane == 0) s.part[wt] = acc;
    }
}

__global__ __launch_bounds__(256) void inq_sumsq_finish_kernel(const TernTab tab, float *__restrict__ out)
{
    __shared__ double sh[256];
    const TernSegDev &s = tab.s[blockIdx.x];
    double a = 0.0;
    for (int k = threadIdx.x; k < s.nwave; k += 256) a += (double)s.part[k];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)sh[0];
}

// The quantiser's gradient rule with torch.optim.Adam's update of the latent weights applied in registers: the
// straight-through gradient (MODE as in tern_bwd_kernel) plus reg * w (the l2 regulariser's 2 * l2_reg * w) never
// travels through memory unless dw is asked for.  Every wavefront reads the step counter as the previous step left
// it; the second-stage kernel (one workgroup per layer, next launch on the stream) advances it.
constexpr int TERN_ADAM_MAX_SEGS = 16;  // per launch: the tables travel in the kernel arguments
struct TernAdamSeg {
    float *m, *v;
    const float *step;
};
struct TernAdamTab {
    TernTab t;
    TernAdamSeg a[TERN_ADAM_MAX_SEGS];
    double lr, b1, b2;
    float omb1, b2f, omb2, eps, reg;
};

template <int MODE>
__global__ __launch_bounds__(256) void tern_bwd_adam_kernel(const TernAdamTab tab)
{
    constexpr int U = 4;  // 16-byte loads per lane and stream, four streams: sixteen loads in flight per lane
    const int si = tern_segment(tab.t);
    const TernSegDev &s = tab.t.s[si];
    const TernAdamSeg &a = tab.a[si];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x - s.blk0;
    const int wt = blk * 4 + wave;
    const long long e0 = (long long)wt * (256 * U);
    float delta = 0.0f, pos = 1.0f, neg = 1.0f;
    if (MODE == 1) {
        delta = s.stats[0];
        pos = s.wp[0];
        neg = s.wn[0];
    }
    const float tnew = a.step[0] + 1.0f;
    const double bc1 = 1.0 - pow(tab.b1, (double)tnew), bc2 = 1.0 - pow(tab.b2, (double)tnew);
    const float step_size = (float)(tab.lr / bc1), rbc2 = (float)(1.0 / sqrt(bc2));
    float g[U][4], w[U][4], m0[U][4], v0[U][4];
    const bool vec = (((uintptr_t)s.dwt | (uintptr_t)s.w | (uintptr_t)a.m | (uintptr_t)a.v | (uintptr_t)s.dw) & 15) == 0;
    load_run<U, false>(s.dwt, e0, s.n, lane, g, vec);
    load_run<U, false>(s.w, e0, s.n, lane, w, vec);
    load_run<U, false>(a.m, e0, s.n, lane, m0, vec);
    load_run<U, false>(a.v, e0, s.n, lane, v0, vec);
    float sp = 0.0f, sn = 0.0f;
    float *wout = const_cast<float *>(s.w);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long e = e0 + 256 * u + 4 * lane;
        float o[4], pn[4], mn[4], vn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = g[u][k];
            const float x = w[u][k];
            if (MODE == 1) {
                if (x > delta) {
                    sp += d;
                    d *= pos;
                } else if (x < -delta) {
                    sn += d;
                    d *= neg;
                }
            } else if (MODE == 2) {
                d = (e + k < s.n && s.mask[e + k]) ? d : 0.0f;
            }
            if (tab.reg != 0.0f) d = __builtin_fmaf(tab.reg, x, d);
            o[k] = d;
            const float mm = __builtin_fmaf(tab.omb1, d - m0[u][k], m0[u][k]);
            const float vv = __builtin_fmaf(tab.omb2 * d, d, tab.b2f * v0[u][k]);
            const float denom = __builtin_fmaf(sqrtf(vv), rbc2, tab.eps);
            pn[k] = x - step_size * (mm / denom);
            mn[k] = mm;
            vn[k] = vv;
        }
        if (vec && e + 3 < s.n) {
            *reinterpret_cast<tern_f4 *>(wout + e) = tern_f4{pn[0], pn[1], pn[2], pn[3]};
            *reinterpret_cast<tern_f4 *>(a.m + e) = tern_f4{mn[0], mn[1], mn[2], mn[3]};
            *reinterpret_cast<tern_f4 *>(a.v + e) = tern_f4{vn[0], vn[1], vn[2], vn[3]};
            if (s.dw) *reinterpret_cast<tern_f4 *>(s.dw + e) = tern_f4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < s.n) {
                    wout[e + k] = pn[k];
                    a.m[e + k] = mn[k];
                    a.v[e + k] = vn[k];
                    if (s.dw) s.dw[e + k] = o[k];
                }
        }
    }
    if (MODE == 1) {
        const float vals[2] = {wave_sum(sp), wave_sum(sn)};
        wave_partials<2>(vals, s.part, wt, s.nwave);
    }
}

// Second stage of the fused form, one workgroup per layer: finishes d loss / d Wp, d loss / d Wn (TTQ), applies Adam to
// the two scales, and advances the step counters the first stage read.
struct TernScaleAdam {
    float *wp, *wn;                 // the learned scales (updated in place)
    float *pm, *pv, *pstep;         // Adam state of Wp
    float *nm, *nv, *nstep;         // Adam state of Wn
    float *wstep;                   // step counter of the latent weight
};
struct TernScaleTab {
    TernTab t;
    TernScaleAdam a[TERN_ADAM_MAX_SEGS];
    double lr, b1, b2;
    float omb1, b2f, omb2, eps;
    int ttq;
};
__device__ __forceinline__ void adam_scalar(float *p, float g, float *m, float *v, float *step, const TernScaleTab &tab)
{
    const float tnew = step[0] + 1.0f;
    const double bc1 = 1.0 - pow(tab.b1, (double)tnew), bc2 = 1.0 - pow(tab.b2, (double)tnew);
    const float step_size = (float)(tab.lr / bc1), rbc2 = (float)(1.0 / sqrt(bc2));
    const float mm = __builtin_fmaf(tab.omb1, g - m[0], m[0]);
    const float vv = __builtin_fmaf(tab.omb2 * g, g, tab.b2f * v[0]);
    const float denom = __builtin_fmaf(sqrtf(vv), rbc2, tab.eps);
    p[0] = p[0] - step_size * (mm / denom);
    m[0] = mm;
    v[0] = vv;
    step[0] = tnew;
}
__global__ __launch_bounds__(1024) void tern_scale_adam_kernel(const TernScaleTab tab)
{
    __shared__ float sh[1024];
    const TernSegDev &s = tab.t.s[blockIdx.x];
    const TernScaleAdam &a = tab.a[blockIdx.x];
    float sp = 0.0f, sn = 0.0f;
    if (tab.ttq) {
        int nblk;
        const float *pp = last_level(s, nblk);
        sp = block_sum_strided(pp, nblk, 2, sh);
        sn = block_sum_strided(pp + 1, nblk, 2, sh);
    }
    if (threadIdx.x == 0) {
        if (tab.ttq) {
            if (s.dwp) s.dwp[0] = sp;
            if (s.dwn) s.dwn[0] = -sn;
            adam_scalar(a.wp, sp, a.pm, a.pv, a.pstep, tab);
            adam_scalar(a.wn, -sn, a.nm, a.nv, a.nstep, tab);
        }
        a.wstep[0] = a.wstep[0] + 1.0f;
    }
}

// ------------------------------------------------------------------------------------------------- host side

static inline int tern_waves(long long n, int epw) { return (int)((n + epw - 1) / epw); }

static inline int tern_level2(long long n, int epw)
{
    const int nw = tern_waves(n, epw);
    return nw > 2048 ? (nw + LEVEL_CHUNK - 1) / LEVEL_CHUNK : 0;
}
// workspace of a layer: sized for the kernels with the most wavefronts (the two-input gradient kernel)
static inline size_t tern_seg_bytes(long long n)
{
    const int epw = TERN_ELEMS_PER_WAVE_2IN;
    return ((((size_t)tern_waves(n, epw) * 4 * sizeof(float)) + 255) & ~(size_t)255) + ((((size_t)tern_level2(n, epw) * 4 * sizeof(float)) + 255) & ~(size_t)255);
}
static size_t tern_ws_bytes(const aq_tern_seg *segs, int nseg)
{
    size_t off = 0;
    for (int i = 0; i < nseg; ++i) off += tern_seg_bytes(segs[i].n);
    return off;
}

static int tern_validate(const char *who, const aq_tern_seg *segs, int nseg)
{
    AQ_REQUIRE(segs && nseg >= 1 && nseg <= TERN_MAX_SEGS, -EINVAL, "%s: 1..%d layers per call (got %d)", who, TERN_MAX_SEGS, nseg);
    for (int i = 0; i < nseg; ++i) {
        AQ_REQUIRE(segs[i].n >= 1 && segs[i].n < (1LL << 40), -EINVAL, "%s: layer %d: bad element count", who, i);
        AQ_REQUIRE(segs[i].W && ((uintptr_t)segs[i].W & 3) == 0, -EFAULT, "%s: layer %d: W is NULL or misaligned", who, i);
    }
    return 0;
}

static int tern_table(const aq_tern_seg *segs, int nseg, void *workspace, TernTab &tab, int epw = TERN_ELEMS_PER_WAVE)
{
    tab.nseg = nseg;
    long long total = 0;
    for (int i = 0; i < nseg; ++i) total += segs[i].n;
    tab.nt = total * 4 > (128LL << 20) ? 1 : 0;
    size_t off = 0;
    long long blocks = 0;
    for (int i = 0; i < nseg; ++i) {
        TernSegDev &d = tab.s[i];
        const aq_tern_seg &s = segs[i];
        d.w = s.W; d.dwt = s.dWt; d.mask = s.mask; d.codes = s.codes; d.packed = s.packed; d.wt = s.Wt; d.dw = s.dW;
        d.stats = s.stats; d.q = s.q; d.wp = s.Wp; d.wn = s.Wn; d.dwp = s.dWp; d.dwn = s.dWn;
        d.n = s.n;
        d.nwave = tern_waves(s.n, epw);
        d.part = reinterpret_cast<float *>(static_cast<char *>(workspace) + off);
        d.n2 = tern_level2(s.n, epw);
        d.part2 = reinterpret_cast<float *>(static_cast<char *>(workspace) + off + ((((size_t)d.nwave * 4 * sizeof(float)) + 255) & ~(size_t)255));
        off += tern_seg_bytes(s.n);
        d.blk0 = (int)blocks;
        blocks += (d.nwave + 3) / 4;
        if (blocks >= (1LL << 31)) return -1;
    }
    return (int)blocks;
}

template <int F, int MAXF>
static void launch_levels(const TernTab &tab, hipStream_t st)
{
    for (int i = 0; i < tab.nseg; ++i)
        if (tab.s[i].n2)
            hipLaunchKernelGGL((tern_level_kernel<F, MAXF>), dim3(tab.s[i].n2), dim3(1024), 0, st, tab.s[i].part, tab.s[i].nwave, tab.s[i].part2);
}

// threads of the one-workgroup-per-layer second stages: 1024 once some layer has thousands of partials
static int second_stage_threads(const TernTab &tab)
{
    int most = 0;
    for (int i = 0; i < tab.nseg; ++i) most = std::max(most, tab.s[i].n2 ? tab.s[i].n2 : tab.s[i].nwave);
    return most > 2048 ? 1024 : 256;
}

} // namespace aq

extern "C" {

size_t aq_tern_workspace_bytes(const aq_tern_seg *segs, int nseg)
{
    if (!segs || nseg < 1 || nseg > aq::TERN_MAX_SEGS) return 0;
    return aq::tern_ws_bytes(segs, nseg);
}

int aq_tern_fwd(const aq_tern_seg *segs, int nseg, int mode, float factor, int threshold_from_max, void *workspace, size_t workspace_bytes,
                void *stream)
{
    using namespace aq;
    if (int rc = tern_validate("aq_tern_fwd", segs, nseg)) return rc;
    AQ_REQUIRE(mode == AQ_TERN_TWN || mode == AQ_TERN_TTQ, -EINVAL, "aq_tern_fwd: mode %d (AQ_TERN_TWN or AQ_TERN_TTQ)", mode);
    AQ_REQUIRE(factor >= 0.0f && factor == factor, -EDOM, "aq_tern_fwd: threshold factor %g", (double)factor);
    bool decode = false;
    for (int i = 0; i < nseg; ++i) {
        AQ_REQUIRE(segs[i].stats && segs[i].q, -EFAULT, "aq_tern_fwd: layer %d: stats / q is NULL", i);
        AQ_REQUIRE(mode != AQ_TERN_TTQ || (segs[i].Wp && segs[i].Wn), -EFAULT, "aq_tern_fwd: layer %d: TTQ needs Wp and Wn", i);
        AQ_REQUIRE(mode == AQ_TERN_TTQ || !segs[i].Wt || segs[i].codes, -EINVAL, "aq_tern_fwd: layer %d: TWN decodes Wt from codes: pass both", i);
        decode = decode || (mode == AQ_TERN_TWN && segs[i].Wt);
    }
    const size_t need = tern_ws_bytes(segs, nseg);
    AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, -ENOSPC,
               "aq_tern_fwd: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
    TernTab tab;
    const int blocks = tern_table(segs, nseg, workspace, tab);
    AQ_REQUIRE(blocks > 0, -E2BIG, "aq_tern_fwd: too many elements for one launch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (tab.nt) hipLaunchKernelGGL(tern_absstat_kernel<true>, dim3(blocks), dim3(256), 0, st, tab);
    else hipLaunchKernelGGL(tern_absstat_kernel<false>, dim3(blocks), dim3(256), 0, st, tab);
    launch_levels<2, 1>(tab, st);
    hipLaunchKernelGGL(tern_delta_kernel, dim3(nseg), dim3(second_stage_threads(tab)), 0, st, tab, factor, threshold_from_max ? 1 : 0);
    if (mode == AQ_TERN_TTQ) {
        if (tab.nt) { note_kernel("aq::tern_emit_kernel<true, true>"); hipLaunchKernelGGL((tern_emit_kernel<true, true>), dim3(blocks), dim3(256), 0, st, tab); }
        else { note_kernel("aq::tern_emit_kernel<true, false>"); hipLaunchKernelGGL((tern_emit_kernel<true, false>), dim3(blocks), dim3(256), 0, st, tab); }
        launch_levels<3, -1>(tab, st);
        hipLaunchKernelGGL(tern_codebook_kernel<true>, dim3(nseg), dim3(second_stage_threads(tab)), 0, st, tab);
    } else {
        if (tab.nt) { note_kernel("aq::tern_emit_kernel<false, true>"); hipLaunchKernelGGL((tern_emit_kernel<false, true>), dim3(blocks), dim3(256), 0, st, tab); }
        else { note_kernel("aq::tern_emit_kernel<false, false>"); hipLaunchKernelGGL((tern_emit_kernel<false, false>), dim3(blocks), dim3(256), 0, st, tab); }
        launch_levels<3, -1>(tab, st);
        hipLaunchKernelGGL(tern_codebook_kernel<false>, dim3(nseg), dim3(second_stage_threads(tab)), 0, st, tab);
        if (decode) hipLaunchKernelGGL(tern_decode_kernel, dim3(blocks), dim3(256), 0, st, tab);
    }
    return check_launch("aq_tern_fwd");
}

int aq_tern_bwd(const aq_tern_seg *segs, int nseg, int mode, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = tern_validate("aq_tern_bwd", segs, nseg)) return rc;
    AQ_REQUIRE(mode == AQ_TERN_TWN || mode == AQ_TERN_TTQ || mode == AQ_TERN_INQ, -EINVAL, "aq_tern_bwd: mode %d", mode);
    for (int i = 0; i < nseg; ++i) {
        AQ_REQUIRE(segs[i].dWt && segs[i].dW, -EFAULT, "aq_tern_bwd: layer %d: dWt / dW is NULL", i);
        AQ_REQUIRE(mode != AQ_TERN_TTQ || (segs[i].Wp && segs[i].Wn && segs[i].stats), -EFAULT, "aq_tern_bwd: layer %d: TTQ needs Wp, Wn and stats", i);
        AQ_REQUIRE(mode != AQ_TERN_INQ || segs[i].mask, -EFAULT, "aq_tern_bwd: layer %d: INQ needs the partition mask", i);
    }
    const size_t need = tern_ws_bytes(segs, nseg);
    AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, -ENOSPC,
               "aq_tern_bwd: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
    TernTab tab;
    const int blocks = tern_table(segs, nseg, workspace, tab, mode == AQ_TERN_TTQ ? TERN_ELEMS_PER_WAVE_2IN : TERN_ELEMS_PER_WAVE);
    AQ_REQUIRE(blocks > 0, -E2BIG, "aq_tern_bwd: too many elements for one launch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == AQ_TERN_TTQ) {
        if (tab.nt) { note_kernel("aq::tern_bwd_kernel<1, true>"); hipLaunchKernelGGL((tern_bwd_kernel<1, true>), dim3(blocks), dim3(256), 0, st, tab); }
        else { note_kernel("aq::tern_bwd_kernel<1, false>"); hipLaunchKernelGGL((tern_bwd_kernel<1, false>), dim3(blocks), dim3(256), 0, st, tab); }
        launch_levels<2, -1>(tab, st);
        hipLaunchKernelGGL(tern_scale_grad_kernel, dim3(nseg), dim3(second_stage_threads(tab)), 0, st, tab);
    } else if (mode == AQ_TERN_INQ) {
        if (tab.nt) { note_kernel("aq::tern_bwd_kernel<2, true>"); hipLaunchKernelGGL((tern_bwd_kernel<2, true>), dim3(blocks), dim3(256), 0, st, tab); }
        else { note_kernel("aq::tern_bwd_kernel<2, false>"); hipLaunchKernelGGL((tern_bwd_kernel<2, false>), dim3(blocks), dim3(256), 0, st, tab); }
    } else {
        if (tab.nt) { note_kernel("aq::tern_bwd_kernel<0, true>"); hipLaunchKernelGGL((tern_bwd_kernel<0, true>), dim3(blocks), dim3(256), 0, st, tab); }
        else { note_kernel("aq::tern_bwd_kernel<0, false>"); hipLaunchKernelGGL((tern_bwd_kernel<0, false>), dim3(blocks), dim3(256), 0, st, tab); }
    }
    return check_launch("aq_tern_bwd");
}

int aq_inq_partition(float *W, uint8_t *mask, uint8_t *codes, long long n, float tau, int n1, int n2, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && n < (1LL << 38), -EINVAL, "aq_inq_partition: bad element count");
    if (n == 0) return 0;
    AQ_REQUIRE(W && mask, -EFAULT, "aq_inq_partition: NULL pointer");
    AQ_REQUIRE(n1 >= n2 && n1 <= 126 && n2 >= -125 && n1 - n2 < 120, -EDOM, "aq_inq_partition: exponent range [%d, %d]", n2, n1);
    AQ_REQUIRE(tau >= 0.0f, -EDOM, "aq_inq_partition: negative threshold");
    hipLaunchKernelGGL(inq_partition_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), W, mask, codes, n, tau,
                       n1, n2);
    return check_launch("aq_inq_partition");
}

static int inq_effective_common(const char *who, const aq_tern_seg *segs, int nseg, float *sumsq, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = tern_validate(who, segs, nseg)) return rc;
    for (int i = 0; i < nseg; ++i)
        AQ_REQUIRE(segs[i].mask && segs[i].frozen && segs[i].Wt, -EFAULT, "%s: layer %d: mask / frozen / Wt is NULL", who, i);
    if (sumsq) {
        const size_t need = tern_ws_bytes(segs, nseg);
        AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, -ENOSPC,
                   "%s: workspace of %zu bytes (16-byte aligned) needed, %zu given", who, need, workspace_bytes);
    }
    TernTab tab;
    aq_tern_seg tmp[TERN_MAX_SEGS];
    for (int i = 0; i < nseg; ++i) {
        tmp[i] = segs[i];
        tmp[i].dWt = segs[i].frozen;  // the kernels' second input stream
    }
    const int blocks = tern_table(tmp, nseg, sumsq ? workspace : nullptr, tab);
    AQ_REQUIRE(blocks > 0, -E2BIG, "%s: too many elements for one launch", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (sumsq) {
        hipLaunchKernelGGL(inq_effective_kernel<true>, dim3(blocks), dim3(256), 0, st, tab);
        hipLaunchKernelGGL(inq_sumsq_finish_kernel, dim3(nseg), dim3(256), 0, st, tab, sumsq);
    } else {
        hipLaunchKernelGGL(inq_effective_kernel<false>, dim3(blocks), dim3(256), 0, st, tab);
    }
    return check_launch(who);
}

int aq_inq_effective(const aq_tern_seg *segs, int nseg, void *stream)
{
    return inq_effective_common("aq_inq_effective", segs, nseg, nullptr, nullptr, 0, stream);
}

int aq_inq_effective_sumsq(const aq_tern_seg *segs, int nseg, float *sumsq, void *workspace, size_t workspace_bytes, void *stream)
{
    AQ_REQUIRE(sumsq, -EFAULT, "aq_inq_effective_sumsq: NULL sumsq");
    return inq_effective_common("aq_inq_effective_sumsq", segs, nseg, sumsq, workspace, workspace_bytes, stream);
}

int aq_tern_bwd_adam(const aq_tern_seg *segs, int nseg, int mode, double lr, double beta1, double beta2, double eps, float reg, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    using namespace aq;
    if (int rc = tern_validate("aq_tern_bwd_adam", segs, nseg)) return rc;
    AQ_REQUIRE(mode == AQ_TERN_TWN || mode == AQ_TERN_TTQ || mode == AQ_TERN_INQ, -EINVAL, "aq_tern_bwd_adam: mode %d", mode);
    AQ_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, -EDOM, "aq_tern_bwd_adam: lr=%g betas=(%g, %g) eps=%g out of range",
               lr, beta1, beta2, eps);
    for (int i = 0; i < nseg; ++i) {
        const aq_tern_seg &q = segs[i];
        AQ_REQUIRE(q.dWt && q.adam_m && q.adam_v && q.adam_step, -EFAULT, "aq_tern_bwd_adam: layer %d: dWt / Adam state is NULL", i);
        AQ_REQUIRE(mode != AQ_TERN_TTQ || (q.Wp && q.Wn && q.stats && q.Wp_m && q.Wp_v && q.Wp_step && q.Wn_m && q.Wn_v && q.Wn_step), -EFAULT,
                   "aq_tern_bwd_adam: layer %d: TTQ needs Wp, Wn, stats and the Adam state of both scales", i);
        AQ_REQUIRE(mode != AQ_TERN_INQ || q.mask, -EFAULT, "aq_tern_bwd_adam: layer %d: INQ needs the partition mask", i);
        for (int k = 0; k < i; ++k)
            AQ_REQUIRE(segs[k].adam_step != q.adam_step, -EINVAL, "aq_tern_bwd_adam: layers %d and %d share a step counter", k, i);
    }
    const size_t need = tern_ws_bytes(segs, nseg);
    AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, -ENOSPC,
               "aq_tern_bwd_adam: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t off = 0;
    for (int base = 0; base < nseg; base += TERN_ADAM_MAX_SEGS) {
        const int n = std::min(TERN_ADAM_MAX_SEGS, nseg - base);
        TernAdamTab tab = {};
        const int blocks = tern_table(segs + base, n, static_cast<char *>(workspace) + off, tab.t, TERN_ELEMS_PER_WAVE_2IN);
        AQ_REQUIRE(blocks > 0, -E2BIG, "aq_tern_bwd_adam: too many elements for one launch");
        for (int i = 0; i < n; ++i) {
            off += tern_seg_bytes(segs[base + i].n);
            tab.a[i] = TernAdamSeg{segs[base + i].adam_m, segs[base + i].adam_v, segs[base + i].adam_step};
        }
        tab.lr = lr; tab.b1 = beta1; tab.b2 = beta2;
        tab.omb1 = (float)(1.0 - beta1); tab.b2f = (float)beta2; tab.omb2 = (float)(1.0 - beta2); tab.eps = (float)eps; tab.reg = reg;
        if (mode == AQ_TERN_TTQ) hipLaunchKernelGGL(tern_bwd_adam_kernel<1>, dim3(blocks), dim3(256), 0, st, tab);
        else if (mode == AQ_TERN_INQ) hipLaunchKernelGGL(tern_bwd_adam_kernel<2>, dim3(blocks), dim3(256), 0, st, tab);
        else hipLaunchKernelGGL(tern_bwd_adam_kernel<0>, dim3(blocks), dim3(256), 0, st, tab);
        if (mode == AQ_TERN_TTQ) launch_levels<2, -1>(tab.t, st);
        TernScaleTab sc = {};
        sc.t = tab.t;
        for (int i = 0; i < n; ++i) {
            const aq_tern_seg &q = segs[base + i];
            sc.a[i] = TernScaleAdam{const_cast<float *>(q.Wp), const_cast<float *>(q.Wn), q.Wp_m, q.Wp_v, q.Wp_step, q.Wn_m, q.Wn_v, q.Wn_step, q.adam_step};
        }
        sc.lr = lr; sc.b1 = beta1; sc.b2 = beta2;
        sc.omb1 = tab.omb1; sc.b2f = tab.b2f; sc.omb2 = tab.omb2; sc.eps = tab.eps;
        sc.ttq = mode == AQ_TERN_TTQ ? 1 : 0;
        hipLaunchKernelGGL(tern_scale_adam_kernel, dim3(n), dim3(second_stage_threads(tab.t)), 0, st, sc);
        if (int rc = check_launch("aq_tern_bwd_adam")) return rc;
    }
    return 0;
}

} // extern "C"
