// Batch normalisation over (N, H, W) of an NCHW fp32 tensor fused with the ReLU that follows it
// everywhere in the reference's ResNet (`F.relu(self.bnX(..))`, adp_qtz_fan.py:674-675, 690, 726,
// 730), forward and backward, with the residual gradient added in the backward output pass.
//
// Both directions are two launches over a (blocks per channel, channel) grid:
//   forward   stats:  every block reduces its slice of one channel to (mean, sum of squared
//                     deviations from that mean) -- two passes over a slice that stays in cache,
//                     so no E[x^2]-E[x]^2 cancellation;
//             apply:  every block merges the channel's partials (fixed order, one wavefront),
//                     derives scale = gamma*rstd, shift = beta - mean*scale and writes
//                     y = max(0, x*scale + shift); block 0 of the channel also stores mean / rstd and
//                     updates the running statistics;
//   backward  reduce: partial sums of g and g*xhat with g = dy*[y > 0];
//             apply:  dx = scale*(g - mean(g) - xhat*mean(g*xhat)) [+ res]; block 0 stores dgamma, dbeta.
// No atomics, fixed summation order: results are reproducible run to run.  The tensors of ResNet-20
// (<= 8.4 MB) stay in L2 / Infinity Cache between the two launches; the kernels are bandwidth- and
// launch-bound.
#include "aq_common.h"
#include "aq_bn_shared.h"
#include "../../include/aq_hip.h"

namespace aq {

typedef float vec4 __attribute__((ext_vector_type(4)));

template <int VEC> struct Vec;
template <> struct Vec<4> {
    vec4 v;
    __device__ static Vec load(const float *p) { return Vec{*reinterpret_cast<const vec4 *>(p)}; }
    __device__ void store(float *p) const { *reinterpret_cast<vec4 *>(p) = v; }
    __device__ float &operator[](int i) { return reinterpret_cast<float *>(&v)[i]; }
};
template <> struct Vec<1> {
    float v;
    __device__ static Vec load(const float *p) { return Vec{*p}; }
    __device__ void store(float *p) const { *p = v; }
    __device__ float &operator[](int) { return v; }
};

// sum over the 256 threads of a block, result in every thread; sh holds 4 floats
__device__ inline float block_sum(float v, float *sh)
{
    v = wave_sum(v);
    __syncthreads();  // protects sh against the previous use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// iterate over the vectors of channel c in images [n0, n1): f(offset into the tensor)
template <int VEC, typename F>
__device__ inline void for_slice(int n0, int n1, int c, int C, int HW, F f, int first_round = 0)
{
    const int per = HW / VEC;
    const int total = (n1 - n0) * per;
    for (int i = threadIdx.x + first_round * BN_T; i < total; i += BN_T) {
        const int n = i / per, k = i - n * per;
        f(((size_t)(n0 + n) * C + c) * HW + (size_t)k * VEC, n0 + n);
    }
}

// The streaming kernels first need per-channel numbers that one wavefront merges from partials (a memory
// round trip plus cross-lane sums); their own first loads do not depend on those, so the first PRE rounds of
// a thread's slice (all of it at the usual sizes) are requested before the merge and land during it.
constexpr int BN_PRE = 2;
template <int VEC>
__device__ inline bool slice_item(int round, int n0, int n1, int c, int C, int HW, size_t &o, int &n)
{
    const int per = HW / VEC;
    const int i = threadIdx.x + round * BN_T;
    if (i >= (n1 - n0) * per) return false;
    const int nn = i / per, k = i - nn * per;
    n = n0 + nn;
    o = ((size_t)n * C + c) * HW + (size_t)k * VEC;
    return true;
}

template <int VEC>
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float *__restrict__ x, float2 *__restrict__ part, BnGeom g)
{
    __shared__ float sh[4];
    const int c = blockIdx.y, s = blockIdx.x;
    const int n0 = s * g.npb, n1 = min(g.N, n0 + g.npb);
    const float cnt = (float)(n1 - n0) * (float)g.HW;
    float a = 0.0f;
    for_slice<VEC>(n0, n1, c, g.C, g.HW, [&](size_t o, int) {
        auto v = Vec<VEC>::load(x + o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a += v[j];
    });
    const float mean = block_sum(a, sh) / cnt;
    float m2 = 0.0f;
    for_slice<VEC>(n0, n1, c, g.C, g.HW, [&](size_t o, int) {
        auto v = Vec<VEC>::load(x + o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float d = v[j] - mean;
            m2 = __builtin_fmaf(d, d, m2);
        }
    });
    m2 = block_sum(m2, sh);
    if (threadIdx.x == 0) part[c * g.S + s] = make_float2(mean, m2);
}

struct BnApplyArgs {
    const float *x;
    float *y;
    BnStatsIn st;
    int eval;
    // sout != NULL: the kernel also writes the block's shortcut of the SAME input, s = ZeroPad(AvgPool2d(2)(x))
    // (adp_qtz_fan.py:669-671): [N, sCs, H/2, sW/2] with x's channel c at c + sfront, zeros elsewhere
    float *sout;
    int sW, sfront, sCs;
    // add != NULL (kernel instantiated with ADD): y = relu(batch_norm(x) + add), the tail of a post-activation residual
    // block (torchvision's BasicBlock: out = relu(bn2(conv2(..)) + shortcut))
    const float *add;
};

// first wavefront of a block: channel statistics -> sc[0] = scale, sc[1] = shift; block 0 of the channel
// also stores them for the backward pass and updates the running statistics
__device__ inline void bn_finalize(const BnApplyArgs &a, const BnGeom &g, int c, float *sc)
{
    if (threadIdx.x < 64) {
        float mean, rstd;
        if (a.eval) {
            mean = a.st.running_mean[c];
            rstd = 1.0f / sqrtf(a.st.running_var[c] + a.st.eps);
        } else {
            bn_channel_stats(a.st, g, c, threadIdx.x, blockIdx.x == 0, mean, rstd);
        }
        if (threadIdx.x == 0) scale_shift(a.st.gamma[c], a.st.beta[c], mean, rstd, sc[0], sc[1]);
    }
    __syncthreads();
}

template <int VEC, bool RELU, bool SHORT, bool ADD = false>
__global__ __launch_bounds__(BN_T) void bn_apply_kernel(BnApplyArgs a, BnGeom g)
{
    __shared__ float sc[2];
    const int c = blockIdx.y;
    const int n0 = blockIdx.x * g.npb2, n1 = min(g.N, n0 + g.npb2);
    if constexpr (SHORT) {
        // shortcut of the raw input: this block's images and channel, or (grid rows beyond C) one of the zero
        // channels in front of / behind them; summation order of torch's avg_pool2d
        const int H = g.HW / a.sW, Ho = H >> 1, Wo = a.sW >> 1;
        const int per = Ho * Wo, total = (n1 - n0) * per;
        if (c >= g.C) {
            const int pc = c - g.C, cs = pc < a.sfront ? pc : pc + g.C;
            for (int i = threadIdx.x; i < total; i += BN_T) {
                const int n = i / per, k = i - n * per;
                a.sout[((size_t)(n0 + n) * a.sCs + cs) * per + k] = 0.0f;
            }
            return;
        }
        for (int i = threadIdx.x; i < total; i += BN_T) {
            const int n = i / per, k = i - n * per;
            const int oy = k / Wo, ox = k - oy * Wo;
            const float *p = a.x + (((size_t)(n0 + n) * g.C + c) * H + 2 * oy) * a.sW + 2 * ox;
            a.sout[((size_t)(n0 + n) * a.sCs + c + a.sfront) * per + k] = (((p[0] + p[1]) + p[a.sW]) + p[a.sW + 1]) * 0.25f;
        }
    }
    Vec<VEC> pv[BN_PRE];
    size_t po[BN_PRE];
    bool pok[BN_PRE];
#pragma unroll
    for (int u = 0; u < BN_PRE; ++u) {
        int n;
        pok[u] = slice_item<VEC>(u, n0, n1, c, g.C, g.HW, po[u], n);
        if (pok[u]) pv[u] = Vec<VEC>::load(a.x + po[u]);
    }
    bn_finalize(a, g, c, sc);
    const float scale = sc[0], shift = sc[1];
    auto apply = [&](Vec<VEC> v, size_t o) {
        Vec<VEC> r;
        if (ADD) r = Vec<VEC>::load(a.add + o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = __builtin_fmaf(v[j], scale, shift);
            if (ADD) t += r[j];
            v[j] = RELU ? fmaxf(t, 0.0f) : t;
        }
        v.store(a.y + o);
    };
#pragma unroll
    for (int u = 0; u < BN_PRE; ++u)
        if (pok[u]) apply(pv[u], po[u]);
    for_slice<VEC>(n0, n1, c, g.C, g.HW, [&](size_t o, int) { apply(Vec<VEC>::load(a.x + o), o); }, BN_PRE);
}

// normalisation + ReLU + mean over the plane (the network's tail: relu(bnlast(x)) followed by an
// average pool over the whole 8x8 map, adp_qtz_fan.py:730-731) without writing the activated tensor:
// one wavefront per (image, channel), a.y receives [N, C] means.
template <bool RELU>
__global__ __launch_bounds__(BN_T) void bn_apply_pool_kernel(BnApplyArgs a, BnGeom g)
{
    __shared__ float sc[2];
    const int c = blockIdx.y;
    bn_finalize(a, g, c, sc);
    const float scale = sc[0], shift = sc[1];
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= g.N) return;
    const float *p = a.x + ((size_t)n * g.C + c) * g.HW;
    float acc = 0.0f;
    for (int i = lane; i < g.HW; i += 64) {
        const float t = __builtin_fmaf(p[i], scale, shift);
        acc += RELU ? fmaxf(t, 0.0f) : t;
    }
    acc = wave_sum(acc);
    if (lane == 0) a.y[(size_t)n * g.C + c] = acc / (float)g.HW;
}

struct BnBwdArgs {
    const float *dpool;  // when set: d loss / d pooled [N, C]; the gradient of every element of plane (n, c) is dpool[n, c] / HW
    const float *dy, *x, *res;
    float *dx;
    float2 *part;
    const float *gamma, *beta, *save_mean, *save_rstd;
    float *dgamma, *dbeta;
    // pool_W > 0: res is not a tensor of x's shape but d loss / d s for s = ZeroPad(AvgPool2d(2)(x)), the block's
    // shortcut (adp_qtz_fan.py:669-671), shape [N, pool_Cs, H/2, W/2] with x's channel c at c + pool_front; the
    // gradient it sends to x[n, c, y, x'] is a quarter of its element [n, c + pool_front, y/2, x'/2]
    int pool_W, pool_front, pool_Cs;
    // kernels instantiated with YM: the forward was y = relu(batch_norm(x) + add); the ReLU's mask is y > 0 (ymask = y), and
    // the masked gradient, which is also d loss / d add, is written to gout
    const float *ymask;
    float *gout;
};

template <int VEC, bool RELU, bool YM = false>
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(BnBwdArgs a, BnGeom g)
{
    __shared__ float sh[4];
    const int c = blockIdx.y, s = blockIdx.x;
    const int n0 = s * g.npb, n1 = min(g.N, n0 + g.npb);
    const float mean = a.save_mean[c], rstd = a.save_rstd[c];
    float scale, shift;
    scale_shift(a.gamma[c], a.beta[c], mean, rstd, scale, shift);
    float s1 = 0.0f, s2 = 0.0f;
    const float inv_hw = 1.0f / (float)g.HW;
    for_slice<VEC>(n0, n1, c, g.C, g.HW, [&](size_t o, int n) {
        Vec<VEC> d;
        if (a.dpool) {
            const float t = a.dpool[(size_t)n * g.C + c] * inv_hw;
#pragma unroll
            for (int j = 0; j < VEC; ++j) d[j] = t;
        } else {
            d = Vec<VEC>::load(a.dy + o);
        }
        auto v = Vec<VEC>::load(a.x + o);
        Vec<VEC> ym;
        if (YM) ym = Vec<VEC>::load(a.ymask + o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool live = YM ? ym[j] > 0.0f : (!RELU || __builtin_fmaf(v[j], scale, shift) > 0.0f);
            const float gq = live ? d[j] : 0.0f;
            s1 += gq;
            s2 = __builtin_fmaf(gq, (v[j] - mean) * rstd, s2);
        }
    });
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) a.part[c * g.S + s] = make_float2(s1, s2);
}

template <int VEC, bool RELU, bool RES, bool YM = false>
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(BnBwdArgs a, BnGeom g)
{
    __shared__ float sc[2];
    const int c = blockIdx.y;
    const int n0 = blockIdx.x * g.npb2, n1 = min(g.N, n0 + g.npb2);
    const float inv_hw = 1.0f / (float)g.HW;
    auto fetch = [&](size_t o, int n, Vec<VEC> &d, Vec<VEC> &v, Vec<VEC> &r) {
        if (a.dpool) {
            const float t = a.dpool[(size_t)n * g.C + c] * inv_hw;
#pragma unroll
            for (int j = 0; j < VEC; ++j) d[j] = t;
        } else {
            d = Vec<VEC>::load(a.dy + o);
        }
        v = Vec<VEC>::load(a.x + o);
        if (RES) {
            if (a.pool_W > 0) {
                const int k = (int)(o - ((size_t)n * g.C + c) * g.HW);
                const int y = k / a.pool_W, x0 = k - y * a.pool_W;
                const int Ho = (g.HW / a.pool_W) >> 1, Wo = a.pool_W >> 1;
                const float *src = a.res + (((size_t)n * a.pool_Cs + c + a.pool_front) * Ho + (y >> 1)) * Wo;
#pragma unroll
                for (int j = 0; j < VEC; ++j) r[j] = ((y >> 1) < Ho && ((x0 + j) >> 1) < Wo) ? 0.25f * src[(x0 + j) >> 1] : 0.0f;
            } else {
                r = Vec<VEC>::load(a.res + o);
            }
        }
        if (YM) r = Vec<VEC>::load(a.ymask + o);  // (never together with RES: r carries y)
    };
    Vec<VEC> pd[BN_PRE], pv[BN_PRE], pr[BN_PRE];
    size_t po[BN_PRE];
    bool pok[BN_PRE];
#pragma unroll
    for (int u = 0; u < BN_PRE; ++u) {
        int n;
        pok[u] = slice_item<VEC>(u, n0, n1, c, g.C, g.HW, po[u], n);
        if (pok[u]) fetch(po[u], n, pd[u], pv[u], pr[u]);
    }
    if (threadIdx.x < 64) {
        const int s = threadIdx.x;
        float2 p = make_float2(0.0f, 0.0f);
        if (g.P > 0) {  // sums left behind by the dgrad epilogue: P <= 512 per channel, lane l adds l, l + 64, ..
            float2 q[BN_CONV_PER_LANE];
#pragma unroll
            for (int u = 0; u < BN_CONV_PER_LANE; ++u) q[u] = s + 64 * u < g.P ? a.part[c * g.P + s + 64 * u] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int u = 0; u < BN_CONV_PER_LANE; ++u) {
                p.x += q[u].x;
                p.y += q[u].y;
            }
        } else if (s < g.S) {
            p = a.part[c * g.S + s];
        }
        const float s1 = wave_sum(p.x), s2 = wave_sum(p.y);
        if (threadIdx.x == 0) {
            sc[0] = s1;
            sc[1] = s2;
            if (blockIdx.x == 0) {
                if (a.dbeta) a.dbeta[c] = s1;
                if (a.dgamma) a.dgamma[c] = s2;
            }
        }
    }
    const float mean = a.save_mean[c], rstd = a.save_rstd[c];
    float scale, shift;
    scale_shift(a.gamma[c], a.beta[c], mean, rstd, scale, shift);
    __syncthreads();
    const float cnt = (float)g.N * (float)g.HW;
    const float k1 = sc[0] / cnt, k2 = sc[1] / cnt;
    auto apply = [&](Vec<VEC> d, Vec<VEC> v, Vec<VEC> r, size_t o) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool live = YM ? r[j] > 0.0f : (!RELU || __builtin_fmaf(v[j], scale, shift) > 0.0f);
            const float gq = live ? d[j] : 0.0f;
            const float xh = (v[j] - mean) * rstd;
            float t = scale * ((gq - k1) - xh * k2);
            if (RES) t += r[j];
            d[j] = t;
            if (YM) r[j] = gq;
        }
        d.store(a.dx + o);
        if (YM) r.store(a.gout + o);
    };
#pragma unroll
    for (int u = 0; u < BN_PRE; ++u)
        if (pok[u]) apply(pd[u], pv[u], pr[u], po[u]);
    for_slice<VEC>(n0, n1, c, g.C, g.HW, [&](size_t o, int n) {
        Vec<VEC> d, v, r;
        fetch(o, n, d, v, r);
        apply(d, v, r, o);
    }, BN_PRE);
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_bn_stats(const float *x, float2 *part, const BnGeom &g, hipStream_t st)
{
    if (g.HW % 4 == 0 && aligned16(x)) hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(g.S, g.C), dim3(BN_T), 0, st, x, part, g);
    else hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(g.S, g.C), dim3(BN_T), 0, st, x, part, g);
    return check_launch("bn statistics");
}

static int check_dims(const char *fn, int N, int C, int HW, bool batch_stats)
{
    AQ_REQUIRE(N >= 1 && C >= 1 && HW >= 1, -EINVAL, "%s: bad sizes N=%d C=%d HW=%d", fn, N, C, HW);
    AQ_REQUIRE(C <= 65535, -E2BIG, "%s: more than 65535 channels", fn);
    AQ_REQUIRE((long long)N * HW < (1LL << 31), -E2BIG, "%s: N*H*W too large", fn);
    AQ_REQUIRE(!batch_stats || (long long)N * HW >= 2, -EINVAL, "%s: batch statistics need more than one value per channel", fn);
    return 0;
}

} // namespace aq

extern "C" {

size_t aq_bn_workspace_bytes(int N, int C, int HW)
{
    if (N < 1 || C < 1 || HW < 1) return 0;
    const aq::BnGeom g = aq::bn_geom(N, C, HW);
    return (size_t)C * g.S * sizeof(float2);
}

static int bn_fwd_impl(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                       long long *num_batches_tracked, int N, int C, int HW, float eps, float momentum, unsigned flags, float *y,
                       float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes, const aq_conv_stats *cs, void *stream,
                       bool pool, float *sout = nullptr, int sW = 0, int sfront = 0, int sback = 0, const float *add = nullptr)
{
    using namespace aq;
    const bool eval = flags & AQ_BN_EVAL, relu = flags & AQ_BN_RELU;
    if (int rc = check_dims("aq_bn_relu_fwd", N, C, HW, !eval)) return rc;
    AQ_REQUIRE(x && gamma && beta && y, -EFAULT, "aq_bn_relu_fwd: NULL pointer");
    AQ_REQUIRE(eps >= 0.0f, -EINVAL, "aq_bn_relu_fwd: negative eps");
    AQ_REQUIRE((flags & ~(AQ_BN_RELU | AQ_BN_EVAL)) == 0, -EINVAL, "aq_bn_relu_fwd: unknown flags 0x%x", flags);
    BnGeom g = bn_geom(N, C, HW);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = HW % 4 == 0 && aligned16(x) && aligned16(y) && (!add || aligned16(add));
    if (eval) {
        AQ_REQUIRE(running_mean && running_var, -EFAULT, "aq_bn_relu_fwd: evaluation mode needs the running statistics");
    } else if (cs) {
        AQ_REQUIRE(save_mean && save_rstd, -EFAULT, "aq_bn_relu_fwd: NULL save_mean / save_rstd");
        AQ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), -EINVAL, "aq_bn_relu_fwd: running_mean and running_var go together");
        AQ_REQUIRE(cs->P >= 1 && cs->P <= 64 * BN_CONV_PER_LANE && cs->ntiles >= cs->P && cs->tile_px >= 1 &&
                       (long long)cs->ntiles * cs->tile_px == (long long)N * HW,
                   -EINVAL, "aq_bn_relu_fwd: convolution statistics (P=%d, %d tiles of %d pixels) do not describe %d x %d values per channel",
                   cs->P, cs->ntiles, cs->tile_px, N, HW);
        AQ_REQUIRE(workspace && workspace_bytes >= (size_t)C * cs->P * sizeof(float2), -ENOSPC,
                   "aq_bn_relu_fwd: convolution statistics take %zu bytes, %zu given", (size_t)C * cs->P * sizeof(float2), workspace_bytes);
        g.P = cs->P;
        g.pbase = cs->ntiles / cs->P;
        g.prem = cs->ntiles % cs->P;
        g.tile_px = cs->tile_px;
    } else {
        AQ_REQUIRE(save_mean && save_rstd, -EFAULT, "aq_bn_relu_fwd: NULL save_mean / save_rstd");
        AQ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), -EINVAL, "aq_bn_relu_fwd: running_mean and running_var go together");
        AQ_REQUIRE(workspace && workspace_bytes >= (size_t)C * g.S * sizeof(float2), -ENOSPC, "aq_bn_relu_fwd: workspace of %zu bytes needed, %zu given",
                   (size_t)C * g.S * sizeof(float2), workspace_bytes);
        launch_bn_stats(x, static_cast<float2 *>(workspace), g, st);
    }
    BnApplyArgs a{x, y, BnStatsIn{static_cast<const float2 *>(workspace), gamma, beta, running_mean, running_var, num_batches_tracked,
                                  save_mean, save_rstd, eps, momentum}, eval ? 1 : 0, sout, sW, sfront, C + sfront + sback, add};
    const dim3 grid(g.S2, sout ? C + sfront + sback : C);
    AQ_REQUIRE(grid.y <= 65535, -E2BIG, "aq_bn_relu_fwd: more than 65535 channels");
    if (pool) {
        const dim3 pgrid((N + 3) / 4, C);
        if (relu) hipLaunchKernelGGL((bn_apply_pool_kernel<true>), pgrid, dim3(BN_T), 0, st, a, g);
        else hipLaunchKernelGGL((bn_apply_pool_kernel<false>), pgrid, dim3(BN_T), 0, st, a, g);
        return check_launch("aq_bn_relu_pool_fwd");
    }
    if (add) {  // (with the ReLU; the entry point checks)
        if (vec) hipLaunchKernelGGL((bn_apply_kernel<4, true, false, true>), grid, dim3(BN_T), 0, st, a, g);
        else hipLaunchKernelGGL((bn_apply_kernel<1, true, false, true>), grid, dim3(BN_T), 0, st, a, g);
        return check_launch("aq_bn_add_relu_fwd");
    }
    if (sout) {  // the variant that also writes the shortcut is a kernel of its own: the common one stays lean
        if (vec && relu) hipLaunchKernelGGL((bn_apply_kernel<4, true, true>), grid, dim3(BN_T), 0, st, a, g);
        else if (vec) hipLaunchKernelGGL((bn_apply_kernel<4, false, true>), grid, dim3(BN_T), 0, st, a, g);
        else if (relu) hipLaunchKernelGGL((bn_apply_kernel<1, true, true>), grid, dim3(BN_T), 0, st, a, g);
        else hipLaunchKernelGGL((bn_apply_kernel<1, false, true>), grid, dim3(BN_T), 0, st, a, g);
    } else if (vec && relu) hipLaunchKernelGGL((bn_apply_kernel<4, true, false>), grid, dim3(BN_T), 0, st, a, g);
    else if (vec) hipLaunchKernelGGL((bn_apply_kernel<4, false, false>), grid, dim3(BN_T), 0, st, a, g);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<1, true, false>), grid, dim3(BN_T), 0, st, a, g);
    else hipLaunchKernelGGL((bn_apply_kernel<1, false, false>), grid, dim3(BN_T), 0, st, a, g);
    return check_launch("aq_bn_relu_fwd");
}

int aq_bn_relu_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                   long long *num_batches_tracked, int N, int C, int HW, float eps, float momentum, unsigned flags, float *y,
                   float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats, void *stream)
{
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, N, C, HW, eps, momentum, flags, y, save_mean, save_rstd,
                       workspace, workspace_bytes, conv_stats, stream, false);
}

int aq_bn_relu_fwd_shortcut(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                            long long *num_batches_tracked, int N, int C, int H, int Wd, float eps, float momentum, unsigned flags, float *y,
                            float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats,
                            int pad_front, int pad_back, float *shortcut, void *stream)
{
    AQ_REQUIRE(shortcut, -EFAULT, "aq_bn_relu_fwd_shortcut: NULL shortcut");
    AQ_REQUIRE(H >= 2 && Wd >= 2 && pad_front >= 0 && pad_back >= 0 && (long long)H * Wd < (1LL << 31), -EINVAL,
               "aq_bn_relu_fwd_shortcut: bad sizes H=%d W=%d pad=%d,%d", H, Wd, pad_front, pad_back);
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, N, C, H * Wd, eps, momentum, flags, y, save_mean, save_rstd,
                       workspace, workspace_bytes, conv_stats, stream, false, shortcut, Wd, pad_front, pad_back);
}

int aq_bn_relu_pool_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                        long long *num_batches_tracked, int N, int C, int HW, float eps, float momentum, unsigned flags, float *pooled,
                        float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats,
                        void *stream)
{
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, N, C, HW, eps, momentum, flags, pooled, save_mean,
                       save_rstd, workspace, workspace_bytes, conv_stats, stream, true);
}

static int bn_bwd_impl(const float *dy, const float *dpool, const float *x, const float *gamma, const float *beta, const float *save_mean,
                       const float *save_rstd, const float *res, int N, int C, int HW, unsigned flags, float *dx, float *dgamma, float *dbeta,
                       void *workspace, size_t workspace_bytes, const aq_conv_stats *cs, void *stream, int pool_W = 0, int pool_front = 0,
                       int pool_Cs = 0, const float *ymask = nullptr, float *gout = nullptr)
{
    using namespace aq;
    if (int rc = check_dims("aq_bn_relu_bwd", N, C, HW, true)) return rc;
    AQ_REQUIRE((dy || dpool) && x && gamma && beta && save_mean && save_rstd && dx, -EFAULT, "aq_bn_relu_bwd: NULL pointer");
    AQ_REQUIRE((flags & ~AQ_BN_RELU) == 0, -EINVAL, "aq_bn_relu_bwd: unsupported flags 0x%x (batch-statistics mode only)", flags);
    BnGeom g = bn_geom(N, C, HW);
    if (cs) {
        AQ_REQUIRE(flags & AQ_BN_RELU, -EINVAL, "aq_bn_relu_bwd: the dgrad epilogue reduces for the ReLU form only");
        AQ_REQUIRE(cs->P >= 1 && cs->P <= 64 * BN_CONV_PER_LANE && (long long)cs->ntiles * cs->tile_px == (long long)N * HW, -EINVAL,
                   "aq_bn_relu_bwd: dgrad sums (P=%d, %d tiles of %d pixels) do not describe %d x %d values per channel", cs->P, cs->ntiles,
                   cs->tile_px, N, HW);
        AQ_REQUIRE(workspace && workspace_bytes >= (size_t)C * cs->P * sizeof(float2), -ENOSPC, "aq_bn_relu_bwd: dgrad sums take %zu bytes, %zu given",
                   (size_t)C * cs->P * sizeof(float2), workspace_bytes);
        g.P = cs->P;
    } else {
        AQ_REQUIRE(workspace && workspace_bytes >= (size_t)C * g.S * sizeof(float2), -ENOSPC, "aq_bn_relu_bwd: workspace of %zu bytes needed, %zu given",
                   (size_t)C * g.S * sizeof(float2), workspace_bytes);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool relu = flags & AQ_BN_RELU;
    // (a vector of four stays inside one image row when the pooled shortcut gradient is addressed)
    const bool vec = HW % 4 == 0 && aligned16(x) && (!dy || aligned16(dy)) && aligned16(dx) && (!res || pool_W > 0 || aligned16(res)) &&
                     (pool_W == 0 || pool_W % 4 == 0) && (!ymask || (aligned16(ymask) && aligned16(gout)));
    BnBwdArgs a{dpool, dy, x, res, dx, static_cast<float2 *>(workspace), gamma, beta, save_mean, save_rstd, dgamma, dbeta, pool_W, pool_front, pool_Cs,
                ymask, gout};
    const dim3 g1(g.S, C), g2(g.S2, C);
    if (ymask) {
        if (vec) {
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true, true>), g1, dim3(BN_T), 0, st, a, g);
            hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true, false, true>), g2, dim3(BN_T), 0, st, a, g);
        } else {
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, true, true>), g1, dim3(BN_T), 0, st, a, g);
            hipLaunchKernelGGL((bn_bwd_apply_kernel<1, true, false, true>), g2, dim3(BN_T), 0, st, a, g);
        }
        return check_launch("aq_bn_add_relu_bwd");
    }
#define AQ_BN_BWD(V, R)                                                                                        \
    {                                                                                                          \
        if (!cs) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, R>), g1, dim3(BN_T), 0, st, a, g);                \
        if (res) hipLaunchKernelGGL((bn_bwd_apply_kernel<V, R, true>), g2, dim3(BN_T), 0, st, a, g);           \
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<V, R, false>), g2, dim3(BN_T), 0, st, a, g);              \
    }
    if (vec && relu) AQ_BN_BWD(4, true)
    else if (vec) AQ_BN_BWD(4, false)
    else if (relu) AQ_BN_BWD(1, true)
    else AQ_BN_BWD(1, false)
#undef AQ_BN_BWD
    return check_launch("aq_bn_relu_bwd");
}

int aq_bn_relu_bwd(const float *dy, const float *x, const float *gamma, const float *beta, const float *save_mean,
                   const float *save_rstd, const float *res, int N, int C, int HW, unsigned flags, float *dx, float *dgamma,
                   float *dbeta, void *workspace, size_t workspace_bytes, const aq_conv_stats *dgrad_sums, void *stream)
{
    AQ_REQUIRE(dy, -EFAULT, "aq_bn_relu_bwd: NULL dy");
    return bn_bwd_impl(dy, nullptr, x, gamma, beta, save_mean, save_rstd, res, N, C, HW, flags, dx, dgamma, dbeta, workspace, workspace_bytes,
                       dgrad_sums, stream);
}

int aq_bn_relu_bwd_shortcut(const float *dy, const float *x, const float *gamma, const float *beta, const float *save_mean, const float *save_rstd,
                            const float *dshort, int N, int C, int H, int Wd, int pad_front, int pad_back, unsigned flags, float *dx, float *dgamma,
                            float *dbeta, void *workspace, size_t workspace_bytes, const aq_conv_stats *dgrad_sums, void *stream)
{
    AQ_REQUIRE(dy && dshort, -EFAULT, "aq_bn_relu_bwd_shortcut: NULL pointer");
    AQ_REQUIRE(H >= 2 && Wd >= 2 && pad_front >= 0 && pad_back >= 0 && (long long)H * Wd < (1LL << 31), -EINVAL,
               "aq_bn_relu_bwd_shortcut: bad sizes H=%d W=%d pad=%d,%d", H, Wd, pad_front, pad_back);
    return bn_bwd_impl(dy, nullptr, x, gamma, beta, save_mean, save_rstd, dshort, N, C, H * Wd, flags, dx, dgamma, dbeta, workspace, workspace_bytes,
                       dgrad_sums, stream, Wd, pad_front, C + pad_front + pad_back);
}

int aq_bn_add_relu_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                       long long *num_batches_tracked, int N, int C, int HW, float eps, float momentum, const float *add, float *y,
                       float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats, void *stream)
{
    AQ_REQUIRE(add, -EFAULT, "aq_bn_add_relu_fwd: NULL add");
    return bn_fwd_impl(x, gamma, beta, running_mean, running_var, num_batches_tracked, N, C, HW, eps, momentum, AQ_BN_RELU, y, save_mean, save_rstd,
                       workspace, workspace_bytes, conv_stats, stream, false, nullptr, 0, 0, 0, add);
}

int aq_bn_add_relu_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *beta, const float *save_mean,
                       const float *save_rstd, int N, int C, int HW, float *dx, float *dadd, float *dgamma, float *dbeta, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    AQ_REQUIRE(dy && y && dadd, -EFAULT, "aq_bn_add_relu_bwd: NULL pointer");
    return bn_bwd_impl(dy, nullptr, x, gamma, beta, save_mean, save_rstd, nullptr, N, C, HW, AQ_BN_RELU, dx, dgamma, dbeta, workspace, workspace_bytes,
                       nullptr, stream, 0, 0, 0, y, dadd);
}

int aq_bn_relu_pool_bwd(const float *dpooled, const float *x, const float *gamma, const float *beta, const float *save_mean,
                        const float *save_rstd, int N, int C, int HW, unsigned flags, float *dx, float *dgamma, float *dbeta,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    AQ_REQUIRE(dpooled, -EFAULT, "aq_bn_relu_pool_bwd: NULL dpooled");
    return bn_bwd_impl(nullptr, dpooled, x, gamma, beta, save_mean, save_rstd, nullptr, N, C, HW, flags, dx, dgamma, dbeta, workspace,
                       workspace_bytes, nullptr, stream);
}

} // extern "C"
