// Batch-norm pieces shared by the stand-alone kernels (aq_bn.hip) and by the convolution that applies the
// normalisation of its input while staging it (aq_conv.hip): slice geometry of the statistics pass, the
// fixed-order merge of the per-slice (mean, M2) pairs, and scale / shift.  Both users run the SAME
// operation sequence, so a fused and an unfused evaluation give identical bits.
#pragma once
#include <hip/hip_runtime.h>

namespace aq {

constexpr int BN_T = 256;
constexpr int BN_MAX_PARTS = 64;  // partial blocks per channel (merged by one wavefront)

struct BnGeom {
    int N, C, HW;
    int S, npb;    // reduction launches: S blocks per channel, npb images per block
    int S2, npb2;  // streaming launches
    // statistics left behind by the producing convolution instead of the statistics pass (P > 0): partial
    // w of P covers pbase (+1 when w < prem) tiles of tile_px pixels
    int P, pbase, prem, tile_px;
};

inline BnGeom bn_geom(int N, int C, int HW)
{
    BnGeom g{N, C, HW, 1, 1, 1, 1, 0, 0, 0, 0};
    int npb = HW >= 2048 ? 1 : 2048 / HW;
    g.npb2 = npb;
    g.S2 = (N + npb - 1) / npb;
    const int cap = (N + BN_MAX_PARTS - 1) / BN_MAX_PARTS;
    g.npb = npb > cap ? npb : cap;
    g.S = (N + g.npb - 1) / g.npb;
    return g;
}

// what the consumer of batch statistics needs: the partials of the statistics pass and the module state
struct BnStatsIn {
    const float2 *part;  // [C][S] (mean, M2) per slice
    const float *gamma, *beta;
    float *running_mean, *running_var;      // may be NULL
    long long *num_batches_tracked;         // may be NULL
    float *save_mean, *save_rstd;           // [C], written by the designated wavefront
    float eps, momentum;
};

// Sum over the 64 lanes of a wavefront (all lanes must be active), the same bits in every lane: four DPP
// steps leave each row of 16 lanes with its sum (quad xor 1, xor 2, mirror inside 8, mirror inside 16), the
// four row sums are read as scalars and added as (r0 + r1) + (r2 + r3).  A dozen cheap instructions; the
// equivalent chain of six ds_bpermute shuffles costs several hundred cycles of latency.
template <int CTRL>
__device__ __forceinline__ float bn_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ inline float wave_sum(float v)
{
    v = v + bn_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + bn_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + bn_dpp<0x141>(v);  // row_half_mirror
    v = v + bn_dpp<0x140>(v);  // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ inline void scale_shift(float gamma, float beta, float mean, float rstd, float &scale, float &shift)
{
    scale = gamma * rstd;
    shift = __builtin_fmaf(-mean, scale, beta);
}

// Merge of the S <= 64 per-slice partials of NC channels by one wavefront (all 64 lanes active; lane s
// holds slice s; c[u] < 0: skip):
//   mean = sum n_s mean_s / n,   M2 = sum M2_s + sum n_s (mean_s - mean)^2,   rstd = rsq(M2 / n + eps)
// -> mean, rstd in every lane.  Written for few instructions, because the convolution runs it in every
// wavefront: the divisions by n are multiplications by 1/n computed once, the reciprocal square root is the
// hardware's (1 ulp), all loads are issued before the first reduction.  `writer`: this wavefront also
// stores mean / rstd for the backward pass and updates the running statistics (exactly one wavefront per
// channel and launch must do so).
template <int NC>
__device__ inline void bn_channel_stats_multi(const BnStatsIn &a, const BnGeom &g, const int (&c)[NC], int lane, bool writer,
                                              float (&mean)[NC], float (&rstd)[NC])
{
    const bool on = lane < g.S;
    const float ns = on ? (float)(min(g.N, (lane + 1) * g.npb) - lane * g.npb) * (float)g.HW : 0.0f;
    const float cnt = (float)g.N * (float)g.HW;
    const float rcnt = 1.0f / cnt;
    float2 p[NC];
    float M2[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) p[u] = (on && c[u] >= 0) ? a.part[c[u] * g.S + lane] : make_float2(0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        mean[u] = wave_sum(ns * p[u].x) * rcnt;
        const float d = p[u].x - mean[u];
        M2[u] = wave_sum(__builtin_fmaf(ns * d, d, p[u].y));
        rstd[u] = __builtin_amdgcn_rsqf(__builtin_fmaf(M2[u], rcnt, a.eps));
    }
    if (writer && lane == 0) {
        const float runb = 1.0f / fmaxf(cnt - 1.0f, 1.0f);
        for (int u = 0; u < NC; ++u) {
            if (c[u] < 0) continue;
            a.save_mean[c[u]] = mean[u];
            a.save_rstd[c[u]] = rstd[u];
            if (a.running_mean) {
                a.running_mean[c[u]] = __builtin_fmaf(a.momentum, mean[u] - a.running_mean[c[u]], a.running_mean[c[u]]);
                a.running_var[c[u]] = __builtin_fmaf(a.momentum, M2[u] * runb - a.running_var[c[u]], a.running_var[c[u]]);
            }
            if (c[u] == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
        }
    }
}

// The same merge over the P <= 512 partials of a convolution's epilogue (part[c * P + w], counts from the tile
// distribution): lane l holds partials l, l + 64, ..; all loads are issued before the first reduction.
constexpr int BN_CONV_PER_LANE = 8;
__device__ inline void bn_channel_stats_conv(const BnStatsIn &a, const BnGeom &g, int c, int lane, bool writer, float &mean, float &rstd)
{
    float2 p[BN_CONV_PER_LANE];
    float nw[BN_CONV_PER_LANE];
#pragma unroll
    for (int u = 0; u < BN_CONV_PER_LANE; ++u) {
        const int w = lane + 64 * u;
        const bool on = w < g.P;
        p[u] = on ? a.part[c * g.P + w] : make_float2(0.0f, 0.0f);
        nw[u] = on ? (float)((g.pbase + (w < g.prem ? 1 : 0)) * g.tile_px) : 0.0f;
    }
    const float cnt = (float)g.N * (float)g.HW;
    const float rcnt = 1.0f / cnt;
    float sm = 0.0f;
#pragma unroll
    for (int u = 0; u < BN_CONV_PER_LANE; ++u) sm = __builtin_fmaf(nw[u], p[u].x, sm);
    mean = wave_sum(sm) * rcnt;
    float m2 = 0.0f;
#pragma unroll
    for (int u = 0; u < BN_CONV_PER_LANE; ++u) {
        const float d = p[u].x - mean;
        m2 += __builtin_fmaf(nw[u] * d, d, p[u].y);
    }
    const float M2 = wave_sum(m2);
    rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(M2, rcnt, a.eps));
    if (writer && lane == 0) {
        const float runb = 1.0f / fmaxf(cnt - 1.0f, 1.0f);
        a.save_mean[c] = mean;
        a.save_rstd[c] = rstd;
        if (a.running_mean) {
            a.running_mean[c] = __builtin_fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
            a.running_var[c] = __builtin_fmaf(a.momentum, M2 * runb - a.running_var[c], a.running_var[c]);
        }
        if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
    }
}

__device__ inline void bn_channel_stats(const BnStatsIn &a, const BnGeom &g, int c, int lane, bool writer, float &mean, float &rstd)
{
    if (g.P > 0) {
        bn_channel_stats_conv(a, g, c, lane, writer, mean, rstd);
        return;
    }
    const int cs[1] = {c};
    float m[1], r[1];
    bn_channel_stats_multi<1>(a, g, cs, lane, writer, m, r);
    mean = m[0];
    rstd = r[0];
}

// statistics pass (aq_bn.hip): part[c * S + s] = (mean, M2) of slice s of channel c
int launch_bn_stats(const float *x, float2 *part, const BnGeom &g, hipStream_t st);

} // namespace aq
