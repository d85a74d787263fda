// Device-side arithmetic of the weight synthesis: one fixed sequence of IEEE-754 binary32
// operations per weight row, so that results are reproducible bit for bit (the file that
// includes this header is compiled with -ffp-contract=off).
//
// Layout of a row of L primary coefficients over G = 8, 16 or 32 consecutive lanes of a
// wavefront (the smallest G with 4*G >= L): lane j owns elements 4j..4j+3, elements >= L are
// zero padding.  Per-lane partial sums are left-to-right chains; lane partials are combined with
// an xor butterfly over the G lanes (a balanced binary tree in lane order), done with DPP so no
// LDS is touched.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace aq {

// exp(x), about 1 ulp, valid for |x| <= 64 (callers bound |beta|): x = q*ln2 + s with a
// two-constant Cody-Waite reduction, degree-5 polynomial for expm1(s)-s, scaling by 2^q.
// NaN in -> NaN out.  No range checks: inside the domain 2^q*u neither overflows nor underflows.
__device__ __forceinline__ float expf_spec(float x)
{
    const float R_LN2 = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2U = 0.693145751953125f;
    const float L2L = 1.428606765330187045e-06f;
    const float qf = __builtin_rintf(x * R_LN2);
    float s = __builtin_fmaf(qf, -L2U, x);
    s = __builtin_fmaf(qf, -L2L, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    return __builtin_amdgcn_ldexpf(u, (int)qf);
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each 16 lanes

// Sum over the G lanes that share a row; every lane of the group ends with the same bits.
// After the two quad steps all four lanes of a quad agree, so the mirror steps add exactly the
// partner quad / partner half that an xor-4 / xor-8 butterfly step would.
template <int G>
__device__ __forceinline__ float group_sum(float v)
{
    v = v + dpp<DPP_XOR1>(v);
    v = v + dpp<DPP_XOR2>(v);
    v = v + dpp<DPP_HALF_MIRROR>(v);
    if (G >= 16) v = v + dpp<DPP_MIRROR>(v);
    if (G >= 32) v = v + __shfl_xor(v, 16);
    return v;
}

// (value, index) of the largest e among the G lanes: larger value wins, equal values -> lower
// index (torch.max semantics, adp_qtz_fan_test.py:304).  e >= 0 and never NaN here, so its bit
// pattern orders like an unsigned integer.
template <int G>
__device__ __forceinline__ void group_argmax(uint32_t &key, uint32_t &idx)
{
#define AQ_AMAX_STEP(MOVE_K, MOVE_I)                             \
    {                                                            \
        const uint32_t ok = MOVE_K, oi = MOVE_I;                 \
        const bool take = (ok > key) || (ok == key && oi < idx); \
        key = take ? ok : key;                                   \
        idx = take ? oi : idx;                                   \
    }
    AQ_AMAX_STEP(dpp_u<DPP_XOR1>(key), dpp_u<DPP_XOR1>(idx))
    AQ_AMAX_STEP(dpp_u<DPP_XOR2>(key), dpp_u<DPP_XOR2>(idx))
    AQ_AMAX_STEP(dpp_u<DPP_HALF_MIRROR>(key), dpp_u<DPP_HALF_MIRROR>(idx))
    if (G >= 16) AQ_AMAX_STEP(dpp_u<DPP_MIRROR>(key), dpp_u<DPP_MIRROR>(idx))
    if (G >= 32) AQ_AMAX_STEP((uint32_t)__shfl_xor((int)key, 16), (uint32_t)__shfl_xor((int)idx, 16))
#undef AQ_AMAX_STEP
}

// Value held by lane U of each G-lane group, delivered to all lanes of the group (LDS crossbar,
// no LDS memory): ds_swizzle bit-mask mode, lane' = (lane & ~(G-1)) | U inside each 32 lanes.
template <int G, int U>
__device__ __forceinline__ float group_bcast(float v)
{
    constexpr int AND = 0x1F & ~(G - 1), OR = U;
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (OR << 5) | AND));
}

// compile-time loop: f(std::integral_constant<int, 0>{}), .., f(<N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// ---- the row-local chain, split into the pieces the kernels interleave over 8 rows -----------
__device__ __forceinline__ float chain_sq(const float (&p)[4])
{
    float t = p[0] * p[0];
    t = __builtin_fmaf(p[1], p[1], t);
    t = __builtin_fmaf(p[2], p[2], t);
    return __builtin_fmaf(p[3], p[3], t);
}
__device__ __forceinline__ float chain_dot(const float (&a)[4], const float (&b)[4])
{
    float t = a[0] * b[0];
    t = __builtin_fmaf(a[1], b[1], t);
    t = __builtin_fmaf(a[2], b[2], t);
    return __builtin_fmaf(a[3], b[3], t);
}
__device__ __forceinline__ float inv_norm(float ss) { return 1.0f / __builtin_sqrtf(ss); }

// Fast arithmetic (default): exp(x) = 2^(x*log2 e) on the transcendental unit (v_exp_f32, about
// 1 ulp, plus |x|*2^-24 from the scaled argument).  Not reproducible on a CPU, so results are
// compared with the oracle within a tolerance; the EXACT kernels use expf_spec instead.
constexpr float LOG2E = 1.442695040888963407359924681001892137426645954152985934135449406931f;
__device__ __forceinline__ float exp2_hw(float x2) { return __builtin_amdgcn_exp2f(x2); }

} // namespace aq
