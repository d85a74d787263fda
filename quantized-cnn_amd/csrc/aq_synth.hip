// Weight synthesis ("adaptive quantisation") kernels for gfx950 and their C-ABI entry points.
//
//   forward  : W = sum_l softmax(beta * p_c/||p_c||)_l * q_l   (adp_qtz_fan.py:563-568)
//              idx = argmax_l s_c, W_hard = q[idx]             (adp_qtz_fan_test.py:304-305)
//              sumsq = sum_rows W^2                            (adp_qtz_fan.py:752)
//   backward : dp_c, dq for g = dW + wd*W                      (autograd of the above)
//
// Both are HBM-bound streaming kernels over p_c (L floats per weight):
//   - a wavefront reads 64 consecutive rows per iteration as G-lane groups of 16-byte loads, so
//     every load instruction covers 1 KiB of contiguous memory; 8 loads per lane are issued
//     before the first is consumed;
//   - all per-row reductions stay inside the G lanes of a row (DPP, no LDS, no barrier);
//   - after the 64 rows are done, lane (g, j) keeps the result of row j*(64/G)+g, so W / idx are
//     written with one fully coalesced store per iteration;
//   - sums over rows (sum W^2, dq) are accumulated per lane over K iterations, reduced inside the
//     wavefront and written as one partial per wavefront; a second tiny kernel adds the partials
//     of each layer in a fixed order.  No atomics: results are run-to-run deterministic.
//   - several layers ("segments") share one launch through a table passed by value in kernarg.
#include "aq_arith.h"
#include "aq_common.h"
#include "../../include/aq_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace aq {

constexpr int MAX_SEGS = 24;  // segments per launch (kernarg budget)
constexpr int WAVES = 4;      // wavefronts per workgroup
constexpr int ROWS_IT = 64;   // rows per wavefront iteration
constexpr int STAGE_FLOATS = 260;  // LDS per wavefront for store_rows_staged: 256 floats of rows + the spill of the last ragged lane

struct SegDev {
    const float *p, *q, *dW, *dss;
    float *W, *Wh, *dp, *part;
    uint8_t *idx;
    float *am, *av;        // Adam first / second moment of p (fused update only)
    const float *acoef;    // {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} of this segment (fused update only)
    long long rows;
    int L, blk0;
    int rag;  // rows whose length is no multiple of 4, all output bases 16-byte aligned: stores go through LDS (store_rows_staged)
};
struct AdamArgs {
    float omb1, b2, omb2, eps;  // 1-beta1, beta2, 1-beta2, eps
};
struct SegTab {
    int nseg, K;
    SegDev s[MAX_SEGS];
};

// MEM: 0 = rows only 4-byte aligned (L != 4G or unaligned base), 1 = 16-byte vector access,
// 2 = 16-byte vector access with the nontemporal (streaming) cache policy, chosen by the host for
// tensors larger than the last-level cache: +7 % (forward) / +10 % (backward) at 1 GB; 3 = the rows of 0 with the
// streaming policy.  The policy is a template parameter, never a run-time flag: two branches that differ only in the
// nontemporal marking are merged by the compiler into one plain access.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4_a4 __attribute__((aligned(4)));  // a 16-byte access at a 4-byte-aligned address
constexpr bool mem_vec(int mem) { return mem == 1 || mem == 2; }
constexpr bool mem_nt(int mem) { return mem >= 2; }

template <int G, int MEM>
__device__ __forceinline__ void load_elems(const float *__restrict__ p, long long r, int L, int j, float (&v)[4])
{
    if constexpr (mem_vec(MEM)) {
        const f4 *src = reinterpret_cast<const f4 *>(p) + (r * G + j);
        const f4 t = mem_nt(MEM) ? __builtin_nontemporal_load(src) : *src;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const float *a = p + r * (long long)L + 4 * j;
        if (4 * j + 3 < L) {
            const f4_a4 *src = reinterpret_cast<const f4_a4 *>(a);  // rows are only 4-byte aligned here
            const f4 t = mem_nt(MEM) ? __builtin_nontemporal_load(src) : *src;
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (4 * j + k < L) ? a[k] : 0.0f;
        }
    }
}

template <int G, int MEM>
__device__ __forceinline__ void store_elems(float *__restrict__ p, long long r, int L, int j, const float (&v)[4])
{
    if constexpr (mem_vec(MEM)) {
        f4 *dst = reinterpret_cast<f4 *>(p) + (r * G + j);
        const f4 t = {v[0], v[1], v[2], v[3]};
        if (mem_nt(MEM)) __builtin_nontemporal_store(t, dst);
        else *dst = t;
    } else {
        float *a = p + r * (long long)L + 4 * j;
        if (4 * j + 3 < L) {
            const float4 t = make_float4(v[0], v[1], v[2], v[3]);
            __builtin_memcpy(a, &t, 16);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * j + k < L) a[k] = v[k];
        }
    }
}

// The 8 rows a lane group takes part in per batch: rows R0 + u * RPP + gq, u = 0..7.  With 16-byte-aligned rows (MEM != 0)
// or in the tail of a tensor this is load_elems per row (row index clamped to the last row).  Otherwise (`fast`: every row of
// the batch exists and at least 4 more rows follow it) each lane that owns part of a row issues ONE 16-byte access per row, also
// the lane at the ragged row end, whose access runs up to 3 elements into the rows that follow; those elements are zeroed after
// the load.  That is one memory instruction per row instead of four (a vector access for the full lanes plus three predicated
// scalars for the ragged one), and no address clamping.
template <int G, int MEM, int NU = 8>
__device__ __forceinline__ void load_rows8(const float *__restrict__ p, long long R0, long long rows, int L, int j, int gq, bool fast, float (&v)[NU][4])
{
    constexpr int RPP = 64 / G;
    if (!mem_vec(MEM) && fast) {
        f4 t[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) t[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (4 * j < L) {
            const float *a = p + (R0 + gq) * (long long)L + 4 * j;
            const long long step = (long long)RPP * L;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const f4_a4 *src = reinterpret_cast<const f4_a4 *>(a + u * step);
                t[u] = mem_nt(MEM) ? __builtin_nontemporal_load(src) : *src;
            }
        }
        const bool k1 = 4 * j + 1 < L, k2 = 4 * j + 2 < L, k3 = 4 * j + 3 < L;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            v[u][0] = t[u].x;
            v[u][1] = k1 ? t[u].y : 0.0f;
            v[u][2] = k2 ? t[u].z : 0.0f;
            v[u][3] = k3 ? t[u].w : 0.0f;
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long long r = R0 + u * RPP + gq;
            load_elems<G, MEM>(p, r < rows ? r : rows - 1, L, j, v[u]);
        }
    }
}

// Rows of L % 4 != 0 floats (124 bytes at L = 31) start at every 4-byte phase, so a lane's four elements are an
// unaligned 16-byte access and the ragged row end takes predicated scalars; written that way the gradient
// kernel reaches only 3.4 TB/s.  But the 64/G rows one instruction covers are CONTIGUOUS in memory and start
// 16-byte aligned: the wavefront puts them into LDS in (row, element) order and takes them out as one flat
// run of aligned 16-byte pieces, one store instruction for the whole run.  R0 = first row, nrows = valid rows
// (uniform); buf = this wavefront's 256 floats of LDS; called by all lanes.
template <int G, bool FULL>  // FULL: all 64 / G rows present and their floats a multiple of 4 (no scalar tail)
__device__ __forceinline__ void store_rows_staged(float *__restrict__ p, long long R0, int nrows, int L, int j, int gq, int lane, const float (&v)[4],
                                                  float *buf, const bool stream)
{
    __builtin_amdgcn_wave_barrier();  // (scheduling only: LDS operations of one wavefront execute in order)
    // Four unpredicated 4-byte writes per lane, element 3 first: the lane at the ragged row end spills up to 3 values
    // into the first elements of the next row, and the lane that owns those elements writes them with a LATER
    // instruction (its element index is smaller), so the valid value is the one that stays.  Lanes beyond the row
    // end take no part.  (buf has STAGE_FLOATS > 256 floats: the last row's spill stays inside it.)
    if (4 * j < L) {
        float *w = buf + gq * L + 4 * j;
        w[3] = v[3];
        __builtin_amdgcn_wave_barrier();
        w[2] = v[2];
        __builtin_amdgcn_wave_barrier();
        w[1] = v[1];
        __builtin_amdgcn_wave_barrier();
        w[0] = v[0];
    }
    __builtin_amdgcn_wave_barrier();
    const int nf = nrows * L;
    float *dst = p + R0 * (long long)L + 4 * lane;
    if (4 * lane + 3 < nf) {
        const f4 t = *reinterpret_cast<const f4 *>(buf + 4 * lane);
        if (stream) __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(dst));
        else *reinterpret_cast<f4 *>(dst) = t;
    } else if (!FULL) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * lane + k < nf) dst[k] = buf[4 * lane + k];
    }
}

__device__ __forceinline__ int find_segment(const SegTab &tab)
{
    int si = 0;
    for (int i = 1; i < tab.nseg; ++i)
        if ((int)blockIdx.x >= tab.s[i].blk0) si = i;
    return si;
}

// Order-preserving map of a float onto an unsigned integer (a < b  <=>  key(a) < key(b) for non-NaN values).
__device__ __forceinline__ uint32_t float_key(float v)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    return __builtin_bit_cast(float, k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// EXACT: the bit-reproducible operation sequence (polynomial exp); otherwise exp on the
// transcendental unit.  HARD (codes) implies EXACT at the call site.
// CODES_ONLY (with HARD; no W, no sum W^2 wanted): the code of a row is the arg-max of exp(beta * p / |p|), and
// for beta > 0 that is the arg-max of p itself unless rounding merges the two largest values.  So the kernel
// takes the arg-max of p with a group reduction and counts how many elements lie within a safety distance
// of the maximum: x = beta * (p * rn) carries two roundings (<= 1.2e-7 * |beta| each side) and the polynomial
// exp about 1.3e-7 relative, so elements whose x trails the maximum by at least 2e-6 + 1e-6 * |beta| are
// strictly smaller after exp as well.  A row with a second element inside that distance (about one row in
// 10^4 for uniform p), a non-finite or zero norm, or beta <= 0 goes through the exact sequence, so the codes are
// those of the exact kernel bit for bit, at the cost of one compare-and-count per element instead of an exp.
// NU: rows per lane group and batch.  8 = 64 rows per wavefront iteration.  4 (G = 8 only) = 32 rows: on tensors
// beyond the last-level cache twice as many, half as long wavefronts run the soft fast-arithmetic pass 3-4 % faster
// (6.33 against 6.15 TB/s at 250 M coefficients; lanes j < 4 of a group then own the rows).  Measured slower for the
// instruction-heavier forms (codes-only 5.2 against 5.6 TB/s, exact 4.3 against 4.7) and for ragged rows (5.95 against
// 6.24), which stay at 64 rows.
template <int G, int MEM, bool HARD, bool EXACT, bool CODES_ONLY = false, int NU = 8>
__global__ __launch_bounds__(256) void synth_fwd_kernel(const SegTab tab, const float beta)
{
    static_assert(NU == 8 || (NU == 4 && G == 8), "half batches exist for 8 lanes per row only");
    constexpr int RPP = 64 / G;
    constexpr int ROWS = 8 * NU;  // rows per wavefront iteration
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wavefront-uniform: row ranges stay in scalar registers
    const int j = lane % G, gq = lane / G;
    const int si = find_segment(tab);
    const float *__restrict__ p = tab.s[si].p;
    const float *__restrict__ q = tab.s[si].q;
    float *__restrict__ W = tab.s[si].W;
    float *__restrict__ Wh = tab.s[si].Wh;
    uint8_t *__restrict__ idx = tab.s[si].idx;
    float *__restrict__ part = tab.s[si].part;
    const long long rows = tab.s[si].rows;
    const int L = tab.s[si].L;

    const long long wt = (long long)((int)blockIdx.x - tab.s[si].blk0) * WAVES + wave;
    const long long row_begin = wt * tab.K * ROWS;
    if (row_begin >= rows) return;

    float q4[4];
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        valid[k] = 4 * j + k < L;
        q4[k] = valid[k] ? q[4 * j + k] : 0.0f;
    }

    float acc = 0.0f;
    for (int it = 0; it < tab.K; ++it) {
        const long long row0 = row_begin + (long long)it * ROWS;
        if (row0 >= rows) break;
        const bool fast_rows = row0 + ROWS + 4 <= rows;
        float myW = 0.0f;
        uint32_t myIdx = 0;
        static_for<G / 8>([&](auto UB) {
            constexpr int ub = 8 * decltype(UB)::value;
            float v[NU][4];
            load_rows8<G, MEM, NU>(p, row0 + ub * RPP, rows, L, j, gq, fast_rows, v);
            // squared norms of the 8 rows; lane j == ub+u keeps row u's and inverts it once
            float my_ss = 1.0f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float ss = group_sum<G>(chain_sq(v[u]));
                my_ss = (j == ub + u) ? ss : my_ss;
            }
            // EXACT: 1/||p||; fast: beta*log2(e)/||p||, so that one multiply per element feeds v_exp_f32
            const float my_rn = EXACT ? inv_norm(my_ss) : (beta * inv_norm(my_ss)) * LOG2E;
            float my_den = 1.0f, my_num = 0.0f;
            uint32_t my_bi = 0;
            bool slow = true;  // CODES_ONLY: some row of this batch needs the exact sequence (wavefront-uniform)
            if constexpr (CODES_ONLY) {
                // distance in p that corresponds to the safety distance in x: (2e-6 + 1e-6 * beta) / (beta * rn)
                const float my_thr = (2e-6f / beta + 1e-6f) * __builtin_sqrtf(my_ss);
                bool my_bad = !(my_ss > 0.0f && my_ss < __builtin_inff()) || !(beta > 0.0f);
                static_for<NU>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    const float thr = group_bcast<G, ub + u>(my_thr);
                    float pv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv[k] = (mem_vec(MEM) || valid[k]) ? v[u][k] : -__builtin_inff();
                    uint32_t bk = float_key(pv[0]), bi = 4 * j;
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const uint32_t kk = float_key(pv[k]);
                        const bool take = kk > bk;
                        bk = take ? kk : bk;
                        bi = take ? 4 * j + k : bi;
                    }
                    group_argmax<G>(bk, bi);
                    const float pmax = key_float(bk);
                    float cnt = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) cnt += (pmax - pv[k] <= thr) ? 1.0f : 0.0f;
                    cnt = group_sum<G>(cnt);
                    const bool mine = (j == ub + u);
                    my_bi = mine ? bi : my_bi;
                    my_bad = mine ? (my_bad || !(cnt == 1.0f)) : my_bad;
                });
                slow = __builtin_amdgcn_ballot_w64(my_bad) != 0;
            }
            if (slow) static_for<NU>([&](auto U) {
                constexpr int u = decltype(U)::value;
                const float rn = group_bcast<G, ub + u>(my_rn);
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ex = EXACT ? expf_spec(beta * (v[u][k] * rn)) : exp2_hw(v[u][k] * rn);
                    e[k] = (mem_vec(MEM) || valid[k]) ? ex : 0.0f;
                }
                const float den = group_sum<G>(((e[0] + e[1]) + e[2]) + e[3]);
                const float num = group_sum<G>(chain_dot(e, q4));
                uint32_t bi = 0;
                if constexpr (HARD) {
                    uint32_t bk = __builtin_bit_cast(uint32_t, e[0]);
                    bi = 4 * j;
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const uint32_t kk = __builtin_bit_cast(uint32_t, e[k]);
                        const bool take = kk > bk;  // padding is 0 with a larger index: never wins
                        bk = take ? kk : bk;
                        bi = take ? 4 * j + k : bi;
                    }
                    group_argmax<G>(bk, bi);
                }
                const bool mine = (j == ub + u);
                my_den = mine ? den : my_den;
                my_num = mine ? num : my_num;
                my_bi = mine ? bi : my_bi;
            });
            const float w = my_num * (1.0f / my_den);
            const bool owner = (j >= ub) && (j < ub + NU);
            myW = owner ? w : myW;
            if constexpr (HARD) myIdx = owner ? ((my_den == my_den) ? my_bi : 0u) : myIdx;  // (fast path: my_den stays 1)
        });
        // lane (gq, j) now holds row j*RPP + gq of this iteration: one coalesced store each
        const long long rr = row0 + (long long)j * RPP + gq;
        if (rr < rows && (NU == 8 || j < NU)) {
            if (W) W[rr] = myW;
            if constexpr (HARD) {
                if (idx) idx[rr] = (uint8_t)myIdx;
                if (Wh) Wh[rr] = q[myIdx];
            }
            acc = __builtin_fmaf(myW, myW, acc);
        }
    }
    if (part) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) part[wt] = acc;
    }
}

// ADAM: instead of (or besides) storing dp_c, apply torch.optim.Adam's update rule to p_c in place
// (adp_qtz_fan.py:208-209, 291: optimizer_prim = optim.Adam(p_c parameters); .step()):
//   m += (1-beta1)*(g-m);  v = beta2*v + (1-beta2)*g*g;  p -= step_size * m / (sqrt(v)*rbc2 + eps)
// with step_size = lr/(1-beta1^t), rbc2 = 1/sqrt(1-beta2^t) prepared per layer by adam_prep_kernel.
// This removes the dp_c round trip through memory: 6 tensor passes (p, m, v read and written) instead
// of 2 (K2) + 7 (separate Adam).
// DQ = false: no layer of the launch wants the gradient of its levels (adp_qtz_fix freezes them, adp_qtz_fix.py:649-652):
// the per-lane accumulators of dq, their updates and the closing shuffles are compiled out.
template <int G, int MEM, bool EXACT, bool ADAM, int NU = 8, bool DQ = true>  // NU: as in synth_fwd_kernel
__global__ __launch_bounds__(256) void synth_bwd_kernel(const SegTab tab, const float beta, const float wd, const AdamArgs ad)
{
    static_assert(NU == 8 || (NU == 4 && G == 8), "half batches exist for 8 lanes per row only");
    constexpr int ROWS = 8 * NU;  // rows per wavefront iteration
    constexpr int RPP = 64 / G;
    constexpr int LP = 4 * G;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wavefront-uniform: row ranges stay in scalar registers
    const int j = lane % G, gq = lane / G;
    const int si = find_segment(tab);
    const float *__restrict__ p = tab.s[si].p;
    const float *__restrict__ q = tab.s[si].q;
    const float *__restrict__ dW = tab.s[si].dW;
    float *__restrict__ dp = tab.s[si].dp;
    float *__restrict__ am = tab.s[si].am;
    float *__restrict__ av = tab.s[si].av;
    float step_size = 0.0f, rbc2 = 0.0f;
    if (ADAM) {
        step_size = tab.s[si].acoef[0];
        rbc2 = tab.s[si].acoef[1];
    }
    const float wds = tab.s[si].dss ? __builtin_fmaf(2.0f, tab.s[si].dss[0], wd) : wd;
    float *__restrict__ part = tab.s[si].part;
    const long long rows = tab.s[si].rows;
    const int L = tab.s[si].L;
    __shared__ __attribute__((aligned(16))) float stage_all[!mem_vec(MEM) ? WAVES * STAGE_FLOATS : 4];
    float *stage = stage_all + (!mem_vec(MEM) ? wave * STAGE_FLOATS : 0);
    const bool rag = !mem_vec(MEM) && tab.s[si].rag != 0;
    constexpr bool rag_stream = mem_nt(MEM);

    const long long wt = (long long)((int)blockIdx.x - tab.s[si].blk0) * WAVES + wave;
    const long long row_begin = wt * tab.K * ROWS;
    if (row_begin >= rows) return;

    float q4[4];
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        valid[k] = 4 * j + k < L;
        q4[k] = valid[k] ? q[4 * j + k] : 0.0f;
    }

    float accq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < tab.K; ++it) {
        const long long row0 = row_begin + (long long)it * ROWS;
        if (row0 >= rows) break;
        const bool fast_rows = row0 + ROWS + 4 <= rows;
        // FR = true: ragged rows leaving through LDS, and every row of the batch present (all but the last wavefronts of a
        // layer): no row clamping, no per-row bounds, one 16-byte access per lane and row in both directions.
        auto batch = [&](auto FR) {
            constexpr bool FASTRAG = decltype(FR)::value;
            static_for<G / 8>([&](auto UB) {
                constexpr int ub = 8 * decltype(UB)::value;
                float v[NU][4];
                float gw[NU];
                float m1[NU][4], m2[NU][4];  // (unused and removed by the compiler without ADAM)
#pragma unroll
                for (int u = 0; u < NU; ++u) gw[u] = 0.0f;
                if (dW) {
                    if (FASTRAG || fast_rows) {
                        const float *g0 = dW + (row0 + ub * RPP + gq);
#pragma unroll
                        for (int u = 0; u < NU; ++u) gw[u] = g0[u * RPP];
                    } else {
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const long long r = row0 + (ub + u) * RPP + gq;
                            gw[u] = dW[r < rows ? r : rows - 1];
                        }
                    }
                }
                load_rows8<G, MEM, NU>(p, row0 + ub * RPP, rows, L, j, gq, FASTRAG || fast_rows, v);
                if constexpr (ADAM) {
                    load_rows8<G, MEM, NU>(am, row0 + ub * RPP, rows, L, j, gq, FASTRAG || fast_rows, m1);
                    load_rows8<G, MEM, NU>(av, row0 + ub * RPP, rows, L, j, gq, FASTRAG || fast_rows, m2);
                }
                float my_ss = 1.0f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float ss = group_sum<G>(chain_sq(v[u]));
                    my_ss = (j == ub + u) ? ss : my_ss;
                }
                const float my_rn = inv_norm(my_ss);
                static_for<NU>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    const long long r = row0 + (ub + u) * RPP + gq;
                    const float rn = group_bcast<G, ub + u>(my_rn);
                    float pn[4], e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pn[k] = v[u][k] * rn;
                        const float ex = EXACT ? expf_spec(beta * pn[k]) : exp2_hw(pn[k] * (beta * LOG2E));
                        e[k] = (mem_vec(MEM) || valid[k]) ? ex : 0.0f;
                    }
                    const float den = group_sum<G>(((e[0] + e[1]) + e[2]) + e[3]);
                    const float num = group_sum<G>(chain_dot(e, q4));
                    const float rden = 1.0f / den;
                    const float w = num * rden;
                    const float g = __builtin_fmaf(wds, w, gw[u]);
                    const float dot = g * w;
                    float s[4], gpn[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s[k] = e[k] * rden;
                        const float gs = g * q4[k];
                        gpn[k] = beta * (s[k] * (gs - dot));
                    }
                    const float dpn = group_sum<G>(chain_dot(pn, gpn));
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(-pn[k], dpn, gpn[k]) * rn;
                    float pn2[4], mo[4], vo[4];
                    if constexpr (ADAM) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            mo[k] = __builtin_fmaf(ad.omb1, o[k] - m1[u][k], m1[u][k]);
                            vo[k] = __builtin_fmaf(ad.omb2 * o[k], o[k], ad.b2 * m2[u][k]);
                            const float denom = __builtin_fmaf(sqrtf(vo[k]), rbc2, ad.eps);
                            pn2[k] = v[u][k] - step_size * (mo[k] / denom);
                        }
                    }
                    if constexpr (FASTRAG) {
                        const long long R0 = row0 + (long long)(ub + u) * RPP;
                        if (!ADAM || dp) store_rows_staged<G, true>(dp, R0, RPP, L, j, gq, lane, o, stage, rag_stream);
                        if constexpr (ADAM) {
                            store_rows_staged<G, true>(const_cast<float *>(p), R0, RPP, L, j, gq, lane, pn2, stage, rag_stream);
                            store_rows_staged<G, true>(am, R0, RPP, L, j, gq, lane, mo, stage, rag_stream);
                            store_rows_staged<G, true>(av, R0, RPP, L, j, gq, lane, vo, stage, rag_stream);
                        }
                        if constexpr (DQ) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) accq[k] = __builtin_fmaf(s[k], g, accq[k]);
                        }
                    } else {
                        if (rag) {  // uniform: the rows of this instruction leave as one aligned run (all lanes take part)
                            const long long R0 = row0 + (long long)(ub + u) * RPP;
                            const int nrows = (int)(rows - R0 < RPP ? (rows - R0 > 0 ? rows - R0 : 0) : RPP);
                            if (!ADAM || dp) store_rows_staged<G, false>(dp, R0, nrows, L, j, gq, lane, o, stage, rag_stream);
                            if constexpr (ADAM) {
                                store_rows_staged<G, false>(const_cast<float *>(p), R0, nrows, L, j, gq, lane, pn2, stage, rag_stream);
                                store_rows_staged<G, false>(am, R0, nrows, L, j, gq, lane, mo, stage, rag_stream);
                                store_rows_staged<G, false>(av, R0, nrows, L, j, gq, lane, vo, stage, rag_stream);
                            }
                        }
                        if (r < rows) {
                            if (!rag) {
                                if (!ADAM || dp) store_elems<G, MEM>(dp, r, L, j, o);
                                if constexpr (ADAM) {
                                    store_elems<G, MEM>(const_cast<float *>(p), r, L, j, pn2);
                                    store_elems<G, MEM>(am, r, L, j, mo);
                                    store_elems<G, MEM>(av, r, L, j, vo);
                                }
                            }
                            if constexpr (DQ) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) accq[k] = __builtin_fmaf(s[k], g, accq[k]);
                            }
                        }
                    }
                });
            });
        };
        if constexpr (!mem_vec(MEM)) {
            if (rag && fast_rows) batch(std::true_type{});
            else batch(std::false_type{});
        } else {
            batch(std::false_type{});
        }
    }
    if (DQ && part) {
#pragma unroll
        for (int m = G; m < 64; m <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) accq[k] += __shfl_xor(accq[k], m);
        }
        if (gq == 0) {
            float *o = part + wt * LP + 4 * j;
            reinterpret_cast<float4 *>(o)[0] = make_float4(accq[0], accq[1], accq[2], accq[3]);
        }
    }
}

// One thread per layer: advance the layer's Adam step counter (a one-element fp32 tensor, as
// torch.optim.Adam keeps it with capturable=True) and derive the step-dependent coefficients in double.
struct AdamPrepTab {
    float *step[MAX_SEGS];
    float *coef[MAX_SEGS];
    int n;
    double lr, b1, b2;
};
__global__ void adam_prep_kernel(const AdamPrepTab t)
{
    const int i = threadIdx.x;
    if (i >= t.n) return;
    bool first = true;  // a step tensor shared by several layers is advanced once
    for (int k = 0; k < i; ++k) first = first && t.step[k] != t.step[i];
    const float tnew = t.step[i][0] + 1.0f;
    const double bc1 = 1.0 - pow(t.b1, (double)tnew), bc2 = 1.0 - pow(t.b2, (double)tnew);
    t.coef[i][0] = (float)(t.lr / bc1);
    t.coef[i][1] = (float)(1.0 / sqrt(bc2));
    __syncthreads();
    if (first) t.step[i][0] = tnew;
}

// Second stage of the cross-row sums: adds the per-wavefront partials of each layer in a fixed
// order.  A workgroup of 1024 threads = NL thread rows x `width` columns (NL = 1024/width) owns a
// chunk of 32*NL consecutive partial rows: thread row r adds rows r, r+NL, r+2NL, .. of the chunk
// in that order, then the NL running sums of a column are combined by a halving tree in LDS.
// Layers with more partials than one chunk are reduced level by level (chunk results become the
// partials of the next level), so the order depends only on the number of rows.
struct RedSeg {
    const float *part;
    float *out;
    long long n;
    int width, outw, blk0, pad;
};
struct RedTab {
    int nseg;
    RedSeg s[MAX_SEGS];
};

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const RedTab tab)
{
    __shared__ double sh[1024];  // summed in double: the partials of a layer's dq cancel to a small fraction of their size
    int si = 0;
    for (int i = 1; i < tab.nseg; ++i)
        if ((int)blockIdx.x >= tab.s[i].blk0) si = i;
    const RedSeg sg = tab.s[si];
    const int tid = threadIdx.x;
    const int width = sg.width, nl = 1024 / width;
    const int lvl = tid % width, ln = tid / width;
    const long long chunk = (long long)((int)blockIdx.x - sg.blk0);
    const long long r0 = chunk * 32 * nl;
    const long long r1 = r0 + 32 * nl < sg.n ? r0 + 32 * nl : sg.n;
    double a = 0.0;
    for (long long i = r0 + ln; i < r1; i += nl) a += (double)sg.part[i * width + lvl];
    sh[tid] = a;
    __syncthreads();
    for (int s = nl >> 1; s >= 1; s >>= 1) {
        if (ln < s) sh[tid] += sh[tid + s * width];
        __syncthreads();
    }
    // a single-chunk level writes the layer's result (outw columns); otherwise one row per chunk
    const bool last = sg.n <= 32 * nl;
    if (ln == 0 && lvl < (last ? sg.outw : width)) sg.out[(last ? 0 : chunk * width) + lvl] = (float)sh[lvl];
}

// ------------------------------------------------------------------------------------ host side

static inline int lanes_for(int L) { return L <= 32 ? 8 : (L <= 64 ? 16 : 32); }

// Iterations (of 64 rows) per wavefront.  One: the kernels are bound by load latency, and more,
// shorter-lived wavefronts overlap better than long ones (measured: 5.7 TB/s at 1, 4.7 TB/s at 16).
static inline int choose_K(long long total_rows)
{
    (void)total_rows;
    return 1;
}

static inline long long wave_tiles(long long rows, int K, int rows_it = ROWS_IT) { return (rows + (long long)K * rows_it - 1) / ((long long)K * rows_it); }

// streaming cache policy when the tensors cannot stay in the 256 MiB last-level cache anyway
// (what counts is the traffic of the call: with Adam, p, m and v are each read and written)
static inline bool call_streams(const aq_seg *segs, int nseg, bool adam)
{
    long long total_bytes = 0;
    for (int i = 0; i < nseg; ++i) total_bytes += segs[i].rows * (long long)segs[i].L * 4;
    return total_bytes * (adam ? 6 : 1) > (128LL << 20);
}
// fewest rows per wavefront iteration a layer can get (sizes the workspace): the soft fast-arithmetic forward pass runs
// 32-row wavefronts on streamed, 16-byte-aligned layers of 32 levels (run() decides with the flags of the call)
static inline int rows_per_wave(bool bwd, int L, bool streams) { (void)bwd; return (streams && L == 32) ? 32 : ROWS_IT; }

struct Plan {
    int K;
    size_t bytes;
    size_t off[1 << 12];   // per-wavefront partials of segment i
    size_t tmp[1 << 12];   // two ping-pong areas for the multi-level reduction of segment i
    size_t tmp_half[1 << 12];
};

static inline long long red_chunks(long long n, int width) { const long long ch = 32LL * (1024 / width); return (n + ch - 1) / ch; }

static int validate(const aq_seg *segs, int nseg, float beta, bool bwd, bool adam = false)
{
    AQ_REQUIRE(segs != nullptr && nseg >= 1 && nseg <= (1 << 12), -EINVAL, "segs must hold 1..4096 segments (got %d)", nseg);
    AQ_REQUIRE(beta == beta && std::abs(beta) <= AQ_MAX_ABS_BETA, -EDOM, "beta=%g outside [-%g, %g]", (double)beta,
               (double)AQ_MAX_ABS_BETA, (double)AQ_MAX_ABS_BETA);
    for (int i = 0; i < nseg; ++i) {
        const aq_seg &s = segs[i];
        AQ_REQUIRE(s.L >= 1 && s.L <= AQ_MAX_LEVELS, -EINVAL, "segment %d: L=%d outside 1..%d", i, s.L, AQ_MAX_LEVELS);
        AQ_REQUIRE(s.rows >= 0, -EINVAL, "segment %d: negative row count", i);
        AQ_REQUIRE(s.rows == 0 || (s.p_c && s.q), -EFAULT, "segment %d: p_c / q is NULL", i);
        AQ_REQUIRE(((uintptr_t)s.p_c & 3) == 0 && ((uintptr_t)s.q & 3) == 0, -EINVAL, "segment %d: misaligned pointer", i);
        if (bwd && !adam) AQ_REQUIRE(s.rows == 0 || s.dp_c, -EFAULT, "segment %d: dp_c is NULL", i);
        if (adam) {
            AQ_REQUIRE(s.adam_step && (s.rows == 0 || (s.adam_m && s.adam_v)), -EFAULT, "segment %d: Adam state is NULL", i);
            AQ_REQUIRE(((uintptr_t)s.adam_m & 3) == 0 && ((uintptr_t)s.adam_v & 3) == 0, -EINVAL, "segment %d: misaligned Adam state", i);
        }
    }
    return 0;
}

static void make_plan(const aq_seg *segs, int nseg, bool bwd, Plan &pl)
{
    long long total = 0;
    for (int i = 0; i < nseg; ++i) total += segs[i].rows;
    pl.K = choose_K(total);
    const bool streams = call_streams(segs, nseg, bwd);  // (backward: the traffic multiplier of the Adam form, an upper bound)
    size_t off = 0;
    for (int i = 0; i < nseg; ++i) {
        pl.off[i] = off;
        const bool need = bwd ? segs[i].dq != nullptr : segs[i].sumsq != nullptr;
        if (need && segs[i].rows > 0) {
            const size_t width = bwd ? 4 * lanes_for(segs[i].L) : 1;
            const long long nw = wave_tiles(segs[i].rows, pl.K, rows_per_wave(bwd, segs[i].L, streams));
            size_t n = (size_t)nw * width * sizeof(float);
            off += (n + 255) & ~(size_t)255;
            pl.tmp[i] = off;
            pl.tmp_half[i] = (((size_t)red_chunks(nw, (int)width) * width * sizeof(float)) + 255) & ~(size_t)255;
            off += 2 * pl.tmp_half[i];
        }
    }
    pl.bytes = off;
}

template <bool BWD>
static void launch_class(int G, int mem, bool hard, bool exact, const SegTab &tab, int nblocks, float beta, float wd, const AdamArgs *adam,
                         hipStream_t st, bool codes_only = false, bool half = false, bool no_dq = false)
{
    const AdamArgs ad = adam ? *adam : AdamArgs{};
#define AQ_LAUNCH(GG, MM)                                                                                             \
    if (BWD && no_dq && GG == 8 && (MM == 3 || MM == 0) && !exact) { /* frozen levels, ragged rows: adp_qtz_fix */  \
        constexpr bool RG = (GG == 8 && (MM == 3 || MM == 0));                                                        \
        if (adam) { note_kernel("aq::synth_bwd_kernel<%d, %d, false, true, 8, false>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, true, 8, !RG>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
        else { note_kernel("aq::synth_bwd_kernel<%d, %d, false, false, 8, false>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, false, 8, !RG>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
    } else if (BWD && half && GG == 8 && MM == 2 && !exact) {                                                              \
        constexpr int NH = (GG == 8 && MM == 2) ? 4 : 8;                                                              \
        if (adam) { note_kernel("aq::synth_bwd_kernel<%d, %d, false, true, 4>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, true, NH>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
        else { note_kernel("aq::synth_bwd_kernel<%d, %d, false, false, 4>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, false, NH>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
    } else if (BWD) {                                                                                                 \
        if (adam && exact) { note_kernel("aq::synth_bwd_kernel<%d, %d, true, true>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, true, true>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
        else if (adam) { note_kernel("aq::synth_bwd_kernel<%d, %d, false, true>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, true>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
        else if (exact) { note_kernel("aq::synth_bwd_kernel<%d, %d, true, false>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, true, false>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
        else { note_kernel("aq::synth_bwd_kernel<%d, %d, false, false>", GG, MM); hipLaunchKernelGGL((synth_bwd_kernel<GG, MM, false, false>), dim3(nblocks), dim3(256), 0, st, tab, beta, wd, ad); } \
    } else if (half && GG == 8 && MM == 2 && !hard && !exact) {                                                      \
        constexpr int NH = (GG == 8 && MM == 2) ? 4 : 8;                                                              \
        { note_kernel("aq::synth_fwd_kernel<%d, %d, false, false, false, 4>", GG, MM); hipLaunchKernelGGL((synth_fwd_kernel<GG, MM, false, false, false, NH>), dim3(nblocks), dim3(256), 0, st, tab, beta); } \
    } else if (hard && codes_only)                                                                                    \
        { note_kernel("aq::synth_fwd_kernel<%d, %d, true, true, true>", GG, MM); hipLaunchKernelGGL((synth_fwd_kernel<GG, MM, true, true, true>), dim3(nblocks), dim3(256), 0, st, tab, beta); } \
    else if (hard)                                                                                                    \
        { note_kernel("aq::synth_fwd_kernel<%d, %d, true, true>", GG, MM); hipLaunchKernelGGL((synth_fwd_kernel<GG, MM, true, true>), dim3(nblocks), dim3(256), 0, st, tab, beta); }       \
    else if (exact)                                                                                                   \
        { note_kernel("aq::synth_fwd_kernel<%d, %d, false, true>", GG, MM); hipLaunchKernelGGL((synth_fwd_kernel<GG, MM, false, true>), dim3(nblocks), dim3(256), 0, st, tab, beta); }      \
    else                                                                                                              \
        { note_kernel("aq::synth_fwd_kernel<%d, %d, false, false>", GG, MM); hipLaunchKernelGGL((synth_fwd_kernel<GG, MM, false, false>), dim3(nblocks), dim3(256), 0, st, tab, beta); }
#define AQ_LAUNCH_G(GG)                      \
    if (mem == 2) { AQ_LAUNCH(GG, 2) }       \
    else if (mem == 1) { AQ_LAUNCH(GG, 1) }  \
    else if (mem == 3) { AQ_LAUNCH(GG, 3) }  \
    else { AQ_LAUNCH(GG, 0) }
    if (G == 8) { AQ_LAUNCH_G(8) }
    else if (G == 16) { AQ_LAUNCH_G(16) }
    else { AQ_LAUNCH_G(32) }
#undef AQ_LAUNCH_G
#undef AQ_LAUNCH
}

struct AdamHost {
    double lr, b1, b2, eps;
};

template <bool BWD>
static int run(const aq_seg *segs, int nseg, float beta, float wd, unsigned flags, void *workspace, size_t workspace_bytes, hipStream_t st,
               const AdamHost *adam = nullptr)
{
    const bool exact = (flags & AQ_FLAG_EXACT) != 0;
    if (int rc = validate(segs, nseg, beta, BWD, adam != nullptr)) return rc;
    static thread_local Plan pl;
    make_plan(segs, nseg, BWD, pl);
    const size_t coef_off = pl.bytes;  // Adam coefficients: two floats per segment behind the partials
    if (adam) pl.bytes += ((size_t)nseg * 2 * sizeof(float) + 255) & ~(size_t)255;
    AQ_REQUIRE(pl.bytes == 0 || (workspace && workspace_bytes >= pl.bytes), -ENOSPC, "workspace too small: need %zu bytes, got %zu",
               pl.bytes, workspace_bytes);
    AQ_REQUIRE(((uintptr_t)workspace & 15) == 0, -EINVAL, "workspace must be 16-byte aligned");
    AdamArgs ad_dev = {};
    float *coef = nullptr;
    if (adam) {
        AQ_REQUIRE(nseg <= MAX_SEGS, -E2BIG, "fused Adam update: at most %d layers per call (got %d)", MAX_SEGS, nseg);
        AQ_REQUIRE(adam->lr >= 0 && adam->b1 >= 0 && adam->b1 < 1 && adam->b2 >= 0 && adam->b2 < 1 && adam->eps >= 0, -EDOM,
                   "fused Adam update: lr=%g betas=(%g, %g) eps=%g out of range", adam->lr, adam->b1, adam->b2, adam->eps);
        coef = reinterpret_cast<float *>(static_cast<char *>(workspace) + coef_off);
        AdamPrepTab t = {};
        t.n = nseg; t.lr = adam->lr; t.b1 = adam->b1; t.b2 = adam->b2;
        for (int i = 0; i < nseg; ++i) {
            t.step[i] = segs[i].adam_step;
            t.coef[i] = coef + 2 * i;
        }
        hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, st, t);
        ad_dev.omb1 = (float)(1.0 - adam->b1); ad_dev.b2 = (float)adam->b2; ad_dev.omb2 = (float)(1.0 - adam->b2); ad_dev.eps = (float)adam->eps;
    }

    const bool stream_policy = call_streams(segs, nseg, adam != nullptr);
    bool any_hard = false;
    for (int i = 0; i < nseg; ++i) any_hard = any_hard || segs[i].idx != nullptr || segs[i].W_hard != nullptr;
    // 32-row wavefronts for the aligned 32-level layers of large calls in fast arithmetic: forward without codes +2-4 %,
    // gradient +3-5 %, gradient with Adam +4-12 % on 50 M .. 1 B coefficients.  (A training step's 19 layers, 0.27 M rows,
    // measured no faster and keep 64 rows; so do ragged rows: 5.24 against 5.41 TB/s in the gradient at L = 31.)
    int half_mask = 7;  // 1 forward, 2 gradient, 4 gradient with Adam
    long long call_rows = 0;
    for (int i = 0; i < nseg; ++i) call_rows += segs[i].rows;
    const bool half_call = stream_policy && !exact && call_rows >= (1LL << 19) &&
                           (BWD ? (half_mask & (adam ? 4 : 2)) != 0 : (!any_hard && (half_mask & 1)));
    static thread_local unsigned char seg_rows_it[1 << 12];
    for (int i = 0; i < nseg; ++i) seg_rows_it[i] = ROWS_IT;

    // main launches: one per (lanes-per-row, memory access) class, at most MAX_SEGS segments each
    for (int G = 8; G <= 32; G <<= 1)
        for (int vec = 0; vec < 2; ++vec) {
            SegTab tab;
            tab.nseg = 0;
            tab.K = pl.K;
            int nblocks = 0;
            bool hard = false, wants_w = false;  // wants_w: some layer of the launch asks for W or sum W^2 (backward: dq)
            auto flush = [&]() {
                if (tab.nseg)
                    launch_class<BWD>(G, vec ? (stream_policy ? 2 : 1) : (stream_policy ? 3 : 0), hard, exact, tab, nblocks, beta, wd, adam ? &ad_dev : nullptr, st,
                                      hard && !wants_w, half_call && vec, BWD && !wants_w);
                tab.nseg = 0;
                nblocks = 0;
                hard = wants_w = false;
            };
            for (int i = 0; i < nseg; ++i) {
                const aq_seg &s = segs[i];
                if (s.rows == 0 || lanes_for(s.L) != G) continue;
                bool v = (s.L == 4 * G) && (((uintptr_t)s.p_c & 15) == 0);
                if (BWD) v = v && (((uintptr_t)s.dp_c & 15) == 0);
                if (adam) v = v && (((uintptr_t)s.adam_m & 15) == 0) && (((uintptr_t)s.adam_v & 15) == 0);
                if ((int)v != vec) continue;
                SegDev &d = tab.s[tab.nseg];
                d.p = s.p_c; d.q = s.q; d.dW = s.dW; d.dss = s.dsumsq; d.W = s.W; d.Wh = s.W_hard; d.dp = s.dp_c; d.idx = s.idx;
                d.am = adam ? s.adam_m : nullptr; d.av = adam ? s.adam_v : nullptr; d.acoef = adam ? coef + 2 * i : nullptr;
                const bool need = BWD ? s.dq != nullptr : s.sumsq != nullptr;
                d.part = need ? reinterpret_cast<float *>(static_cast<char *>(workspace) + pl.off[i]) : nullptr;
                d.rows = s.rows; d.L = s.L; d.blk0 = nblocks;
                // ragged rows leave through LDS when the rows of one instruction form a 16-byte-aligned run
                d.rag = 0;
                if (BWD && !v && (s.L & 3) != 0 && ((64 / G) * s.L) % 4 == 0) {
                    bool al = (((uintptr_t)s.dp_c & 15) == 0);
                    if (adam) al = al && (((uintptr_t)s.p_c | (uintptr_t)s.adam_m | (uintptr_t)s.adam_v) & 15) == 0;
                    if (al) d.rag = 1;
                }
                hard = hard || s.idx != nullptr || s.W_hard != nullptr;
                wants_w = wants_w || s.W != nullptr || d.part != nullptr;
                if (half_call && G == 8 && v) seg_rows_it[i] = 32;
                const long long nb = (wave_tiles(s.rows, pl.K, seg_rows_it[i]) + WAVES - 1) / WAVES;
                AQ_REQUIRE(nb + nblocks < (1LL << 31), -E2BIG, "segment %d: too many rows for one launch", i);
                nblocks += (int)nb;
                if (++tab.nseg == MAX_SEGS) flush();
            }
            flush();
        }
    if (int rc = check_launch(BWD ? "aq_synth_bwd" : "aq_synth_fwd")) return rc;

    // second stage: fixed-order sum of the per-wavefront partials of each layer, level by level
    static thread_local long long cur_n[1 << 12];
    static thread_local const float *cur_src[1 << 12];
    bool more = false;
    for (int i = 0; i < nseg; ++i) {
        const aq_seg &s = segs[i];
        float *out = BWD ? s.dq : s.sumsq;
        cur_n[i] = -1;
        if (!out) continue;
        cur_n[i] = s.rows ? wave_tiles(s.rows, pl.K, seg_rows_it[i]) : 0;
        cur_src[i] = reinterpret_cast<const float *>(static_cast<const char *>(workspace) + pl.off[i]);
        more = true;
    }
    for (int level = 0; more; ++level) {
        more = false;
        RedTab red;
        red.nseg = 0;
        int nblocks = 0;
        auto flush_red = [&]() {
            if (red.nseg) hipLaunchKernelGGL(reduce_partials_kernel, dim3(nblocks), dim3(1024), 0, st, red);
            red.nseg = 0;
            nblocks = 0;
        };
        for (int i = 0; i < nseg; ++i) {
            if (cur_n[i] < 0) continue;
            const aq_seg &s = segs[i];
            const int width = BWD ? 4 * lanes_for(s.L) : 1;
            const long long chunks = std::max<long long>(1, red_chunks(cur_n[i], width));
            RedSeg &r = red.s[red.nseg];
            r.part = cur_src[i];
            r.n = cur_n[i];
            r.width = width;
            r.outw = BWD ? s.L : 1;
            r.blk0 = nblocks;
            if (chunks == 1) {
                r.out = BWD ? s.dq : s.sumsq;
                cur_n[i] = -1;
            } else {
                float *dst = reinterpret_cast<float *>(static_cast<char *>(workspace) + pl.tmp[i] + (level & 1) * pl.tmp_half[i]);
                r.out = dst;
                cur_src[i] = dst;
                cur_n[i] = chunks;
                more = true;
            }
            nblocks += (int)chunks;
            if (++red.nseg == MAX_SEGS) flush_red();
        }
        flush_red();
    }
    return check_launch(BWD ? "aq_synth_bwd(reduce)" : "aq_synth_fwd(reduce)");
}

// ---- workspace of the entry point that allocates nothing (aq_synth_fwd_static) -------------------------
// A file-scope device array; every GPU the library runs on holds its own copy until the process ends.  In
// host code the array's name is only a handle for hipGetSymbolAddress: the device address is looked up once
// per device (with that device current) and from then on only that address is used, as a kernel argument
// like any caller-provided workspace.
static __device__ __align__(256) unsigned char scratch[AQ_STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static void *scratch_addr[MAX_DEVICES];  // [device ordinal]; written once under scratch_mutex
static std::mutex scratch_mutex;

static int static_scratch(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    AQ_REQUIRE(e == hipSuccess, -ENODEV, "aq_synth_fwd_static: hipGetDevice failed: %s", hipGetErrorString(e));
    AQ_REQUIRE(dev >= 0 && dev < MAX_DEVICES, -ENODEV, "aq_synth_fwd_static: device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_addr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        AQ_REQUIRE(e == hipSuccess && ptr != nullptr, -EIO, "aq_synth_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev,
                   hipGetErrorString(e));
        scratch_addr[dev] = ptr;
    }
    *out = scratch_addr[dev];
    return 0;
}

} // namespace aq

extern "C" {

int aq_synth_fwd_static(const aq_seg *segs, int nseg, float beta, unsigned flags, void *stream)
{
    if (int rc = aq::validate(segs, nseg, beta, false)) return rc;
    const size_t need = aq_synth_workspace_bytes(segs, nseg, 0);
    AQ_REQUIRE(need <= (size_t)AQ_STATIC_SCRATCH_BYTES, -E2BIG,
               "aq_synth_fwd_static: these layers need %zu bytes of partial sums, the static workspace holds %zu; use aq_synth_fwd",
               need, (size_t)AQ_STATIC_SCRATCH_BYTES);
    void *ws = nullptr;
    if (int rc = aq::static_scratch(&ws)) return rc;
    return aq::run<false>(segs, nseg, beta, 0.0f, flags, ws, (size_t)AQ_STATIC_SCRATCH_BYTES, static_cast<hipStream_t>(stream));
}

int aq_synth_fwd(const aq_seg *segs, int nseg, float beta, unsigned flags, void *workspace, size_t workspace_bytes, void *stream)
{
    return aq::run<false>(segs, nseg, beta, 0.0f, flags, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int aq_synth_bwd(const aq_seg *segs, int nseg, float beta, float wd, unsigned flags, void *workspace, size_t workspace_bytes, void *stream)
{
    return aq::run<true>(segs, nseg, beta, wd, flags, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int aq_synth_bwd_adam(const aq_seg *segs, int nseg, float beta, float wd, double lr, double beta1, double beta2, double eps, unsigned flags,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    const aq::AdamHost ad{lr, beta1, beta2, eps};
    return aq::run<true>(segs, nseg, beta, wd, flags, workspace, workspace_bytes, static_cast<hipStream_t>(stream), &ad);
}

size_t aq_synth_workspace_bytes(const aq_seg *segs, int nseg, int backward)
{
    if (!segs || nseg < 1 || nseg > (1 << 12)) return 0;
    static thread_local aq::Plan pl;
    aq::make_plan(segs, nseg, backward != 0, pl);
    return pl.bytes + (backward == 2 ? (((size_t)nseg * 2 * sizeof(float) + 255) & ~(size_t)255) : 0);
}

} // extern "C"
