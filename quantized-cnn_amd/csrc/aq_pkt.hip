// Probabilistic Knowledge Transfer loss (Passalis & Tefas, ECCV 2018; cited by the reference, README.md:24-29):
// the batch x batch cosine-similarity kernel matrices of a student's and a teacher's feature vectors, turned
// into conditional probabilities row by row, compared by the divergence  mean_ij Pt_ij * log((Pt_ij + eps) /
// (Ps_ij + eps)).  The reference's scripts never compute it (adp_qtz_hint.py:301-318 has no teacher pass), so this
// restates the published method: parity unpinned - no reference implementation.
//
//   normalise : xh = x / (|x| + eps) per row (NaN -> 0); also written transposed for the backward product
//   gram      : K = (xh xh^T + 1) / 2 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), the dense contraction
//   rows      : r_i = sum_j K_ij for student and teacher; loss_i = sum_j Pt_ij log((Pt_ij + eps) / (Ps_ij + eps)),
//               c_i = sum_j g_ij Ps_ij with g_ij = d loss / d Ps_ij; loss = fixed-order sum of loss_i / N^2
//   backward  : M_ij = dG_ij + dG_ji (element-wise from the symmetric K's and the row scalars), dxh = M xh on the
//               matrix cores, then through the normalisation.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

typedef float pkt_f4 __attribute__((ext_vector_type(4)));

// one workgroup per row: xh[i, :] = x[i, :] / (|x_i| + eps), xhT[:, i] likewise, norm[i] = |x_i|
__global__ __launch_bounds__(256) void pkt_normalize_kernel(const float *__restrict__ x, float *__restrict__ xh, float *__restrict__ xhT,
                                                            float *__restrict__ norm, int N, int D, float eps)
{
    __shared__ float sh[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    float a = 0.0f;
    for (int d = tid; d < D; d += 256) {
        const float v = x[(long long)i * D + d];
        a = __builtin_fmaf(v, v, a);
    }
    sh[tid] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float nr = __builtin_sqrtf(sh[0]);
    const float inv = 1.0f / (nr + eps);
    for (int d = tid; d < D; d += 256) {
        float v = x[(long long)i * D + d] * inv;
        v = v == v ? v : 0.0f;
        xh[(long long)i * D + d] = v;
        if (xhT) xhT[(long long)d * N + i] = v;
    }
    if (tid == 0) norm[i] = nr;
}

// C[M, Nc] = A[M, K] * B[Nc, K]^T (both operands K-contiguous), fp32 on the matrix cores.  A workgroup of four
// wavefronts owns a 64 x 64 tile, a wavefront 32 x 32 (2 x 2 MFMA tiles).  Operands go from memory (L2: these
// matrices are a few MB) straight into registers as 16-byte loads of four consecutive k per lane: lane (r, kq)
// of a 16-row fragment holds k = 16 kb + 4 kq + t in step t, for A and B alike, so the four MFMAs of a k-block
// cover its 16 k exactly once.
// GRAM: the cosine-kernel matrix K = (xh xh^T + 1) / 2 of the RAW features x (A = B = x), xh = x / (|x| + eps): every
// lane sums the squares of the operand elements it streams anyway, four lanes complete a row's |x|^2, and the
// epilogue scales the raw dot product by 1 / ((|x_i| + eps) (|x_j| + eps)) - no normalisation pass, no normalised
// copy.  blockIdx.z selects one of two feature matrices (student / teacher in one launch); tiles below the diagonal
// return at once, tiles above it also write their mirror image (the matrix is symmetric bit for bit by construction).
struct PktGram {
    const float *x[2];
    float *K[2];
    int D[2];
    float eps;
};
template <bool GRAM>
__global__ __launch_bounds__(256) void pkt_gemm_nt_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ Cm, int M,
                                                          int Nc, int K, const PktGram gr)
{
    if (GRAM) {
        if (blockIdx.x < blockIdx.y) return;
        A = B = gr.x[blockIdx.z];
        Cm = gr.K[blockIdx.z];
        K = gr.D[blockIdx.z];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    pkt_f4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = pkt_f4{0.f, 0.f, 0.f, 0.f};
    float ssa[2] = {0.0f, 0.0f}, ssb[2] = {0.0f, 0.0f};  // GRAM: this lane's share of |x|^2 of its two A rows / B rows
    const bool vec = (K & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
    auto load = [&](const float *__restrict__ P, int rows, int row, int k0, float (&v)[4]) {
        const int k = k0 + 4 * kq;
        if (row < rows && vec && k + 3 < K) {
            const float4 t = *reinterpret_cast<const float4 *>(P + (long long)row * K + k);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = (row < rows && k + t < K) ? P[(long long)row * K + k + t] : 0.0f;
        }
    };
    // Interior tiles (every row present, 16-byte-aligned rows, K a multiple of 16; wavefront-uniform): four k-blocks of
    // operands in flight in a ring of register stages, every load unconditional - the block index is clamped, the last
    // block may be fetched again - so that the compiler can wait for exactly the stage it is about to use.  With loads
    // behind per-lane bounds checks (the general path below) it waits for ALL outstanding loads before each block's
    // MFMAs, the prefetch included, and the product runs at the L2 latency: 35 us at 1024 x 1024 x 512, 7 us of it
    // matrix work.
    if (vec && (K & 15) == 0 && K >= 16 && m0 + 32 <= M && n0 + 32 <= Nc) {
        constexpr int S = 4;
        const int nb = K >> 4;
        const pkt_f4 *pa[2], *pb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pa[i] = reinterpret_cast<const pkt_f4 *>(A + (long long)(m0 + 16 * i + r) * K) + kq;
            pb[i] = reinterpret_cast<const pkt_f4 *>(B + (long long)(n0 + 16 * i + r) * K) + kq;
        }
        pkt_f4 av[S][2], bv[S][2];
#pragma unroll
        for (int s0 = 0; s0 < S - 1; ++s0) {
            const int bl = min(s0, nb - 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[s0][i] = pa[i][4 * bl];
                bv[s0][i] = pb[i][4 * bl];
            }
        }
        for (int b = 0; b < nb; b += S) {
#pragma unroll
            for (int s0 = 0; s0 < S; ++s0) {
                const int bl = min(b + s0 + S - 1, nb - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    av[(s0 + S - 1) % S][i] = pa[i][4 * bl];
                    bv[(s0 + S - 1) % S][i] = pb[i][4 * bl];
                }
                if (b + s0 < nb) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s0][i][t], bv[s0][j][t], acc[i][j], 0, 0, 0);
                    if (GRAM) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                ssa[i] = __builtin_fmaf(av[s0][i][t], av[s0][i][t], ssa[i]);
                                ssb[i] = __builtin_fmaf(bv[s0][i][t], bv[s0][i][t], ssb[i]);
                            }
                    }
                }
            }
        }
    } else {
    float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        load(A, M, m0 + 16 * i + r, 0, a0[i]);
        load(B, Nc, n0 + 16 * i + r, 0, b0[i]);
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (k0 + 16 < K) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                load(A, M, m0 + 16 * i + r, k0 + 16, a1[i]);
                load(B, Nc, n0 + 16 * i + r, k0 + 16, b1[i]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][t], b0[j][t], acc[i][j], 0, 0, 0);
        if (GRAM) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ssa[i] = __builtin_fmaf(a0[i][t], a0[i][t], ssa[i]);
                    ssb[i] = __builtin_fmaf(b0[i][t], b0[i][t], ssb[i]);
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a0[i][t] = a1[i][t];
                b0[i][t] = b1[i][t];
            }
    }
    }
    // D[row = 4 * (lane >> 4) + q][col = lane & 15]
    float inva[2] = {1.0f, 1.0f}, invb[2] = {1.0f, 1.0f};
    if (GRAM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // the four lanes (r, kq = 0..3) complete row r: the same value in all four
            ssa[i] += __shfl_xor(ssa[i], 16);
            ssa[i] += __shfl_xor(ssa[i], 32);
            ssb[i] += __shfl_xor(ssb[i], 16);
            ssb[i] += __shfl_xor(ssb[i], 32);
            inva[i] = 1.0f / (__builtin_sqrtf(ssa[i]) + gr.eps);
            invb[i] = 1.0f / (__builtin_sqrtf(ssb[i]) + gr.eps);
        }
    }
    const bool mirror = GRAM && blockIdx.x != blockIdx.y;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 16 * i + 4 * kq + q, col = n0 + 16 * j + r;
                float v = acc[i][j][q];
                if (GRAM) {
                    const float ir = __shfl(inva[i], 4 * kq + q);  // lane 4 kq + q holds the A row of this element
                    v = __builtin_fmaf(v, ir * invb[j], 1.0f) * 0.5f;
                }
                if (row < M && col < Nc) {
                    Cm[(long long)row * Nc + col] = v;
                    if (mirror) Cm[(long long)col * Nc + row] = v;
                }
            }
}

__device__ __forceinline__ float pkt_block_sum(float a, float *sh)
{
    const int tid = threadIdx.x;
    sh[tid] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// one workgroup per row i: row sums of Ks and Kt, the row's share of the loss, and c_i
__global__ __launch_bounds__(256) void pkt_rows_kernel(const float *__restrict__ Ks, const float *__restrict__ Kt, int N, float eps,
                                                       float *__restrict__ rs, float *__restrict__ rt, float *__restrict__ lossrow,
                                                       float *__restrict__ crow)
{
    __shared__ float sh[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    float a = 0.0f, b = 0.0f;
    for (int j = tid; j < N; j += 256) {
        a += Ks[(long long)i * N + j];
        b += Kt[(long long)i * N + j];
    }
    const float sum_s = pkt_block_sum(a, sh), sum_t = pkt_block_sum(b, sh);
    const float is = 1.0f / sum_s, it = 1.0f / sum_t;
    const float inv_n2 = 1.0f / ((float)N * (float)N);
    float l = 0.0f, c = 0.0f;
    for (int j = tid; j < N; j += 256) {
        const float ps = Ks[(long long)i * N + j] * is, pt = Kt[(long long)i * N + j] * it;
        l += pt * log1pf((pt - ps) / (ps + eps));  // = pt * log((pt + eps) / (ps + eps)), without forming a ratio next to 1
        c += (-pt / (ps + eps) * inv_n2) * ps;
    }
    l = pkt_block_sum(l, sh);
    c = pkt_block_sum(c, sh);
    if (tid == 0) {
        rs[i] = sum_s;
        rt[i] = sum_t;
        lossrow[i] = l;
        crow[i] = c;
    }
}

// loss = (sum_i lossrow[i]) / N^2 in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void pkt_loss_sum_kernel(const float *__restrict__ lossrow, int N, float *__restrict__ loss)
{
    __shared__ float sh[256];
    float a = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) a += lossrow[i];
    a = pkt_block_sum(a, sh);
    if (threadIdx.x == 0) loss[0] = a / ((float)N * (float)N);
}

// M_ij = dG_ij + dG_ji with dG_ij = gscale * (g_ij - c_i) / (2 r_i), g_ij = -Pt_ij / (Ps_ij + eps) / N^2
__global__ __launch_bounds__(256) void pkt_m_kernel(const float *__restrict__ Ks, const float *__restrict__ Kt, const float *__restrict__ rs,
                                                    const float *__restrict__ rt, const float *__restrict__ crow, int N, float eps,
                                                    const float *__restrict__ gscale, float *__restrict__ Mm)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * N) return;
    const int i = (int)(e / N), j = (int)(e - (long long)i * N);
    const float ks = Ks[e], kt = Kt[e];
    const float inv_n2 = 1.0f / ((float)N * (float)N);
    const float gs = gscale ? gscale[0] : 1.0f;
    auto half = [&](int a) {
        const float ps = ks / rs[a], pt = kt / rt[a];
        const float g = -pt / (ps + eps) * inv_n2;
        return (g - crow[a]) / rs[a] * 0.5f;
    };
    Mm[e] = gs * (half(i) + half(j));
}

// dx_i = (dxh_i - xh_i * (xh_i . dxh_i) * (|x_i| + eps) / |x_i|) / (|x_i| + eps)
__global__ __launch_bounds__(256) void pkt_norm_bwd_kernel(const float *__restrict__ xh, const float *__restrict__ dxh, const float *__restrict__ norm,
                                                           float *__restrict__ dx, int D, float eps)
{
    __shared__ float sh[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    float a = 0.0f;
    for (int d = tid; d < D; d += 256) a = __builtin_fmaf(xh[(long long)i * D + d], dxh[(long long)i * D + d], a);
    const float dot = pkt_block_sum(a, sh);
    const float nr = norm[i], inv = 1.0f / (nr + eps);
    const float f = nr > 0.0f ? dot * (nr + eps) / nr : 0.0f;
    for (int d = tid; d < D; d += 256) {
        const float v = (dxh[(long long)i * D + d] - xh[(long long)i * D + d] * f) * inv;
        dx[(long long)i * D + d] = nr > 0.0f ? v : 0.0f;
    }
}

static size_t pkt_align(size_t n) { return (n + 255) & ~(size_t)255; }

} // namespace aq

extern "C" {

size_t aq_pkt_workspace_bytes(int N, int Ds, int Dt)
{
    using namespace aq;
    if (N < 1 || Ds < 1 || Dt < 1) return 0;
    const size_t nn = pkt_align((size_t)N * N * 4);
    // Ks, Kt, M | xh_s, xhT_s, xh_t, dxh_s | norm_s, norm_t, rs, rt, lossrow, crow
    return 3 * nn + 3 * pkt_align((size_t)N * Ds * 4) + pkt_align((size_t)N * Dt * 4) + 6 * pkt_align((size_t)N * 4);
}

int aq_pkt_loss(const float *student, const float *teacher, int N, int Ds, int Dt, float eps, float *loss, const float *grad_scale,
                float *dstudent, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 1 && N <= 32768 && Ds >= 1 && Dt >= 1 && Ds <= (1 << 20) && Dt <= (1 << 20), -EINVAL, "aq_pkt_loss: bad sizes N=%d Ds=%d Dt=%d", N, Ds,
               Dt);
    AQ_REQUIRE(student && teacher && loss, -EFAULT, "aq_pkt_loss: NULL pointer");
    AQ_REQUIRE(eps > 0.0f, -EDOM, "aq_pkt_loss: eps must be positive");
    const size_t need = aq_pkt_workspace_bytes(N, Ds, Dt);
    AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, -ENOSPC,
               "aq_pkt_loss: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
    char *p = static_cast<char *>(workspace);
    auto take = [&](size_t bytes) { float *r = reinterpret_cast<float *>(p); p += pkt_align(bytes); return r; };
    const size_t nn = (size_t)N * N * 4;
    float *Ks = take(nn), *Kt = take(nn), *Mm = take(nn);
    float *xs = take((size_t)N * Ds * 4), *xsT = take((size_t)N * Ds * 4), *dxh = take((size_t)N * Ds * 4), *xt = take((size_t)N * Dt * 4);
    float *ns = take((size_t)N * 4), *nt = take((size_t)N * 4), *rs = take((size_t)N * 4), *rt = take((size_t)N * 4);
    float *lrow = take((size_t)N * 4), *crow = take((size_t)N * 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)xt;
    (void)nt;
    // both kernel matrices in one launch, straight from the raw features (normalisation inside the product)
    const dim3 g((N + 63) / 64, (N + 63) / 64, 2);
    const PktGram gr{{student, teacher}, {Ks, Kt}, {Ds, Dt}, eps};
    hipLaunchKernelGGL(pkt_gemm_nt_kernel<true>, g, dim3(256), 0, st, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, N, N, 0, gr);
    // the backward product needs the normalised student features, also transposed
    if (dstudent) hipLaunchKernelGGL(pkt_normalize_kernel, dim3(N), dim3(256), 0, st, student, xs, xsT, ns, N, Ds, eps);
    hipLaunchKernelGGL(pkt_rows_kernel, dim3(N), dim3(256), 0, st, Ks, Kt, N, eps, rs, rt, lrow, crow);
    hipLaunchKernelGGL(pkt_loss_sum_kernel, dim3(1), dim3(256), 0, st, lrow, N, loss);
    if (dstudent) {
        const long long ne = (long long)N * N;
        hipLaunchKernelGGL(pkt_m_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, Ks, Kt, rs, rt, crow, N, eps, grad_scale, Mm);
        const dim3 g2((Ds + 63) / 64, (N + 63) / 64);
        hipLaunchKernelGGL(pkt_gemm_nt_kernel<false>, g2, dim3(256), 0, st, Mm, xsT, dxh, N, Ds, N, PktGram{});  // dxh = M xh
        hipLaunchKernelGGL(pkt_norm_bwd_kernel, dim3(N), dim3(256), 0, st, xs, dxh, ns, dstudent, Ds, eps);
    }
    return check_launch("aq_pkt_loss");
}

/* the kernel matrix alone: K = (xh xh^T + 1) / 2 of row-normalised features (what a caller inspecting the similarities,
 * or timing the dense contraction, wants); workspace: N * D floats + N floats, 256-byte aligned pieces */
int aq_pkt_kernel_matrix(const float *feat, int N, int D, float eps, float *K, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 1 && N <= 32768 && D >= 1 && D <= (1 << 20), -EINVAL, "aq_pkt_kernel_matrix: bad sizes N=%d D=%d", N, D);
    AQ_REQUIRE(feat && K, -EFAULT, "aq_pkt_kernel_matrix: NULL pointer");
    (void)workspace;
    (void)workspace_bytes;  // (kept in the signature: earlier versions normalised into it)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PktGram gr{{feat, feat}, {K, K}, {D, D}, eps};
    hipLaunchKernelGGL(pkt_gemm_nt_kernel<true>, dim3((N + 63) / 64, (N + 63) / 64, 1), dim3(256), 0, st, (const float *)nullptr, (const float *)nullptr,
                       (float *)nullptr, N, N, 0, gr);
    return check_launch("aq_pkt_kernel_matrix");
}

} // extern "C"
