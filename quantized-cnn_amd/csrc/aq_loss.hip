// Tail of the training step: `loss = criterion(logits, y) + model_loss(config, beta, model)`
// (adp_qtz_fan.py:288-289; criterion = nn.CrossEntropyLoss(), mean over the batch; model_loss :737-754).
// aq_sumsq:   sum of squares of many small tensors in one launch (the "weight" terms of model_loss).
// aq_ce_loss: one workgroup computes the cross entropy, its gradient w.r.t. the logits (what autograd
//             would hand back for d loss = 1) and assembles {ce, reg, loss} from the sums of squares, so
//             the dozen tiny stock launches of this tail become two.  Fixed summation order, no atomics.
#include <algorithm>

#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

constexpr int SUMSQ_MAX = 32;
constexpr int SUMSQ_CHUNK = 256 * 64;  // elements per workgroup: a tensor beyond that is shared by several workgroups
struct SumsqTab {
    const float *p[SUMSQ_MAX];
    long long n[SUMSQ_MAX];
    int blk0[SUMSQ_MAX + 1];   // first workgroup of every tensor
    double *part[SUMSQ_MAX];   // per-chunk partial sums of the tensors with more than one chunk
    int ntensors;
    float *out;
};

__device__ inline float block_tree_sum(float v, float *sh)  // 256 threads, fixed halving tree; result in thread 0
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}
__device__ inline double block_tree_sum(double v, double *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}

// Per thread fp32 fma chains over at most 64 elements (one chain, or four chains of 16 for a whole aligned chunk read with
// 16-byte loads: a tensor's result depends on its alignment, not on anything else), everything beyond that in double: the sum of squares of a
// 2.4 M-element tensor is then as exact as its fp32 result can say (the reference's torch.sum is a pairwise fp32 sum).
__global__ __launch_bounds__(256) void sumsq_kernel(const SumsqTab t)
{
    __shared__ double sh[256];
    int i = 0;
    for (int k = 1; k < t.ntensors; ++k)
        if ((int)blockIdx.x >= t.blk0[k]) i = k;
    const int c = (int)blockIdx.x - t.blk0[i];
    const long long lo = (long long)c * SUMSQ_CHUNK, hi = min(lo + SUMSQ_CHUNK, t.n[i]);
    const float *p = t.p[i];
    float a = 0.0f;
    if (hi - lo == SUMSQ_CHUNK && (((uintptr_t)(p + lo)) & 15) == 0) {
        // a whole, 16-byte aligned chunk: 16 loads of 16 bytes per thread, all in flight, four fma chains of 16 elements
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *q = reinterpret_cast<const f4 *>(p + lo) + threadIdx.x;
        f4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = q[256 * j];
        f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c.x = __builtin_fmaf(v[j].x, v[j].x, c.x); c.y = __builtin_fmaf(v[j].y, v[j].y, c.y);
            c.z = __builtin_fmaf(v[j].z, v[j].z, c.z); c.w = __builtin_fmaf(v[j].w, v[j].w, c.w);
        }
        a = (c.x + c.y) + (c.z + c.w);
    } else {
        for (long long k = lo + threadIdx.x; k < hi; k += 256) a = __builtin_fmaf(p[k], p[k], a);
    }
    const double s = block_tree_sum((double)a, sh);
    if (threadIdx.x == 0) {
        if (t.blk0[i + 1] - t.blk0[i] == 1) t.out[i] = (float)s;
        else t.part[i][c] = s;
    }
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const SumsqTab t)
{
    __shared__ double sh[256];
    const int i = blockIdx.x;
    const int nchunk = t.blk0[i + 1] - t.blk0[i];
    if (nchunk <= 1) return;
    double a = 0.0;
    for (int k = threadIdx.x; k < nchunk; k += 256) a += t.part[i][k];
    const double s = block_tree_sum(a, sh);
    if (threadIdx.x == 0) t.out[i] = (float)s;
}

__global__ __launch_bounds__(256) void ce_loss_kernel(const float *__restrict__ logits, const long long *__restrict__ labels, int N, int C,
                                                      const float *__restrict__ ssa, int na, const float *__restrict__ ssb, int nb,
                                                      float l2_reg, float grad_scale, float *__restrict__ dlogits, float *__restrict__ out)
{
    __shared__ float sh[256];
    float part = 0.0f;
    const float rN = 1.0f / (float)N, rNg = rN * grad_scale;
    for (int r = threadIdx.x; r < N; r += 256) {
        const float *x = logits + (size_t)r * C;
        float m = x[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - m);
        const float lse = m + logf(se);
        const long long y = labels[r];
        const bool ok = y >= 0 && y < C;
        part += ok ? lse - x[ok ? y : 0] : __builtin_nanf("");  // a label outside [0, C) poisons the loss
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[(size_t)r * C + c] = (expf(x[c] - lse) - ((long long)c == y ? 1.0f : 0.0f)) * rNg;
    }
    const float tot = block_tree_sum(part, sh);
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int i = 0; i < nb; ++i) ss += ssb[i];  // dense "weight" tensors first, then the quantized layers
        for (int i = 0; i < na; ++i) ss += ssa[i];  // (the order of adp_qtz_fan.py:737-754 does not matter to 1e-7)
        const float ce = tot * rN, reg = l2_reg * ss;
        out[0] = ce;
        out[1] = reg;
        out[2] = ce + reg;
    }
}

// Many classes or images (ImageNet-shaped heads: 32 x 1000 logits): one workgroup per image computes that image's
// log-sum-exp, its gradient row and its loss term; ce_finish_kernel sums the terms in a fixed order and assembles
// {ce, reg, loss} as ce_loss_kernel does.  Fixed halving trees throughout: deterministic.
__global__ __launch_bounds__(256) void ce_rows_kernel(const float *__restrict__ logits, const long long *__restrict__ labels, int N, int C,
                                                      float grad_scale, float *__restrict__ dlogits, float *__restrict__ rowloss)
{
    __shared__ float sh[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float *x = logits + (size_t)r * C;
    float m = -__builtin_inff();
    for (int c = tid; c < C; c += 256) m = fmaxf(m, x[c]);
    sh[tid] = m;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) sh[tid] = fmaxf(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    m = sh[0];
    __syncthreads();
    float se = 0.0f;
    for (int c = tid; c < C; c += 256) se += expf(x[c] - m);
    se = block_tree_sum(se, sh);
    __syncthreads();
    const float lse = m + logf(se);
    const long long y = labels[r];
    const bool ok = y >= 0 && y < C;
    if (tid == 0) rowloss[r] = ok ? lse - x[ok ? y : 0] : __builtin_nanf("");
    if (dlogits) {
        const float rNg = grad_scale / (float)N;
        for (int c = tid; c < C; c += 256) dlogits[(size_t)r * C + c] = (expf(x[c] - lse) - ((long long)c == y ? 1.0f : 0.0f)) * rNg;
    }
}
__global__ __launch_bounds__(256) void ce_finish_kernel(const float *__restrict__ rowloss, int N, const float *__restrict__ ssa, int na,
                                                        const float *__restrict__ ssb, int nb, float l2_reg, float *__restrict__ out)
{
    __shared__ float sh[256];
    float part = 0.0f;
    for (int r = threadIdx.x; r < N; r += 256) part += rowloss[r];
    const float tot = block_tree_sum(part, sh);
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int i = 0; i < nb; ++i) ss += ssb[i];
        for (int i = 0; i < na; ++i) ss += ssa[i];
        const float ce = tot / (float)N, reg = l2_reg * ss;
        out[0] = ce;
        out[1] = reg;
        out[2] = ce + reg;
    }
}
constexpr long long CE_ONE_WORKGROUP = 16384;  // logits up to this many elements: the single-workgroup kernel (ResNet-20: 128 x 10)

} // namespace aq

extern "C" {

size_t aq_sumsq_workspace_bytes(const long long *sizes, int n)
{
    size_t bytes = 0;
    for (int i = 0; sizes && i < n; ++i) {
        const long long chunks = (sizes[i] + aq::SUMSQ_CHUNK - 1) / aq::SUMSQ_CHUNK;
        if (chunks > 1) bytes += (size_t)chunks * sizeof(double);
    }
    return bytes;
}

int aq_sumsq(const float *const *tensors, const long long *sizes, int n, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && (n == 0 || (tensors && sizes && out)), -EINVAL, "aq_sumsq: bad arguments");
    const size_t need = aq_sumsq_workspace_bytes(sizes, n);
    AQ_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && ((uintptr_t)workspace & 7) == 0), -ENOSPC,
               "aq_sumsq: workspace of %zu bytes (8-byte aligned) needed for the tensors of more than %d elements, %zu given", need, SUMSQ_CHUNK,
               workspace_bytes);
    double *ws = static_cast<double *>(workspace);
    for (int base = 0; base < n; base += SUMSQ_MAX) {
        SumsqTab t = {};
        const int m = n - base < SUMSQ_MAX ? n - base : SUMSQ_MAX;
        long long blocks = 0;
        bool finish = false;
        for (int i = 0; i < m; ++i) {
            AQ_REQUIRE(sizes[base + i] >= 0 && (sizes[base + i] == 0 || tensors[base + i]), -EINVAL, "aq_sumsq: tensor %d is NULL", base + i);
            t.p[i] = tensors[base + i];
            t.n[i] = sizes[base + i];
            t.blk0[i] = (int)blocks;
            const long long chunks = std::max<long long>(1, (sizes[base + i] + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK);
            if (chunks > 1) {
                t.part[i] = ws;
                ws += chunks;
                finish = true;
            }
            blocks += chunks;
            AQ_REQUIRE(blocks < (1LL << 31), -E2BIG, "aq_sumsq: too many elements for one launch");
        }
        t.blk0[m] = (int)blocks;
        t.ntensors = m;
        t.out = out + base;
        hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
        if (finish) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(m), dim3(256), 0, static_cast<hipStream_t>(stream), t);
        if (int rc = check_launch("aq_sumsq")) return rc;
    }
    return 0;
}

size_t aq_ce_loss_workspace_bytes(int N, int C) { return (long long)N * C > aq::CE_ONE_WORKGROUP ? (size_t)N * sizeof(float) : 0; }

int aq_ce_loss(const float *logits, const long long *labels, int N, int C, const float *sumsq_a, int na, const float *sumsq_b, int nb,
               float l2_reg, float grad_scale, float *dlogits, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 1 && C >= 1 && na >= 0 && nb >= 0, -EINVAL, "aq_ce_loss: bad sizes N=%d C=%d", N, C);
    AQ_REQUIRE(logits && labels && out && (na == 0 || sumsq_a) && (nb == 0 || sumsq_b), -EFAULT, "aq_ce_loss: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t need = aq_ce_loss_workspace_bytes(N, C);
    if (need == 0) {
        hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, st, logits, labels, N, C, sumsq_a, na, sumsq_b, nb, l2_reg, grad_scale, dlogits, out);
        return check_launch("aq_ce_loss");
    }
    AQ_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 3) == 0, -ENOSPC, "aq_ce_loss: workspace of %zu bytes needed for %d x %d logits, %zu given",
               need, N, C, workspace_bytes);
    float *rowloss = static_cast<float *>(workspace);
    hipLaunchKernelGGL(ce_rows_kernel, dim3(N), dim3(256), 0, st, logits, labels, N, C, grad_scale, dlogits, rowloss);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, rowloss, N, sumsq_a, na, sumsq_b, nb, l2_reg, out);
    return check_launch("aq_ce_loss");
}

} // extern "C"
