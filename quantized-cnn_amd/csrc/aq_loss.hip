// Tail of the training step: `loss = criterion(logits, y) + model_loss(config, beta, model)`
// (adp_qtz_fan.py:288-289; criterion = nn.CrossEntropyLoss(), mean over the batch; model_loss :737-754).
// aq_sumsq:   sum of squares of many small tensors in one launch (the "weight" terms of model_loss).
// aq_ce_loss: one workgroup computes the cross entropy, its gradient w.r.t. the logits (what autograd
//             would hand back for d loss = 1) and assembles {ce, reg, loss} from the sums of squares, so
//             the dozen tiny stock launches of this tail become two.  Fixed summation order, no atomics.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

constexpr int SUMSQ_MAX = 32;
struct SumsqTab {
    const float *p[SUMSQ_MAX];
    long long n[SUMSQ_MAX];
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

__global__ __launch_bounds__(256) void sumsq_kernel(const SumsqTab t)
{
    __shared__ float sh[256];
    const int i = blockIdx.x;
    float a = 0.0f;
    for (long long k = threadIdx.x; k < t.n[i]; k += 256) a = __builtin_fmaf(t.p[i][k], t.p[i][k], a);
    const float s = block_tree_sum(a, sh);
    if (threadIdx.x == 0) t.out[i] = s;
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

} // namespace aq

extern "C" {

int aq_sumsq(const float *const *tensors, const long long *sizes, int n, float *out, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && (n == 0 || (tensors && sizes && out)), -EINVAL, "aq_sumsq: bad arguments");
    for (int base = 0; base < n; base += SUMSQ_MAX) {
        SumsqTab t = {};
        const int m = n - base < SUMSQ_MAX ? n - base : SUMSQ_MAX;
        for (int i = 0; i < m; ++i) {
            AQ_REQUIRE(sizes[base + i] >= 0 && (sizes[base + i] == 0 || tensors[base + i]), -EINVAL, "aq_sumsq: tensor %d is NULL", base + i);
            t.p[i] = tensors[base + i];
            t.n[i] = sizes[base + i];
        }
        t.out = out + base;
        hipLaunchKernelGGL(sumsq_kernel, dim3(m), dim3(256), 0, static_cast<hipStream_t>(stream), t);
        if (int rc = check_launch("aq_sumsq")) return rc;
    }
    return 0;
}

int aq_ce_loss(const float *logits, const long long *labels, int N, int C, const float *sumsq_a, int na, const float *sumsq_b, int nb,
               float l2_reg, float grad_scale, float *dlogits, float *out, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(N >= 1 && C >= 1 && na >= 0 && nb >= 0, -EINVAL, "aq_ce_loss: bad sizes N=%d C=%d", N, C);
    AQ_REQUIRE(logits && labels && out && (na == 0 || sumsq_a) && (nb == 0 || sumsq_b), -EFAULT, "aq_ce_loss: NULL pointer");
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels, N, C, sumsq_a, na, sumsq_b, nb,
                       l2_reg, grad_scale, dlogits, out);
    return check_launch("aq_ce_loss");
}

} // extern "C"
