// Adam update of many small tensors in one launch: the two small parameter groups of the reference's training
// loop, optimizer_q (the q_level vectors) and optimizer (conv0, BatchNorm, linear bias), which it steps with
// two torch.optim.Adam objects (adp_qtz_fan.py:223-242, 296-304).  One workgroup per tensor; the arithmetic is
// the one of the Adam update fused into the synthesis-gradient kernel (aq_synth.hip):
//   t = step + 1;  m += (1-beta1)*(g-m);  v = beta2*v + (1-beta2)*g*g;
//   p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)
// with the step-dependent coefficients derived in double.  An item's `reg` adds reg*p to its gradient first
// (the regulariser 2*l2_reg*w of the dense "weight" tensors, adp_qtz_fan.py:737-754) and writes the sum back
// to g, so that g holds the gradient of the whole loss as autograd would have left it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>

#include "../../include/aq_hip.h"
#include "aq_common.h"

namespace aq {

constexpr int ADAM_MAX_ITEMS = 48;  // per launch (the table travels in the kernel arguments)

struct AdamItem {
    float *p, *g, *m, *v, *step;
    int n;
    float reg;
};
struct AdamTab {
    AdamItem it[ADAM_MAX_ITEMS];
    double lr, b1, b2;
    float omb1, b2f, omb2, eps;
};

__global__ __launch_bounds__(256) void adam_items_kernel(const AdamTab t)
{
    __shared__ float coef[2];
    const AdamItem it = t.it[blockIdx.x];
    if (threadIdx.x == 0) {
        const float tnew = it.step[0] + 1.0f;
        const double bc1 = 1.0 - pow(t.b1, (double)tnew), bc2 = 1.0 - pow(t.b2, (double)tnew);
        coef[0] = (float)(t.lr / bc1);
        coef[1] = (float)(1.0 / sqrt(bc2));
        it.step[0] = tnew;  // this workgroup is the only reader of its step counter (distinct per item)
    }
    __syncthreads();
    const float step_size = coef[0], rbc2 = coef[1];
    for (int i = threadIdx.x; i < it.n; i += 256) {
        const float p = it.p[i];
        float g = it.g[i];
        if (it.reg != 0.0f) {
            g = __builtin_fmaf(it.reg, p, g);
            it.g[i] = g;
        }
        const float m0 = it.m[i], v0 = it.v[i];
        const float m = __builtin_fmaf(t.omb1, g - m0, m0);
        const float v = __builtin_fmaf(t.omb2 * g, g, t.b2f * v0);
        const float denom = __builtin_fmaf(sqrtf(v), rbc2, t.eps);
        it.p[i] = p - step_size * (m / denom);
        it.m[i] = m;
        it.v[i] = v;
    }
}

// Tensors of more than one workgroup's worth of elements: a workgroup per ADAM_CHUNK elements, the step counters read as
// the previous step left them and advanced by adam_bump_kernel afterwards (next launch on the stream).
constexpr int ADAM_CHUNK = 256 * 16;
struct AdamBigTab {
    AdamItem it[ADAM_MAX_ITEMS];
    int blk0[ADAM_MAX_ITEMS + 1];
    int nitems;
    double lr, b1, b2;
    float omb1, b2f, omb2, eps;
};
__global__ __launch_bounds__(256) void adam_chunks_kernel(const AdamBigTab t)
{
    int ii = 0;
    for (int i = 1; i < t.nitems; ++i)
        if ((int)blockIdx.x >= t.blk0[i]) ii = i;
    const AdamItem it = t.it[ii];
    const float tnew = it.step[0] + 1.0f;
    const double bc1 = 1.0 - pow(t.b1, (double)tnew), bc2 = 1.0 - pow(t.b2, (double)tnew);
    const float step_size = (float)(t.lr / bc1), rbc2 = (float)(1.0 / sqrt(bc2));
    const int lo = ((int)blockIdx.x - t.blk0[ii]) * ADAM_CHUNK;
    const int hi = min(lo + ADAM_CHUNK, it.n);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const float p = it.p[i];
        float g = it.g[i];
        if (it.reg != 0.0f) {
            g = __builtin_fmaf(it.reg, p, g);
            it.g[i] = g;
        }
        const float m0 = it.m[i], v0 = it.v[i];
        const float m = __builtin_fmaf(t.omb1, g - m0, m0);
        const float v = __builtin_fmaf(t.omb2 * g, g, t.b2f * v0);
        const float denom = __builtin_fmaf(sqrtf(v), rbc2, t.eps);
        it.p[i] = p - step_size * (m / denom);
        it.m[i] = m;
        it.v[i] = v;
    }
}
__global__ void adam_bump_kernel(const AdamBigTab t)
{
    const int i = threadIdx.x;
    if (i < t.nitems) t.it[i].step[0] = t.it[i].step[0] + 1.0f;
}

} // namespace aq

extern "C" int aq_adam_step(const aq_adam_item *items, int nitems, double lr, double beta1, double beta2, double eps, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(nitems >= 0 && (nitems == 0 || items), -EINVAL, "aq_adam_step: bad item list");
    AQ_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, -EDOM, "aq_adam_step: lr=%g betas=(%g, %g) eps=%g out of range",
               lr, beta1, beta2, eps);
    for (int i = 0; i < nitems; ++i) {
        const aq_adam_item &q = items[i];
        AQ_REQUIRE(q.n >= 0 && q.n < (1LL << 31) && (q.n == 0 || (q.p && q.g && q.m && q.v)) && q.step, -EINVAL, "aq_adam_step: bad item %d", i);
        for (int k = 0; k < i; ++k)
            AQ_REQUIRE(items[k].step != q.step, -EINVAL, "aq_adam_step: items %d and %d share a step counter (one per tensor, as torch.optim.Adam keeps them)",
                       k, i);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < nitems; base += ADAM_MAX_ITEMS) {
        AdamTab t = {};
        const int n = std::min(ADAM_MAX_ITEMS, nitems - base);
        for (int i = 0; i < n; ++i) {
            const aq_adam_item &q = items[base + i];
            t.it[i] = AdamItem{q.p, q.g, q.m, q.v, q.step, (int)q.n, q.reg};
        }
        t.lr = lr; t.b1 = beta1; t.b2 = beta2;
        t.omb1 = (float)(1.0 - beta1); t.b2f = (float)beta2; t.omb2 = (float)(1.0 - beta2); t.eps = (float)eps;
        long long most = 0;
        for (int i = 0; i < n; ++i) most = std::max<long long>(most, items[base + i].n);
        if (most > 4 * ADAM_CHUNK) {  // some tensor would keep one workgroup busy for long: chunks over many workgroups
            AdamBigTab b = {};
            long long blocks = 0;
            for (int i = 0; i < n; ++i) {
                b.it[i] = t.it[i];
                b.blk0[i] = (int)blocks;
                blocks += std::max<long long>(1, (items[base + i].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
            }
            b.blk0[n] = (int)blocks;
            b.nitems = n;
            b.lr = lr; b.b1 = beta1; b.b2 = beta2;
            b.omb1 = t.omb1; b.b2f = t.b2f; b.omb2 = t.omb2; b.eps = t.eps;
            hipLaunchKernelGGL(adam_chunks_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
            hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, st, b);
            if (int rc = check_launch("aq_adam_step")) return rc;
            continue;
        }
        hipLaunchKernelGGL(adam_items_kernel, dim3(n), dim3(256), 0, st, t);
        if (int rc = check_launch("aq_adam_step")) return rc;
    }
    return 0;
}
