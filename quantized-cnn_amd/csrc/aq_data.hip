// Input pipeline on the device.  The reference builds every training image on CPU workers with
//   transforms.RandomCrop(32, padding=4), RandomHorizontalFlip(), ToTensor(),
//   Normalize((0.4915, 0.4821, 0.4462), (0.2472, 0.2437, 0.2617))              (adp_qtz_fan.py:164-169)
// from the raw CIFAR-10 array (N, 32, 32, 3) uint8 (adp_qtz_fan.py:452-530).  Here the raw array stays in
// HBM and one launch produces a whole normalised NCHW fp32 batch: gather by index, zero padding, crop
// offset, mirror, u8 -> float / 255, (v - mean) / std, with the same fp32 operations in the same order
// as ToTensor + Normalize.  Random numbers (indices, offsets, flips) come from the caller.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

struct AugArgs {
    const uint8_t *data;   // [Nimg, H, W, C] channels last
    const int *index;      // [B] image per batch slot (NULL: slot b takes image b)
    const int *off_y, *off_x;  // [B] crop offset inside the padded image, 0..2*pad (NULL: pad, i.e. centred)
    const uint8_t *flip;   // [B] nonzero = mirror horizontally after the crop (NULL: never)
    float *out;            // [B, C, H, W]
    int B, H, W, C, pad;
    float mean[4], std[4];
};

__global__ __launch_bounds__(256) void augment_kernel(const AugArgs a)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)a.H * a.W;
    if (t >= (long long)a.B * per) return;
    const int b = t / per;
    const int r = t - (long long)b * per;
    const int y = r / a.W, x = r - y * a.W;
    const long long img = a.index ? a.index[b] : b;
    const int oy = a.off_y ? a.off_y[b] : a.pad, ox = a.off_x ? a.off_x[b] : a.pad;
    const int xs = (a.flip && a.flip[b]) ? a.W - 1 - x : x;
    const int sy = y + oy - a.pad, sx = xs + ox - a.pad;
    const bool in = sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
    const uint8_t *p = a.data + ((img * a.H + sy) * a.W + sx) * a.C;
    for (int c = 0; c < a.C; ++c) {
        const float v = in ? (float)p[c] / 255.0f : 0.0f;
        a.out[(((long long)b * a.C + c) * a.H + y) * a.W + x] = (v - a.mean[c]) / a.std[c];
    }
}

constexpr int ABSMAX_MAX = 32;
struct AbsmaxTab {
    const float *p[ABSMAX_MAX];
    long long n[ABSMAX_MAX];
    float *out;
};

// max |v| per tensor (exact, order-independent); NaN inputs are ignored like fmaxf does.  Up to ABSMAX_SPLIT workgroups
// share a tensor: each takes every gridDim.x-th run of 1024 elements and merges its maximum into out[i] with an
// integer atomic max on the bit pattern (non-negative floats order like unsigned integers; out starts at +0), so the
// result does not depend on arrival order.  (One workgroup per tensor took 2.1 ms on a 2.4 M-weight layer.)
constexpr int ABSMAX_SPLIT = 128;
__global__ __launch_bounds__(256) void absmax_kernel(const AbsmaxTab t)
{
    __shared__ float sh[4];
    const int i = blockIdx.y;
    const long long n = t.n[i];
    const float *__restrict__ p = t.p[i];
    float m = 0.0f;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(p[k]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        if (r > 0.0f) atomicMax(reinterpret_cast<unsigned *>(t.out + i), __builtin_bit_cast(unsigned, r));
    }
}

} // namespace aq

extern "C" {

int aq_augment_batch(const uint8_t *data, long long n_images, const int *index, const int *off_y, const int *off_x, const uint8_t *flip, int B,
                     int H, int W, int C, int pad, const float *mean, const float *std, float *out, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 1 && C <= 4 && pad >= 0 && n_images >= 0, -EINVAL,
               "aq_augment_batch: bad sizes B=%d H=%d W=%d C=%d pad=%d (C must be 1..4)", B, H, W, C, pad);
    AQ_REQUIRE(mean && std, -EFAULT, "aq_augment_batch: NULL mean / std (host arrays of C floats)");
    if (B == 0) return 0;
    AQ_REQUIRE(data && out, -EFAULT, "aq_augment_batch: NULL pointer");
    AQ_REQUIRE(index || B <= n_images, -EINVAL, "aq_augment_batch: %d batch slots but %lld images and no index", B, n_images);
    AugArgs a = {};
    a.data = data; a.index = index; a.off_y = off_y; a.off_x = off_x; a.flip = flip; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C = C; a.pad = pad;
    for (int c = 0; c < C; ++c) {
        AQ_REQUIRE(std[c] != 0.0f, -EDOM, "aq_augment_batch: std[%d] is zero", c);
        a.mean[c] = mean[c];
        a.std[c] = std[c];
    }
    const long long total = (long long)B * H * W;
    AQ_REQUIRE((total + 255) / 256 < (1LL << 31), -E2BIG, "aq_augment_batch: batch too large");
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("aq_augment_batch");
}

int aq_absmax(const float *const *tensors, const long long *sizes, int n, float *out, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && (n == 0 || (tensors && sizes && out)), -EINVAL, "aq_absmax: bad arguments");
    for (int base = 0; base < n; base += ABSMAX_MAX) {
        AbsmaxTab t = {};
        const int m = n - base < ABSMAX_MAX ? n - base : ABSMAX_MAX;
        for (int i = 0; i < m; ++i) {
            AQ_REQUIRE(sizes[base + i] >= 0 && (sizes[base + i] == 0 || tensors[base + i]), -EINVAL, "aq_absmax: tensor %d is NULL", base + i);
            t.p[i] = tensors[base + i];
            t.n[i] = sizes[base + i];
        }
        t.out = out + base;
        long long most = 0;
        for (int i = 0; i < m; ++i) most = most > t.n[i] ? most : t.n[i];
        const int split = (int)(most / 4096 < 1 ? 1 : (most / 4096 > ABSMAX_SPLIT ? ABSMAX_SPLIT : most / 4096));
        hipError_t e = hipMemsetAsync(out + base, 0, sizeof(float) * m, static_cast<hipStream_t>(stream));
        AQ_REQUIRE(e == hipSuccess, -EIO, "aq_absmax: clearing the result failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(absmax_kernel, dim3(split, m), dim3(256), 0, static_cast<hipStream_t>(stream), t);
        if (int rc = check_launch("aq_absmax")) return rc;
    }
    return 0;
}

} // extern "C"
