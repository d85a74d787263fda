// Development probe: launch shapes of the soft weight-synthesis forward pass (L = 32, 8 lanes per row, fast arithmetic,
// streaming loads): one 64-row batch per wavefront (the library's shape) against grid-stride forms with the next
// batch's loads issued before the current batch's arithmetic.  Build: hipcc --offload-arch=gfx950 -O3 -o k1_variants k1_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float gsum8(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)); // row_half_mirror
    return v;
}

template <int U>
__device__ __forceinline__ void load_batch(const float *__restrict__ p, long long row0, int lane, f4 (&t)[U])
{
    const f4 *src = reinterpret_cast<const f4 *>(p + row0 * 32) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = __builtin_nontemporal_load(src + 64 * u);
}

// rows row0 + 8u + gq (u < U): lane (gq, j) holds elements 4j..4j+3 of its rows; result: lane (gq, j) gets W of row row0 + 8j + gq (j < U)
template <int U>
__device__ __forceinline__ float compute_batch(const f4 (&t)[U], const float (&q4)[4], float bl2e, int j)
{
    float myW = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float ss = gsum8(fmaf(t[u].w, t[u].w, fmaf(t[u].z, t[u].z, fmaf(t[u].y, t[u].y, t[u].x * t[u].x))));
        const float rn = bl2e * __builtin_amdgcn_rsqf(ss);
        const float e0 = __builtin_amdgcn_exp2f(t[u].x * rn), e1 = __builtin_amdgcn_exp2f(t[u].y * rn), e2 = __builtin_amdgcn_exp2f(t[u].z * rn),
                    e3 = __builtin_amdgcn_exp2f(t[u].w * rn);
        const float den = gsum8(((e0 + e1) + e2) + e3);
        const float num = gsum8(fmaf(e3, q4[3], fmaf(e2, q4[2], fmaf(e1, q4[1], e0 * q4[0]))));
        const float w = num * __builtin_amdgcn_rcpf(den);
        myW = (j == u) ? w : myW;
    }
    return myW;
}

// MODE 0: one batch per wavefront.  MODE 1: grid-stride, no prefetch.  MODE 2: grid-stride, next batch's loads issued before the arithmetic.
template <int U, int MODE, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k1_kernel(const float *__restrict__ p, const float *__restrict__ q, float *__restrict__ W, long long nbatch, float beta,
                                                 float *__restrict__ part)
{
    const int lane = threadIdx.x & 63, j = lane & 7, gq = lane >> 3;
    const long long wave0 = (long long)blockIdx.x * WPB + (threadIdx.x >> 6), nw = (long long)gridDim.x * WPB;
    float q4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q4[k] = q[4 * j + k];
    const float bl2e = beta * 1.44269504f;
    float acc = 0.f;
    if (MODE == 0) {
        if (wave0 < nbatch) {
            f4 t[U];
            load_batch<U>(p, wave0 * (8 * U), lane, t);
            const float w = compute_batch<U>(t, q4, bl2e, j);
            if (j < U) W[wave0 * (8 * U) + 8 * j + gq] = w;
            acc = j < U ? w * w : 0.f;
        }
    } else if (MODE == 1) {
        for (long long b = wave0; b < nbatch; b += nw) {
            f4 t[U];
            load_batch<U>(p, b * (8 * U), lane, t);
            const float w = compute_batch<U>(t, q4, bl2e, j);
            if (j < U) W[b * (8 * U) + 8 * j + gq] = w;
            acc += j < U ? w * w : 0.f;
        }
    } else {
        f4 t[U], n[U];
        long long b = wave0;
        if (b < nbatch) load_batch<U>(p, b * (8 * U), lane, t);
        for (; b < nbatch; b += nw) {
            const bool more = b + nw < nbatch;
            if (more) load_batch<U>(p, (b + nw) * (8 * U), lane, n);
            const float w = compute_batch<U>(t, q4, bl2e, j);
            if (j < U) W[b * (8 * U) + 8 * j + gq] = w;
            acc += j < U ? w * w : 0.f;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = n[u];
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) part[wave0] = acc;
}

template <typename F>
static float timeit(F f, int it = 20)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < it; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / it;
}

int main(int argc, char **argv)
{
    const long long rows = (argc > 1 ? atoll(argv[1]) : 250000000LL) / 32 / 128 * 128;
    float *p, *q, *W, *part;
    CK(hipMalloc(&p, rows * 128)); CK(hipMalloc(&q, 128)); CK(hipMalloc(&W, rows * 4)); CK(hipMalloc(&part, 64 << 20));
    {
        std::vector<float> h(1 << 22);
        unsigned s = 12345;
        for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (1.0f / (1 << 24)); }
        for (long long o = 0; o < rows * 32; o += (long long)h.size()) {
            const long long c = std::min<long long>(h.size(), rows * 32 - o);
            CK(hipMemcpy(p + o, h.data(), c * 4, hipMemcpyHostToDevice));
        }
        float hq[32];
        for (int i = 0; i < 32; ++i) hq[i] = (i - 15.5f) / 16.f;
        CK(hipMemcpy(q, hq, 128, hipMemcpyHostToDevice));
    }
    printf("rows = %lld\n", rows);
#define RUN(U, MODE, GRID, label) { const long long nb = rows / (8 * U); const int grid = MODE ? GRID : (int)((nb + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((k1_kernel<U, MODE>), dim3(grid), dim3(256), 0, 0, p, q, W, nb, 4.0f, part); }); \
    printf("k1 %-38s %.4f ms  %.0f GB/s\n", label, ms, rows * 132.0 * 1e-9 / (ms * 1e-3)); }
#define RUNW(U, WPB, label) { const long long nb = rows / (8 * U); const int grid = (int)((nb + WPB - 1) / WPB); \
    float ms = timeit([&] { hipLaunchKernelGGL((k1_kernel<U, 0, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, p, q, W, nb, 4.0f, part); }); \
    printf("k1 %-38s %.4f ms  %.0f GB/s\n", label, ms, rows * 132.0 * 1e-9 / (ms * 1e-3)); }
    RUNW(8, 4, "64 rows/wave, 4 waves/wg (library)");
    RUNW(4, 4, "32 rows/wave, 4 waves/wg");
    RUNW(2, 4, "16 rows/wave, 4 waves/wg");
    RUNW(1, 4, " 8 rows/wave, 4 waves/wg");
    RUNW(4, 1, "32 rows/wave, 1 wave/wg");
    RUNW(4, 2, "32 rows/wave, 2 waves/wg");
    RUNW(4, 8, "32 rows/wave, 8 waves/wg");
    RUNW(2, 8, "16 rows/wave, 8 waves/wg");
    RUNW(2, 16, "16 rows/wave, 16 waves/wg");
    RUNW(8, 1, "64 rows/wave, 1 wave/wg");
    RUNW(8, 8, "64 rows/wave, 8 waves/wg");
    RUNW(4, 4, "32 rows/wave, 4 waves/wg (again)");
    RUNW(8, 4, "64 rows/wave, 4 waves/wg (again)");
    return 0;
}
