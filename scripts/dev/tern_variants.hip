// Development probe: launch-shape variants of the ternary quantiser's two streaming passes (read-only statistics pass,
// code-emitting pass) and of the gradient pass, timed with HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -o tern_variants tern_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wmax(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}


// value of quad lane (CTRL's lane) of the row this lane takes: r0..r3 are this lane's four rows, j = lane & 3 picks the row
template <int CTRL>
__device__ __forceinline__ unsigned quad_pick(unsigned r0, unsigned r1, unsigned r2, unsigned r3, int j)
{
    const int a0 = __builtin_amdgcn_update_dpp(0, (int)r0, CTRL, 0xF, 0xF, false);
    const int a1 = __builtin_amdgcn_update_dpp(0, (int)r1, CTRL, 0xF, 0xF, false);
    const int a2 = __builtin_amdgcn_update_dpp(0, (int)r2, CTRL, 0xF, 0xF, false);
    const int a3 = __builtin_amdgcn_update_dpp(0, (int)r3, CTRL, 0xF, 0xF, false);
    return (unsigned)(j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3);
}

// ---- read pass.  U loads of 16 bytes per lane and iteration; PERSIST: grid-stride over wave chunks
template <int U, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void read_kernel(const float *__restrict__ w, long long nchunk, float *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    float sum = 0.f, mx = 0.f;
    for (long long c = wave0; c < nchunk; c += nw) {
        const f4 *src = reinterpret_cast<const f4 *>(w + c * (256 * U)) + lane;
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(src + 64 * u) : src[64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float a0 = fabsf(t[u].x), a1 = fabsf(t[u].y), a2 = fabsf(t[u].z), a3 = fabsf(t[u].w);
            sum += a0; sum += a1; sum += a2; sum += a3;
            mx = fmaxf(fmaxf(mx, fmaxf(a0, a1)), fmaxf(a2, a3));
        }
        if (!PERSIST) break;
    }
    const float s = wsum(sum), m = wmax(mx);
    if (lane == 0) { part[wave0 * 2] = s; part[wave0 * 2 + 1] = m; }
}

// ---- emit pass.  ST: 0 = 4-byte code stores per lane (8 per iteration at U = 8), 1 = quad-transposed 16-byte stores
template <int U, bool NT, bool PERSIST, int ST, bool REV>
__global__ __launch_bounds__(256) void emit_kernel(const float *__restrict__ w, long long nchunk, const float *__restrict__ stats, uint8_t *__restrict__ codes,
                                                   float *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    const float delta = stats[0];
    float sabs = 0.f, npos = 0.f, nneg = 0.f;
    for (long long cc = wave0; cc < nchunk; cc += nw) {
        const long long c = REV ? nchunk - 1 - cc : cc;
        const f4 *src = reinterpret_cast<const f4 *>(w + c * (256 * U)) + lane;
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(src + 64 * u) : src[64 * u];
        unsigned c4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = t[u][k];
                const unsigned cd = x > delta ? 1u : (x < -delta ? 2u : 0u);
                r |= cd << (8 * k);
                sabs += cd ? fabsf(x) : 0.f;
                npos += cd == 1u ? 1.f : 0.f;
                nneg += cd == 2u ? 1.f : 0.f;
            }
            c4[u] = r;
        }
        uint8_t *dst = codes + c * (256 * U);
        if (ST == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned *p = reinterpret_cast<unsigned *>(dst + 256 * u) + lane;
                if (NT) __builtin_nontemporal_store(c4[u], p); else *p = c4[u];
            }
        } else {
            // lanes 4q..4q+3 hold, for row u, the 16 codes at 256u + 16q .. +15.  Lane 4q+j takes row (4g + j) of group g.
            const int j = lane & 3, q = lane >> 2;
#pragma unroll
            for (int g = 0; g < U / 4; ++g) {
                u4 o;
                o[0] = quad_pick<0x00>(c4[4 * g], c4[4 * g + 1], c4[4 * g + 2], c4[4 * g + 3], j);
                o[1] = quad_pick<0x55>(c4[4 * g], c4[4 * g + 1], c4[4 * g + 2], c4[4 * g + 3], j);
                o[2] = quad_pick<0xAA>(c4[4 * g], c4[4 * g + 1], c4[4 * g + 2], c4[4 * g + 3], j);
                o[3] = quad_pick<0xFF>(c4[4 * g], c4[4 * g + 1], c4[4 * g + 2], c4[4 * g + 3], j);
                u4 *p = reinterpret_cast<u4 *>(dst + 256 * (4 * g + j) + 16 * q);
                if (NT) __builtin_nontemporal_store(o, p); else *p = o;
            }
        }
        if (!PERSIST) break;
    }
    const float s0 = wsum(sabs), s1 = wsum(npos), s2 = wsum(nneg);
    if (lane == 0) { part[wave0 * 3] = s0; part[wave0 * 3 + 1] = s1; part[wave0 * 3 + 2] = s2; }
}

// ---- gradient pass (TTQ): w, dwt read; dw written
template <int U, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void bwd_kernel(const float *__restrict__ w, const float *__restrict__ g, float *__restrict__ dw, long long nchunk,
                                                  const float *__restrict__ stats, float *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    const float delta = stats[0], pos = stats[1], neg = stats[2];
    float sp = 0.f, sn = 0.f;
    for (long long c = wave0; c < nchunk; c += nw) {
        const f4 *sg = reinterpret_cast<const f4 *>(g + c * (256 * U)) + lane;
        const f4 *sw = reinterpret_cast<const f4 *>(w + c * (256 * U)) + lane;
        f4 tg[U], tw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) tg[u] = NT ? __builtin_nontemporal_load(sg + 64 * u) : sg[64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) tw[u] = NT ? __builtin_nontemporal_load(sw + 64 * u) : sw[64 * u];
        f4 *d = reinterpret_cast<f4 *>(dw + c * (256 * U)) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dd = tg[u][k];
                const float x = tw[u][k];
                if (x > delta) { sp += dd; dd *= pos; }
                else if (x < -delta) { sn += dd; dd *= neg; }
                o[k] = dd;
            }
            if (NT) __builtin_nontemporal_store(o, d + 64 * u); else d[64 * u] = o;
        }
        if (!PERSIST) break;
    }
    const float s0 = wsum(sp), s1 = wsum(sn);
    if (lane == 0) { part[wave0 * 2] = s0; part[wave0 * 2 + 1] = s1; }
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
    const long long n = argc > 1 ? atoll(argv[1]) : 250000000LL;
    const long long npad = (n + 8191) / 8192 * 8192;  // multiple of every chunk size tried
    float *w, *g, *dw, *part, *stats;
    uint8_t *codes;
    CK(hipMalloc(&w, npad * 4)); CK(hipMalloc(&g, npad * 4)); CK(hipMalloc(&dw, npad * 4)); CK(hipMalloc(&codes, npad));
    CK(hipMalloc(&part, 64 << 20)); CK(hipMalloc(&stats, 64));
    {
        std::vector<float> h(1 << 22);
        unsigned s = 12345;
        for (auto &x : h) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (0.05f / (1 << 23)); }
        for (long long o = 0; o < npad; o += (long long)h.size()) {
            const long long c = std::min<long long>(h.size(), npad - o);
            CK(hipMemcpy(w + o, h.data(), c * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(g + o, h.data(), c * 4, hipMemcpyHostToDevice));
        }
        const float st[4] = {0.0175f, 0.1f, 0.08f, 0.f};
        CK(hipMemcpy(stats, st, 16, hipMemcpyHostToDevice));
    }
    printf("n = %lld\n", npad);
    const double GB = 1e-9;
#define RUN_READ(U, NT, P, GRID, label) { const long long nch = npad / (256 * U); const int grid = P ? GRID : (int)((nch + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((read_kernel<U, NT, P>), dim3(grid), dim3(256), 0, 0, w, nch, part); }); \
    printf("read  %-34s %.4f ms  %.0f GB/s\n", label, ms, npad * 4.0 * GB / (ms * 1e-3)); }
    RUN_READ(8, true, false, 0, "U8 nt oneshot (current)");
    RUN_READ(8, false, false, 0, "U8 plain oneshot");
    RUN_READ(4, true, false, 0, "U4 nt oneshot");
    RUN_READ(16, true, false, 0, "U16 nt oneshot");
    RUN_READ(8, true, true, 2048, "U8 nt persist 2048");
    RUN_READ(8, true, true, 4096, "U8 nt persist 4096");
    RUN_READ(4, true, true, 2048, "U4 nt persist 2048");
    RUN_READ(4, true, true, 4096, "U4 nt persist 4096");
    RUN_READ(16, true, true, 2048, "U16 nt persist 2048");
    RUN_READ(8, false, true, 2048, "U8 plain persist 2048");
#define RUN_EMIT(U, NT, P, ST, REV, GRID, label) { const long long nch = npad / (256 * U); const int grid = P ? GRID : (int)((nch + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((emit_kernel<U, NT, P, ST, REV>), dim3(grid), dim3(256), 0, 0, w, nch, stats, codes, part); }); \
    printf("emit  %-34s %.4f ms  %.0f GB/s\n", label, ms, npad * 5.0 * GB / (ms * 1e-3)); }
    RUN_EMIT(8, true, false, 0, false, 0, "U8 nt oneshot st4 (current~)");
    RUN_EMIT(8, false, false, 0, false, 0, "U8 plain oneshot st4");
    RUN_EMIT(8, true, false, 1, false, 0, "U8 nt oneshot st16");
    RUN_EMIT(4, true, false, 1, false, 0, "U4 nt oneshot st16");
    RUN_EMIT(16, true, false, 1, false, 0, "U16 nt oneshot st16");
    RUN_EMIT(8, true, true, 0, false, 2048, "U8 nt persist2048 st4");
    RUN_EMIT(8, true, true, 1, false, 2048, "U8 nt persist2048 st16");
    RUN_EMIT(8, true, true, 1, false, 4096, "U8 nt persist4096 st16");
    RUN_EMIT(4, true, true, 1, false, 4096, "U4 nt persist4096 st16");
    // both passes back to back: forward / reversed second pass, plain loads in the first (last-level cache reuse)
#define RUN_PAIR(NT1, REV, label) { const long long nch = npad / (256 * 8); const int grid = (int)((nch + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((read_kernel<8, NT1, false>), dim3(grid), dim3(256), 0, 0, w, nch, part); \
                            hipLaunchKernelGGL((emit_kernel<8, true, false, 1, REV>), dim3(grid), dim3(256), 0, 0, w, nch, stats, codes, part); }); \
    printf("pair  %-34s %.4f ms  %.0f GB/s\n", label, ms, npad * 9.0 * GB / (ms * 1e-3)); }
    RUN_PAIR(true, false, "nt read, forward emit");
    RUN_PAIR(true, true, "nt read, reversed emit");
    RUN_PAIR(false, false, "plain read, forward emit");
    RUN_PAIR(false, true, "plain read, reversed emit");
#define RUN_BWD(U, NT, P, GRID, label) { const long long nch = npad / (256 * U); const int grid = P ? GRID : (int)((nch + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((bwd_kernel<U, NT, P>), dim3(grid), dim3(256), 0, 0, w, g, dw, nch, stats, part); }); \
    printf("bwd   %-34s %.4f ms  %.0f GB/s\n", label, ms, npad * 12.0 * GB / (ms * 1e-3)); }
    RUN_BWD(8, true, false, 0, "U8 nt oneshot (current)");
    RUN_BWD(4, true, false, 0, "U4 nt oneshot");
    RUN_BWD(2, true, false, 0, "U2 nt oneshot");
    RUN_BWD(4, true, true, 2048, "U4 nt persist 2048");
    RUN_BWD(4, true, true, 4096, "U4 nt persist 4096");
    RUN_BWD(8, true, true, 2048, "U8 nt persist 2048");
    RUN_BWD(4, false, false, 0, "U4 plain oneshot");
    return 0;
}
