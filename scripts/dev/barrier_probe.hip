// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip): what a grid-wide hand-off costs
// inside one launch against a kernel boundary, at the geometry of the ResNet-20 convolution kernels (512 workgroups of
// 256 threads, two per CU).  A "phase" reads 16 KB that ANOTHER workgroup wrote in the previous phase, adds one, and
// writes its own 16 KB (so every phase depends on all-to-all completion of the previous one, like a convolution whose
// BatchNorm statistics span the batch).
//   chain   : P phases as P kernel launches in one captured graph (the boundary is the hand-off)
//   fused   : P phases inside ONE launch, separated by a grid barrier (per-XCD counters -> one top counter ->
//             per-XCD generation word; release fence before arriving, acquire fence after leaving)
//   fused, leader release : the same barrier, but only the LAST arriver of each XCD runs the release fence (one write-back of
//             that XCD's L2 per phase instead of one per workgroup); every workgroup still runs the acquire
// Every spin is bounded: a workgroup that waits longer than ~2 s of wall clock raises an error word and leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WG = 512, T = 256, WORDS = 4096;  // 16 KB per workgroup and phase

struct Bar {
    unsigned xcd_count[8 * 32];  // one counter per XCD, each on a 128-byte line of its own
    unsigned top_count[32];
    unsigned gen[8 * 32];        // generation word per XCD
    unsigned error;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false on time-out
template <bool LEADER>
__device__ bool grid_barrier(Bar *b, unsigned phase, unsigned per_xcd, unsigned nxcd)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wavefront: its stores have left before the barrier
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7;  // blocks are dealt round-robin over the XCDs: b and b + 8 share one (speed only)
        if (!LEADER) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned arrived = __hip_atomic_fetch_add(&b->xcd_count[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const long long t0 = wall_clock64();
        if (arrived == per_xcd * (phase + 1)) {  // last of this XCD: up to the top counter
            if (LEADER) {  // every workgroup of this XCD has its stores in the XCD's L2: one write-back for all of them
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned top = __hip_atomic_fetch_add(&b->top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (top == nxcd * (phase + 1)) {
                for (unsigned x = 0; x < 8; ++x) __hip_atomic_store(&b->gen[x * 32], phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (ld_sc1(&b->gen[xcd * 32]) < phase + 1) {
            __builtin_amdgcn_s_sleep(2);
            if (ld_sc1(&b->error) != 0u) {  // somebody gave up: everybody leaves
                ok = false;
                break;
            }
            if (wall_clock64() - t0 > 200000000LL) {  // 2 s at 100 MHz
                __hip_atomic_store(&b->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __shared__ int okw;
    if (threadIdx.x == 0) okw = ok ? 1 : 0;
    __syncthreads();
    return okw != 0;
}

__device__ __forceinline__ void phase_body(const unsigned *src, unsigned *dst, int wg)
{
    const int from = (wg * 37 + 11) % WG;  // another workgroup's slab (another XCD for most)
    for (int i = threadIdx.x; i < WORDS; i += T) dst[wg * WORDS + i] = src[from * WORDS + i] + 1u;
}

__global__ __launch_bounds__(256, 2) void phase_kernel(const unsigned *src, unsigned *dst) { phase_body(src, dst, blockIdx.x); }

template <bool LEADER>
__global__ __launch_bounds__(256, 2) void fused_kernel(unsigned *a, unsigned *bbuf, Bar *bar, int phases)
{
    unsigned *src = a, *dst = bbuf;
    for (int p = 0; p < phases; ++p) {
        phase_body(src, dst, blockIdx.x);
        if (p + 1 < phases) {
            if (!grid_barrier<LEADER>(bar, (unsigned)p, WG / 8, 8)) return;
        }
        unsigned *t = src; src = dst; dst = t;
    }
}

template <bool LEADER>
static bool run_fused(const char *name, unsigned *a, unsigned *b, Bar *bar, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int P, int R)
{
    std::vector<unsigned> h((size_t)WG * WORDS);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
        CHECK(hipMemsetAsync(a, 0, (size_t)WG * WORDS * 4, st));
        hipLaunchKernelGGL(fused_kernel<LEADER>, dim3(WG), dim3(T), 0, st, a, b, bar, P);
    }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), (P & 1) ? b : a, h.size() * 4, hipMemcpyDeviceToHost));
    size_t wrong = 0;
    for (size_t i = 0; i < h.size(); ++i) wrong += h[i] != (unsigned)P;
    unsigned err = 0;
    CHECK(hipMemcpy(&err, &bar->error, 4, hipMemcpyDeviceToHost));
    printf("%s : result words wrong %zu of %zu, time-out flag %u\n", name, wrong, h.size(), err);
    if (err) return false;
    float tot = 0;
    for (int r = 0; r < R; ++r) {
        CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(fused_kernel<LEADER>, dim3(WG), dim3(T), 0, st, a, b, bar, P);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
    }
    printf("%s : one launch of %d phases, %.2f us per phase (%.1f us per launch)\n", name, P, tot * 1e3 / R / P, tot * 1e3 / R);
    return true;
}

int main()
{
    const int P = 38;  // forward + input-gradient convolutions of a ResNet-20 pass
    unsigned *a, *b;
    Bar *bar;
    CHECK(hipMalloc(&a, (size_t)WG * WORDS * 4));
    CHECK(hipMalloc(&b, (size_t)WG * WORDS * 4));
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fused_kernel<false>, T, 0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, occupancy API %d blocks per CU -> %d resident, grid %d\n", prop.multiProcessorCount, occ, occ * prop.multiProcessorCount, WG);
    if (occ * prop.multiProcessorCount < WG) { printf("grid would not be resident: not running the fused form\n"); return 0; }

    // chain as a captured graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(WG), dim3(T), 0, st, (p & 1) ? b : a, (p & 1) ? a : b);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemsetAsync(a, 0, (size_t)WG * WORDS * 4, st));
        CHECK(hipGraphLaunch(ge, st));
    }
    CHECK(hipStreamSynchronize(st));
    const int R = 20;
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < R; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain : %d launches per graph, %.2f us per phase (graph of %d phases: %.1f us)\n", P, ms * 1e3 / R / P, P, ms * 1e3 / R);

    if (!run_fused<false>("fused", a, b, bar, st, e0, e1, P, R)) return 0;
    if (!run_fused<true>("fused, leader release", a, b, bar, st, e0, e1, P, R)) return 0;
    // the phase body alone (what both forms contain): one launch, no dependency
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(phase_kernel, dim3(WG), dim3(T), 0, st, a, b);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("body  : %.2f us per eager launch back to back\n", ms * 1e3 / R);
    return 0;
}
