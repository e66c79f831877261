// membench5.hip -- what ONE CU can pull from HBM, as a function of how many CUs pull at the same time.
// Pass 1's load phase moves 448 KiB per workgroup (one per CU) in 17-18 us = 25 GB/s per CU, which is also 1/256 of what the
// part delivers when every CU loads at once (EXPERIMENTS "What the loader did NOT buy").  If a CU could pull much more while
// the others compute, a chip-level stagger of the load phases would shorten them.  This measures exactly that: G workgroups of
// 1024 threads (one per CU, 160 KiB of LDS requested so that no two share a CU) each stream SPAN bytes from their own region,
// through VGPRs (dwordx4, UNROLL loads in flight per lane) or through LDS-DMA (global_load_lds_dwordx4, a ring), REPS regions
// per workgroup so that nothing is cache-resident.       hipcc --offload-arch=gfx950 -O3 -o membench5 membench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int THREADS = 1024;
constexpr size_t SPAN = 448 * 1024;          // bytes per workgroup and repetition (pass 1 at N = 4096)

template <int UNROLL>
__global__ void __launch_bounds__(THREADS) k_vgpr(const float4* __restrict__ in, float* __restrict__ sink, size_t stride_f4, int reps) {
    extern __shared__ float4 pad[];
    float4 acc = make_float4(0, 0, 0, 0);
    constexpr int PER_REP = (int)(SPAN / 16 / THREADS);          // float4 per thread and repetition (28)
    for (int r = 0; r < reps; ++r) {
        const float4* p = in + ((size_t)r * gridDim.x + blockIdx.x) * stride_f4 + threadIdx.x;
        for (int i = 0; i < PER_REP; i += UNROLL) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = (i + u < PER_REP) ? p[(size_t)(i + u) * THREADS] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
    if (threadIdx.x == 0 && sink == nullptr) pad[0] = acc;
}

__device__ __forceinline__ void glds16(const void* base_uniform, uint32_t lane_offset, uint32_t lds_dst_uniform) {
    const uint64_t b = (uint64_t)base_uniform;
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_nop 1\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_offset), "s"(sb), "s"(dst) : "memory");
}
// every wave DMAs its share into its own 7 KiB window of LDS, INFLIGHT instructions outstanding
template <int INFLIGHT>
__global__ void __launch_bounds__(THREADS) k_dma(const float4* __restrict__ in, float* __restrict__ sink, size_t stride_f4, int reps) {
    extern __shared__ float4 lds[];
    constexpr int PER_REP = (int)(SPAN / 16 / THREADS);          // wave instructions per wave and repetition (28)
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)lds + (uint32_t)wave * (INFLIGHT * 1024);
    for (int r = 0; r < reps; ++r) {
        const char* base = (const char*)(in + ((size_t)r * gridDim.x + blockIdx.x) * stride_f4 + (size_t)wave * 64);
        for (int i = 0; i < PER_REP; ++i) {
            glds16(base + (size_t)i * THREADS * 16, (uint32_t)lane * 16u, lds0 + (uint32_t)(i % INFLIGHT) * 1024u);
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(INFLIGHT - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __syncthreads();
    if (sink == nullptr) sink[threadIdx.x] = lds[threadIdx.x].x;
}

template <class F> float time_us(F&& f, int iters = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / iters;
}

int main() {
    const int reps = 16;
    const int maxg = 256;
    const size_t stride_f4 = SPAN / 16;
    const size_t bytes = (size_t)reps * maxg * SPAN;              // 1.8 GB: beyond every cache
    float4* in; float* sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&sink, 4096 * 4));
    CK(hipMemset(in, 1, bytes));
    const size_t lds = 160 * 1024 - 512;
    CK(hipFuncSetAttribute((const void*)k_vgpr<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_vgpr<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_dma<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("# G workgroups (one per CU) x %d repetitions of %zu KiB each; per-CU GB/s and chip TB/s\n", reps, SPAN / 1024);
    for (int g : {8, 32, 64, 128, 192, 256}) {
        struct { const char* name; float us; } rows[4];
        rows[0] = {"vgpr x7 ", time_us([&] { hipLaunchKernelGGL(k_vgpr<7>, dim3(g), dim3(THREADS), lds, 0, in, sink, stride_f4, reps); })};
        rows[1] = {"vgpr x14", time_us([&] { hipLaunchKernelGGL(k_vgpr<14>, dim3(g), dim3(THREADS), lds, 0, in, sink, stride_f4, reps); })};
        rows[2] = {"dma  x4 ", time_us([&] { hipLaunchKernelGGL(k_dma<4>, dim3(g), dim3(THREADS), lds, 0, in, sink, stride_f4, reps); })};
        rows[3] = {"dma  x7 ", time_us([&] { hipLaunchKernelGGL(k_dma<7>, dim3(g), dim3(THREADS), lds, 0, in, sink, stride_f4, reps); })};
        for (auto& r : rows) {
            const double per_wg = (double)reps * SPAN;
            printf("G=%3d %s  %8.1f us   %6.1f GB/s per CU   %5.2f TB/s chip   (%5.1f us per 448 KiB)\n", g, r.name, r.us,
                   per_wg / r.us * 1e-3, per_wg * g / r.us * 1e-6, r.us / reps);
        }
    }
    CK(hipGetLastError());
    return 0;
}
