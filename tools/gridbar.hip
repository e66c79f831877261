// gridbar.hip -- what would a single-launch frame at N = 512 pay for the device-scope barrier between its two passes?
// (measurement tool, not product code; DESIGN.md 4.6 "N = 512 in one launch")
// 256 co-resident workgroups (one per CU, 256 threads) meet at K device-scope barriers: each workgroup writes 12 KiB of an
// "intermediate" with write-through stores, arrives on an atomic counter in fine-grained (system-scope) fashion, spins
// until all have arrived, then reads 12 KiB that OTHER workgroups (other XCDs) wrote, bypassing its XCD's L2.
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o tools/gridbar && tools/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_gridbar(unsigned* counter, float* inter, float* sink, int rounds, int traffic) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        if (traffic) {   // 12 KiB per workgroup, visible device-wide: system-scope stores (write through the XCD L2)
#pragma unroll
            for (int k = 0; k < 12; ++k) __hip_atomic_store(inter + ((size_t)wg * 12 + k) * 256 + tid, (float)(r + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * (unsigned)nwg;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (traffic) {   // another XCD's workgroup's data (block wg + 1 runs on the next XCD), L2-bypassing loads
            const int src = (wg + 1) % nwg;
#pragma unroll
            for (int k = 0; k < 12; ++k) acc += __hip_atomic_load(inter + ((size_t)src * 12 + k) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (acc == 12345.0f) sink[0] = acc;
}
__global__ void k_empty() {}

int main() {
    unsigned* counter; float *inter, *sink;
    hipMalloc(&counter, 4); hipMalloc(&inter, 256 * 12 * 256 * 4); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int traffic = 0; traffic < 2; ++traffic)
        for (int rounds : {1, 101}) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipMemset(counter, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(a); hipLaunchKernelGGL(k_gridbar, dim3(256), dim3(256), 0, 0, counter, inter, sink, rounds, traffic); hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("traffic %d rounds %3d: %.2f us per launch\n", traffic, rounds, best * 1000);
        }
    {   // two dependent empty launches back to back: the boundary a fused launch would remove
        const int K = 2000;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("dependent empty launches: %.2f us each\n", ms * 1000 / K);
    }
    return 0;
}
