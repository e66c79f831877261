// membench.hip -- calibration microbenchmarks for the roofline numbers in DESIGN.md:
// plain float4 copy (the chip's achievable HBM rate) and the two access patterns the fused
// kernels add on top of it (chunked gather with a large power-of-two-ish stride).
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void k_copy2(const float2* __restrict__ in, float2* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void k_read4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i < n; i += stride) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.f) out[0] = acc;
}
__global__ void k_write4(float4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float4(1, 2, 3, 4);
}
// Workgroup g gathers `chunks` pieces of CH bytes, piece X at in + X*slab + g*CH (the pass-2 read
// pattern: one P x P chunk from every x-group), and writes them contiguously.
template <int CH>
__global__ void __launch_bounds__(1024) k_gather(const char* __restrict__ in, char* __restrict__ out, size_t slab, int chunks) {
    constexpr int LPC = CH / 16;     // lanes per chunk
    const int lane_in = threadIdx.x % LPC;
    const int c0 = threadIdx.x / LPC;
    const int cstep = blockDim.x / LPC;
    const size_t g = blockIdx.x;
    for (int X = c0; X < chunks; X += cstep) {
        const float4 v = *reinterpret_cast<const float4*>(in + (size_t)X * slab + g * CH + lane_in * 16);
        *reinterpret_cast<float4*>(out + (g * chunks + X) * (size_t)CH + lane_in * 16) = v;
    }
}

template <class F> float time_ms(F&& f, int iters = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB in, 1 GiB out: far beyond the 256 MiB Infinity Cache
    char *in, *out;
    CK(hipMalloc(&in, bytes + (1 << 20))); CK(hipMalloc(&out, bytes + (1 << 20)));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    for (int blocks : {2048, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, 0, (const float4*)in, (float4*)out, bytes / 16); });
        printf("{\"bench\":\"copy_float4\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy2, dim3(blocks), dim3(256), 0, 0, (const float2*)in, (float2*)out, bytes / 8); });
        printf("{\"bench\":\"copy_float2\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, (const float4*)in, (float4*)out, bytes / 16); });
        printf("{\"bench\":\"read_float4\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, 1.0 * bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(256), 0, 0, (float4*)out, bytes / 16); });
        printf("{\"bench\":\"write_float4\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, 1.0 * bytes / ms / 1e6);
    }
    // gather: field of N=4096 complex = 128 MiB; 3 fields -> use 384 MiB region; groups = N/P
    const int N = 4096;
    for (int pad : {0, 256, 4352}) {
        {   // 128-byte chunks: P=4, slab = N*P*8 (+pad), chunks = N/P, groups = N/P
            const size_t slab = (size_t)N * 4 * 8 + pad; const int chunks = N / 4, groups = N / 4;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<128>, dim3(groups), dim3(1024), 0, 0, in, out, slab, chunks); });
            printf("{\"bench\":\"gather128\",\"slab_pad\":%d,\"GBps\":%.1f}\n", pad, 2.0 * groups * chunks * 128 / ms / 1e6);
        }
        {   // 64-byte chunks
            const size_t slab = (size_t)N * 4 * 8 + pad; const int chunks = N / 4, groups = N / 2;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<64>, dim3(groups), dim3(1024), 0, 0, in, out, slab, chunks); });
            printf("{\"bench\":\"gather64\",\"slab_pad\":%d,\"GBps\":%.1f}\n", pad, 2.0 * groups * chunks * 64 / ms / 1e6);
        }
        {   // 256-byte chunks
            const size_t slab = (size_t)N * 8 * 8 + pad; const int chunks = N / 8, groups = N / 4;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<256>, dim3(groups), dim3(1024), 0, 0, in, out, slab, chunks); });
            printf("{\"bench\":\"gather256\",\"slab_pad\":%d,\"GBps\":%.1f}\n", pad, 2.0 * groups * chunks * 256 / ms / 1e6);
        }
    }
    return 0;
}
