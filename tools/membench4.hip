// membench4.hip -- does the 256 MiB Infinity Cache help a re-read working set, and do streaming writes evict it?
// (1) read bandwidth of a buffer re-read every launch, by footprint; (2) the same 192 MiB buffer re-read with a
// 512 MiB streaming write (plain / non-temporal stores) between the reads -- the shape of a frame loop whose
// static inputs (201 MB at N = 4096) would like to stay cached while 670 MB of intermediates stream by.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o tools/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_read4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i < n; i += stride) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.f) out[0] = acc;
}
template <bool NT> __global__ void k_write4(float4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (NT) { v4f t = {1, 2, 3, 4}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(out + i)); }
        else out[i] = make_float4(1, 2, 3, 4);
    }
}

int main() {
    const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
    float4 *in, *out;
    CK(hipMalloc(&in, 2 * GiB)); CK(hipMalloc(&out, GiB));
    CK(hipMemset(in, 1, 2 * GiB)); CK(hipMemset(out, 0, GiB));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 4096;
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n = mb * MiB / 16;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, in, out, n);
        const int reps = 20;
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, in, out, n);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("{\"bench\":\"reread\",\"footprint_MiB\":%zu,\"GBps\":%.0f}\n", mb, (double)mb * MiB * reps / ms / 1e6);
    }
    for (int nt = 0; nt < 2; ++nt) {
        const size_t n = 192 * MiB / 16, wn = 512 * MiB / 16;
        float ms_r = 0;
        for (int r = 0; r < 12; ++r) {
            if (nt) hipLaunchKernelGGL(k_write4<true>, dim3(blocks), dim3(256), 0, 0, out, wn);
            else hipLaunchKernelGGL(k_write4<false>, dim3(blocks), dim3(256), 0, 0, out, wn);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, in, out + wn, n);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 2) ms_r += ms;
        }
        printf("{\"bench\":\"reread_192MiB_after_512MiB_%s_write\",\"GBps\":%.0f}\n", nt ? "nontemporal" : "plain", 192.0 * MiB * 10 / ms_r / 1e6);
    }
    return 0;
}
