// membench2.hip -- what is the best achievable HBM write / mixed rate on this MI355X?
// Variants: store flavour (plain / nontemporal), work assignment (grid-stride vs contiguous block per WG),
// read:write mixes matching the fused kernels (pass1 ~ 0.73:1, pass2 1.5:1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(float4* p, float4 v) {
    if (NT) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); } else *p = v;
}
template <bool NT> __device__ __forceinline__ float4 ld(const float4* p) {
    if (NT) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
    else return *p;
}
// every WG owns a contiguous span of `span` float4 (like a row-owning FFT workgroup)
template <bool NT> __global__ void __launch_bounds__(256) k_write_block(float4* out, size_t span) {
    float4* p = out + blockIdx.x * span;
    for (size_t i = threadIdx.x; i < span; i += 256) st<NT>(p + i, make_float4(1, 2, 3, 4));
}
template <bool NT> __global__ void __launch_bounds__(256) k_write_stride(float4* out, size_t n) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) st<NT>(out + i, make_float4(1, 2, 3, 4));
}
// copy with R reads per W writes in 16-KiB units per thread-iteration (R:W = RN:WN)
template <bool NTL, bool NTS, int RN, int WN> __global__ void __launch_bounds__(256)
k_mix_block(const float4* in, float4* out, size_t span_iters) {
    // per iteration a WG reads RN*256 float4 and writes WN*256 float4, contiguous per WG
    const float4* pi = in + blockIdx.x * span_iters * RN * 256;
    float4* po = out + blockIdx.x * span_iters * WN * 256;
    for (size_t it = 0; it < span_iters; ++it) {
        float4 v[RN];
#pragma unroll
        for (int r = 0; r < RN; ++r) v[r] = ld<NTL>(pi + (it * RN + r) * 256 + threadIdx.x);
        float4 acc = v[0];
#pragma unroll
        for (int r = 1; r < RN; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
#pragma unroll
        for (int w = 0; w < WN; ++w) st<NTS>(po + (it * WN + w) * 256 + threadIdx.x, acc);
    }
}
template <class F> float time_ms(F&& f, int iters = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    char *in, *out;
    CK(hipMalloc(&in, 2 * bytes)); CK(hipMalloc(&out, 2 * bytes));
    CK(hipMemset(in, 1, 2 * bytes)); CK(hipMemset(out, 0, 2 * bytes));
    const size_t n4 = bytes / 16;
    for (int wgs : {1024, 4096, 16384, 65536}) {
        const size_t span = n4 / wgs;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_write_block<false>, dim3(wgs), dim3(256), 0, 0, (float4*)out, span); });
        printf("{\"bench\":\"write_block\",\"wgs\":%d,\"span_KiB\":%zu,\"GBps\":%.1f}\n", wgs, span * 16 / 1024, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_write_block<true>, dim3(wgs), dim3(256), 0, 0, (float4*)out, span); });
        printf("{\"bench\":\"write_block_nt\",\"wgs\":%d,\"span_KiB\":%zu,\"GBps\":%.1f}\n", wgs, span * 16 / 1024, bytes / ms / 1e6);
    }
    for (int blocks : {1024, 2048, 4096}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_write_stride<false>, dim3(blocks), dim3(256), 0, 0, (float4*)out, n4); });
        printf("{\"bench\":\"write_stride\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_write_stride<true>, dim3(blocks), dim3(256), 0, 0, (float4*)out, n4); });
        printf("{\"bench\":\"write_stride_nt\",\"blocks\":%d,\"GBps\":%.1f}\n", blocks, bytes / ms / 1e6);
    }
    // mixes: 1:1 (copy), 3:2 (pass2), 3:4 (pass1-like)
#define MIX(NTL, NTS, RN, WN, wgs) { \
        const size_t iters = n4 / ((size_t)(wgs) * 256 * ((RN) > (WN) ? (RN) : (WN))); \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mix_block<NTL, NTS, RN, WN>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, (float4*)out, iters); }); \
        const double moved = (double)(wgs) * iters * 256 * 16 * ((RN) + (WN)); \
        printf("{\"bench\":\"mix\",\"ntl\":%d,\"nts\":%d,\"R\":%d,\"W\":%d,\"wgs\":%d,\"GBps\":%.1f}\n", NTL, NTS, RN, WN, wgs, moved / ms / 1e6); }
    for (int wgs : {2048, 8192}) {
        MIX(false, false, 1, 1, wgs) MIX(false, true, 1, 1, wgs) MIX(true, true, 1, 1, wgs)
        MIX(false, false, 3, 2, wgs) MIX(false, true, 3, 2, wgs) MIX(true, true, 3, 2, wgs)
        MIX(false, false, 3, 4, wgs) MIX(false, true, 3, 4, wgs) MIX(true, true, 3, 4, wgs)
        MIX(false, false, 4, 1, wgs) MIX(false, false, 1, 4, wgs)
    }
    return 0;
}
