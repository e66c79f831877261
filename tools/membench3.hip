// membench3.hip -- scattered-chunk reads / writes beyond the 256 MiB Infinity Cache:
// WG g moves chunk g of each of K slabs (addr = X*slab + g*CH) to/from a contiguous span.
// This is the pass-1 -> pass-2 hand-off pattern of the fused frame; CH = P1*P2*8 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int CH, bool SCATTER> __global__ void __launch_bounds__(256)
k_chunks(const char* __restrict__ in, char* __restrict__ out, size_t slab, int K, int xcd_contig) {
    constexpr int LPC = CH / 16;                 // lanes per chunk
    int g = blockIdx.x;
    if (xcd_contig && (gridDim.x % 8) == 0) g = (g & 7) * (gridDim.x >> 3) + (g >> 3);
    const int lane = threadIdx.x % LPC, c0 = threadIdx.x / LPC, cstep = 256 / LPC;
    for (int X = c0; X < K; X += cstep) {
        const size_t chunked = (size_t)X * slab + (size_t)g * CH + lane * 16;
        const size_t linear = ((size_t)g * K + X) * CH + lane * 16;
        if (SCATTER) *reinterpret_cast<float4*>(out + chunked) = *reinterpret_cast<const float4*>(in + linear);
        else *reinterpret_cast<float4*>(out + linear) = *reinterpret_cast<const float4*>(in + chunked);
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
template <int CH> void run(const char* in, char* out, size_t total) {
    const int K = 1024;
    const int G = (int)(total / K / CH);
    for (int pad : {0, 256}) for (int xc : {0, 1}) {
        const size_t slab = (size_t)G * CH + pad;
        float ms = time_ms([&] { hipLaunchKernelGGL((k_chunks<CH, false>), dim3(G), dim3(256), 0, 0, in, out, slab, K, xc); });
        printf("{\"bench\":\"gather\",\"chunk\":%d,\"pad\":%d,\"xcd_contig\":%d,\"GBps\":%.1f}\n", CH, pad, xc, 2.0 * total / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_chunks<CH, true>), dim3(G), dim3(256), 0, 0, in, out, slab, K, xc); });
        printf("{\"bench\":\"scatter\",\"chunk\":%d,\"pad\":%d,\"xcd_contig\":%d,\"GBps\":%.1f}\n", CH, pad, xc, 2.0 * total / ms / 1e6);
    }
}
int main() {
    const size_t total = (size_t)1 << 30;
    char *in, *out;
    CK(hipMalloc(&in, total + (1 << 22))); CK(hipMalloc(&out, total + (1 << 22)));
    CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
    run<64>(in, out, total); run<128>(in, out, total); run<256>(in, out, total); run<512>(in, out, total);
    run<1024>(in, out, total); run<4096>(in, out, total);
    return 0;
}
