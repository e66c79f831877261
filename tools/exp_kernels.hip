// exp_kernels.hip -- ablation harness for the fused kernels (not product code): the same access
// patterns with pieces of the work removed, to see what bounds each kernel at N = 4096.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ocean_kernels.hpp"
using namespace ocean;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// MODE: 0 full, 1 no FFT (memory pattern only), 2 prefetch next field's loads before the FFT
template <int N, int E, int P1, int R2, int MODE>
__global__ void __launch_bounds__((N / E) * R2)
x_pass2_thin(const c32* __restrict__ inter, float4* __restrict__ out, const c32* __restrict__ tw, size_t sx, size_t sy, size_t field_stride, int xmap) {
    constexpr int T = N / E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    constexpr int S = (P1 > R2) ? (P1 / R2) : 1;
    int rb = blockIdx.x;
    if (xmap == 1 && S > 1 && (gridDim.x % (8 * S)) == 0) { const int xcd = rb & 7, slot = rb >> 3; rb = ((slot / S) * 8 + xcd) * S + (slot % S); }
    if (xmap == 2 && (gridDim.x % 8) == 0) rb = (rb & 7) * (gridDim.x >> 3) + (rb >> 3);
    const int y = rb * R2 + ll;
    c32* lds_line = lds + ll * LinePitch<N>::elems;
    float keep[2][E];
    c32 nxt[E];
    if (MODE == 2) {
        const c32* src = inter + (size_t)(j / P1) * sx + (size_t)(y / P1) * sy + (size_t)(y % P1) * P1 + (j % P1);
#pragma unroll
        for (int e = 0; e < E; ++e) nxt[e] = src[(size_t)e * (T / P1) * sx];
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int jf = opaque_lane(j);
        c32 reg[E];
        if (MODE == 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) reg[e] = nxt[e];
            if (f < 2) {
                const c32* src = inter + (size_t)(f + 1) * field_stride + (size_t)(jf / P1) * sx + (size_t)(y / P1) * sy + (size_t)(y % P1) * P1 + (jf % P1);
#pragma unroll
                for (int e = 0; e < E; ++e) nxt[e] = src[(size_t)e * (T / P1) * sx];
            }
        } else {
            const c32* src = inter + (size_t)f * field_stride + (size_t)(jf / P1) * sx + (size_t)(y / P1) * sy + (size_t)(y % P1) * P1 + (jf % P1);
#pragma unroll
            for (int e = 0; e < E; ++e) reg[e] = src[(size_t)e * (T / P1) * sx];
        }
        if (MODE != 1) {
            if (f > 0) __syncthreads();
            fft_line<N, E>(reg, jf, tw, lds_line);
        }
        if (f < 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) keep[f][e] = reg[e].x;
        } else {
            float4* orow = out + (size_t)y * N;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int xo = j + e * T;
                const float s = (((xo + y) & 1) == 0) ? -1.0f : 1.0f;
                orow[xo] = make_float4(keep[0][e] * s, keep[1][e] * s, reg[e].x * s, 0.0f);
            }
        }
    }
}

// pass 1 ablations: MODE 0 full, 1 no FFT (propagate + stores through LDS chunking), 2 no propagate math (h = own + mirror), 3 neither
template <int N, int E, int P, int MODE>
__global__ void __launch_bounds__((N / E) * P)
x_pass1(const c32* __restrict__ h0T, const float* __restrict__ omegaT, c32* __restrict__ inter,
        const c32* __restrict__ tw, size_t sx, size_t sy, size_t field_stride, float time, float domain_size) {
    constexpr int T = N / E;
    constexpr int H2 = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    const int X = pass1_group(blockIdx.x, gridDim.x);
    const uint32_t x = (uint32_t)(X * P + ll);
    c32* lds_line = lds + ll * LinePitch<N>::elems;
    const c32* own = h0T + (size_t)x * N;
    const c32* mir = h0T + (size_t)(N - 1 - x) * N;
    const float* om = omegaT + (size_t)x * N;
    const float kscale = OCEAN_PI_F / domain_size;
    const float kx = wave_index_q1(x, N) * kscale;
    c32 hs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int y = j + e * T;
        if (MODE & 2) { const c32 a = own[y], m = mir[N - 1 - y]; const float w = om[y]; hs[e] = make_float2(a.x + m.x * w, a.y + m.y); }
        else hs[e] = propagate_height(own[y], mir[N - 1 - y], om[y], time);
    }
    c32* out_group = inter + (size_t)X * sx;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        c32 reg[E];
        const int jf = opaque_lane(j);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (f == 1) reg[e] = hs[e];
            else if (MODE & 2) reg[e] = make_float2(hs[e].y * (float)f, -hs[e].x);
            else {
                const float ky = wave_index_q1((uint32_t)(jf + e * T), N) * kscale;
                float knx, kny;
                k_normalised_fast(kx, ky, knx, kny);
                reg[e] = mul_minus_i_kn((f == 0) ? knx : kny, hs[e]);
            }
        }
        if (f > 0) __syncthreads();
        if (MODE & 1) {
            c32* g = lds_line + lds_pad(jf);
#pragma unroll
            for (int e = 0; e < E; ++e) g[e * (T + T / 16)] = reg[e];
            __syncthreads();
        } else {
            fft_line_to_lds<N, E>(reg, jf, tw, lds_line);
        }
        c32* dst = out_group + (size_t)f * field_stride;
        const int h = tid % H2;
        const int i = tid / H2;
        const c32* l0 = lds + (2 * h) * LinePitch<N>::elems;
        const c32* l1 = lds + (2 * h + 1) * LinePitch<N>::elems;
#pragma unroll
        for (int q = 0; q < E / 2; ++q) {
            const int y = i + q * (2 * T);
            const c32 v0 = l0[lds_pad(y)];
            const c32 v1 = l1[lds_pad(y)];
            *reinterpret_cast<float4*>(dst + (size_t)(y / P) * sy + (size_t)(y % P) * P + 2 * h) = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
    }
}

// fat pass 2 (1024 threads, P rows): MODE 0 full, 1 no FFT
template <int N, int E, int P, int MODE>
__global__ void __launch_bounds__((N / E) * P)
x_pass2_fat(const c32* __restrict__ inter, float4* __restrict__ out, const c32* __restrict__ tw, size_t sx, size_t sy, size_t field_stride) {
    constexpr int T = N / E;
    constexpr int H2 = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    const int Y = blockIdx.x;
    const int y = Y * P + ll;
    c32* lds_line = lds + ll * LinePitch<N>::elems;
    const int h = tid % H2;
    const int r = (tid / H2) % P;
    const int xi = tid / (H2 * P);
    constexpr int XSTEP = (2 * T) / P;
    c32* lds_r = lds + r * LinePitch<N>::elems;
    float keep[2][E];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const c32* src = inter + (size_t)f * field_stride + (size_t)Y * sy + (size_t)r * P + 2 * h;
        float4 v[E / 2];
#pragma unroll
        for (int q = 0; q < E / 2; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(xi + q * XSTEP) * sx);
        if (f > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < E / 2; ++q) {
            const int x0 = (xi + q * XSTEP) * P + 2 * h;
            lds_r[lds_pad(x0)] = make_float2(v[q].x, v[q].y);
            lds_r[lds_pad(x0 + 1)] = make_float2(v[q].z, v[q].w);
        }
        __syncthreads();
        c32 reg[E];
        const int jf = opaque_lane(j);
        const c32* g = lds_line + lds_pad(jf);
#pragma unroll
        for (int e = 0; e < E; ++e) reg[e] = g[e * (T + T / 16)];
        if (MODE != 1) { __syncthreads(); fft_line<N, E>(reg, jf, tw, lds_line); }
        if (f < 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) keep[f][e] = reg[e].x;
        } else {
            float4* orow = out + (size_t)y * N;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int xo = j + e * T;
                const float s = (((xo + y) & 1) == 0) ? -1.0f : 1.0f;
                orow[xo] = make_float4(keep[0][e] * s, keep[1][e] * s, reg[e].x * s, 0.0f);
            }
        }
    }
}

template <class F> float time_ms(F&& f, int iters = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    constexpr int N = 4096;
    using G = Geo<N>;
    const size_t n2 = (size_t)N * N;
    const size_t slab = (size_t)N * G::P + 32, fs = slab * (N / G::P);
    c32 *h0T, *inter, *tw; float *omT; float4* out;
    CK(hipMalloc(&h0T, n2 * 8)); CK(hipMalloc(&omT, n2 * 4)); CK(hipMalloc(&inter, 3 * fs * 8)); CK(hipMalloc(&out, n2 * 16)); CK(hipMalloc(&tw, N * 8));
    std::vector<float> r(n2 * 2);
    for (size_t i = 0; i < r.size(); ++i) r[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    CK(hipMemcpy(h0T, r.data(), n2 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(omT, r.data(), n2 * 4, hipMemcpyHostToDevice));
    for (int f = 0; f < 3; ++f) CK(hipMemcpy(inter + f * fs, r.data(), n2 * 8, hipMemcpyHostToDevice));
    std::vector<c32> t(N);
    for (int i = 0; i < N; ++i) t[i] = make_float2((float)cos(2 * M_PI * i / N), (float)sin(2 * M_PI * i / N));
    CK(hipMemcpy(tw, t.data(), N * 8, hipMemcpyHostToDevice));
    for (int layout = 0; layout < 2; ++layout) {
        // layout 0: pass-1-contiguous (chunk(X,Y) = X*slab + Y*16); layout 1: pass-2-contiguous (Y*slabY + X*16)
        const size_t sx = layout == 0 ? slab : 16, sy = layout == 0 ? 16 : (size_t)(N / G::P) * 16 + 32;
#define P2(MODE, XMAP) { auto k = x_pass2_thin<N, G::E, G::P, G::R2, MODE>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::thin_lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(G::thin_grid), dim3(G::thin_threads), G::thin_lds, 0, inter, out, tw, sx, sy, fs, XMAP); }); \
        printf("{\"kernel\":\"pass2_thin\",\"layout\":%d,\"mode\":%d,\"xmap\":%d,\"ms\":%.4f,\"GBps_alg\":%.0f}\n", layout, MODE, XMAP, ms, 40.0 * n2 / ms / 1e6); }
        P2(0, 1) P2(1, 1) P2(0, 0) P2(1, 0) P2(0, 2) P2(1, 2)
#define P2F(MODE) { auto k = x_pass2_fat<N, G::E, G::P, MODE>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::frame_lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(G::frame_grid), dim3(G::frame_threads), G::frame_lds, 0, inter, out, tw, sx, sy, fs); }); \
        printf("{\"kernel\":\"pass2_fat\",\"layout\":%d,\"mode\":%d,\"ms\":%.4f,\"GBps_alg\":%.0f}\n", layout, MODE, ms, 40.0 * n2 / ms / 1e6); }
        P2F(0) P2F(1)
#define P1(MODE) { auto k = x_pass1<N, G::E, G::P, MODE>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::frame_lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(G::frame_grid), dim3(G::frame_threads), G::frame_lds, 0, h0T, omT, inter, tw, sx, sy, fs, 1.5f, 1000.0f); }); \
        printf("{\"kernel\":\"pass1\",\"layout\":%d,\"mode\":%d,\"ms\":%.4f,\"GBps_alg\":%.0f}\n", layout, MODE, ms, 36.0 * n2 / ms / 1e6); }
        P1(0) P1(1)
    }
    return 0;
}
