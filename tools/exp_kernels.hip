// exp_kernels.hip -- ablation harness for the fused kernels (not product code): the same access
// patterns with pieces of the work removed, to see what bounds each kernel at N = 4096.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ocean_kernels.hpp"
using namespace ocean;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st4(c32* p, float4 v, bool nt) { if (nt) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); } else *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ c32 ld2(const c32* p, bool nt) { if (nt) { v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)); return mk(t.x, t.y); } return *p; }
__device__ __forceinline__ float ld1(const float* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
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
        if (MODE & 2) { const c32 a = ld2(own + y, MODE & 32), m = ld2(mir + (N - 1 - y), MODE & 32); const float w = ld1(om + y, MODE & 32); hs[e] = mk(a.x + m.x * w, a.y + m.y); }
        else hs[e] = propagate_height(ld2(own + y, MODE & 32), ld2(mir + (N - 1 - y), MODE & 32), ld1(om + y, MODE & 32), time);
    }
    c32* out_group = inter + (size_t)X * sx;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        c32 reg[E];
        const int jf = opaque_lane(j);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (f == 1) reg[e] = hs[e];
            else if (MODE & 2) reg[e] = mk(hs[e].y * (float)f, -hs[e].x);
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
            st4(dst + (size_t)(y / P) * sy + (size_t)(y % P) * P + 2 * h, make_float4(v0.x, v0.y, v1.x, v1.y), MODE & 16);
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
            lds_r[lds_pad(x0)] = mk(v[q].x, v[q].y);
            lds_r[lds_pad(x0 + 1)] = mk(v[q].z, v[q].w);
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


template <int N, int E, int SKIP>
__device__ __forceinline__ void x_load_AB(const c32* __restrict__ h0T, const float* __restrict__ omegaT,
                                          uint32_t x, int j, float time, c32 (&A)[E], c32 (&B)[E]) {
    constexpr int T = N / E;
    const uint32_t x2 = (N - x) & (N - 1);
    const uint32_t xm = (x - 1u) & (N - 1);
    const c32* own = h0T + (size_t)x * N;
    const c32* mir = h0T + (size_t)(N - 1 - x) * N;
    const c32* own2 = h0T + (size_t)x2 * N;
    const c32* mir2 = h0T + (size_t)xm * N;
    const float* om = omegaT + (size_t)x * N;
    const float* om2 = omegaT + (size_t)x2 * N;
    constexpr int PER = E / 4;
    int jj = j;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e > 0 && (e % PER) == 0) jj = opaque_after(j, A[e - 1].x + B[e - 1].y);
        const c32 a = (own + e * T)[jj];
        const c32 m = (mir + (N - (e + 1) * T))[T - 1 - jj];
        const float w = (om + e * T)[jj];
        c32 a2, m2; float w2;
        if (SKIP & 1) { a2 = m; m2 = a; }
        else if (e == 0) { a2 = own2[(N - jj) & (N - 1)]; m2 = mir2[(jj - 1) & (N - 1)]; }
        else { a2 = (own2 + (N - (e + 1) * T))[T - jj]; m2 = (mir2 + (e * T - 1))[jj]; }
        if (SKIP & 2) w2 = w;
        else if (e == 0) w2 = om2[(N - jj) & (N - 1)];
        else w2 = (om2 + (N - (e + 1) * T))[T - jj];
        A[e] = propagate_height(a, m, w, time);
        const c32 h2 = propagate_height(a2, m2, w2, time);
        B[e] = mk(h2.x, -h2.y);
    }
}

// half-spectrum pass 1 ablations: MODE bit0: no Nyquist block (grid = groups); bit1: no FFT; bit2: B = conj(A) (no second propagate/loads)
template <int N, int E, int P, int MODE>
__global__ void __launch_bounds__((N / E) * P)
x_half_pass1(const c32* __restrict__ h0T, const float* __restrict__ omegaT, c32* __restrict__ inter,
             float* __restrict__ nyq, const c32* __restrict__ tw, InterLayout lay, float time, float domain_size,
             unsigned long long* __restrict__ stamps = nullptr) {
#define STAMP(k) do { if ((MODE & 256) && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
    STAMP(0);
    constexpr int T = N / E;
    constexpr int H2 = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int c = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    c32* lds_line = lds + c * LinePitch<N>::elems;
    const float kscale = OCEAN_PI_F / domain_size;
    int bid = blockIdx.x, nb = gridDim.x;
    if (!(MODE & 1)) {
        if (blockIdx.x == 0) {
            const int f = (c < 3) ? c : 2;
            c32 A[E], B[E], reg[E];
            half_load_AB<N, E, false>(h0T, 1.0f, omegaT, (uint32_t)(N / 2), j, time, A, B);
            const c32 kxn = xx(mk(wave_index_q1((uint32_t)(N / 2), N) * kscale, 0.0f));
            half_spectrum<N, E>(f, A, B, kxn, kscale, j, reg);
            fft_line<N, E>(reg, j, tw, lds_line);
            if (c < 3) {
#pragma unroll
                for (int e = 0; e < E; ++e) nyq[(size_t)f * N + j + e * T] = reg[e].x;
            }
            return;
        }
        bid -= 1; nb -= 1;
    }
    if (MODE & 8) {   // phase staggering: every other first-round workgroup starts ~SLEEPS*3.9 us late
        if (bid < 256 && (bid & 8)) { for (int s = 0; s < (MODE >> 4); ++s) __builtin_amdgcn_s_sleep(127); }
    }
    const int X = xcd_contiguous(bid, nb);
    const uint32_t x = (uint32_t)(X * P + c);
    const uint32_t x2 = (N - x) & (N - 1);
    c32 A[E], B[E];
    if (MODE & 4) {
        const c32* own = h0T + (size_t)x * N; const c32* mir = h0T + (size_t)(N - 1 - x) * N; const float* om = omegaT + (size_t)x * N;
#pragma unroll
        for (int e = 0; e < E; ++e) { const int y = j + e * T; A[e] = propagate_height(own[y], mir[N - 1 - y], om[y], time); B[e] = mk(A[e].x, -A[e].y); }
    } else {
        if (MODE & 512) x_load_AB<N, E, 1>(h0T, omegaT, x, j, time, A, B);
        else if (MODE & 1024) x_load_AB<N, E, 3>(h0T, omegaT, x, j, time, A, B);
        else half_load_AB<N, E, false>(h0T, 1.0f, omegaT, x, j, time, A, B);
    }
    const c32 kxv = mk(wave_index_q1(x, N), wave_index_q1(x2, N)) * kscale;
    const int h = tid % H2;
    const int i = tid / H2;
    const c32* l0 = lds + (2 * h) * LinePitch<N>::elems;
    const c32* l1 = lds + (2 * h + 1) * LinePitch<N>::elems;
    if (MODE & 256) { asm volatile("" :: "v"(A[E - 1].x), "v"(B[E - 1].y)); __syncthreads(); STAMP(1); }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        c32 reg[E];
        const int jf = opaque_lane(j);
        half_spectrum<N, E>(f, A, B, kxv, kscale, jf, reg);
        if (MODE & 256) { asm volatile("" :: "v"(reg[E - 1].x)); STAMP(2 + 3 * f); }
        if (f > 0) __syncthreads();
        if (MODE & 2) {
            c32* g = lds_line + lds_pad(jf);
#pragma unroll
            for (int e = 0; e < E; ++e) g[e * (T + T / 16)] = reg[e];
            __syncthreads();
        } else {
            fft_line_to_lds<N, E>(reg, jf, tw, lds_line);
        }
        STAMP(3 + 3 * f);
        c32* dst = inter + (size_t)f * lay.fs + (size_t)X * lay.sx + (size_t)(i / P) * lay.sy + (i % P) * P + 2 * h;
#pragma unroll
        for (int q = 0; q < E / 2; ++q) {
            const int y = i + q * (2 * T);
            const c32 v0 = l0[lds_pad(y)];
            const c32 v1 = l1[lds_pad(y)];
            *reinterpret_cast<float4*>(dst + (size_t)q * ((2 * T) / P) * lay.sy) = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
        STAMP(4 + 3 * f);
    }
    STAMP(11);
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
    for (int i = 0; i < N; ++i) t[i] = mk((float)cos(2 * M_PI * i / N), (float)sin(2 * M_PI * i / N));
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
    {
        float* nyq; CK(hipMalloc(&nyq, 3 * N * 4));
        const size_t gx = (N / G::P) / 2;
        InterLayout lh{16, gx * 16 + 32, (gx * 16 + 32) * (size_t)(N / G::P)};
#define PH(MODE) { auto k = x_half_pass1<N, G::E, G::P, MODE>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::frame_lds)); \
        const int grid = ((MODE) & 1) ? (N / 2) / G::P : 1 + (N / 2) / G::P; \
        float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(G::frame_threads), G::frame_lds, 0, h0T, omT, inter, nyq, tw, lh, 1.5f, 1000.0f, (unsigned long long*)nullptr); }); \
        printf("{\"kernel\":\"half_pass1\",\"mode\":%d,\"ms\":%.4f}\n", MODE, ms); }
        PH(1) PH(1 + 512) PH(1 + 1024) PH(1) PH(1 + 512) PH(1 + 1024)
        {
            const size_t sx = 16, sy = (size_t)(N / G::P) * 16 + 32;
#define P1H(MODE, GRID) { auto k = x_pass1<N, G::E, G::P, MODE>; \
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::frame_lds)); \
            float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(GRID), dim3(G::frame_threads), G::frame_lds, 0, h0T, omT, inter, tw, sx, sy, fs, 1.5f, 1000.0f); }); \
            printf("{\"kernel\":\"c2c_pass1\",\"grid\":%d,\"mode\":%d,\"ms\":%.4f}\n", GRID, MODE, ms); }
            P1H(0, 1024)
        }
        if (0) {   // timeline of one launch
            unsigned long long* st; const int grid = (N / 2) / G::P;
            CK(hipMalloc(&st, (size_t)grid * 16 * 8)); CK(hipMemset(st, 0, (size_t)grid * 16 * 8));
            auto k = x_half_pass1<N, G::E, G::P, 257>;
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::frame_lds));
            float msi = time_ms([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(G::frame_threads), G::frame_lds, 0, h0T, omT, inter, nyq, tw, lh, 1.5f, 1000.0f, st); });
            printf("{\"kernel\":\"half_pass1 instrumented\",\"ms\":%.4f}\n", msi);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs((size_t)grid * 16);
            CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull; for (int b = 0; b < grid; ++b) if (hs[(size_t)b * 16] < t0) t0 = hs[(size_t)b * 16];
            const char* names[12] = {"start", "AB_ready", "spec0", "fft0", "store0", "spec1", "fft1", "store1", "spec2", "fft2", "store2", "end"};
            for (int b : {0, 1, 7, 100, 255, 256, 257, 300, 511}) {
                printf("{\"wg\":%d", b);
                for (int k2 = 0; k2 < 12; ++k2) printf(",\"%s\":%.1f", names[k2], (double)(hs[(size_t)b * 16 + k2] - t0) / 100.0);
                printf("}\n");
            }
            double acc[12] = {0};
            for (int b = 0; b < grid; ++b) for (int k2 = 1; k2 < 12; ++k2) acc[k2] += (double)(hs[(size_t)b * 16 + k2] - hs[(size_t)b * 16 + k2 - 1]) / 100.0;
            printf("{\"avg_phase_us\":{");
            for (int k2 = 1; k2 < 12; ++k2) printf("\"%s\":%.2f%s", names[k2], acc[k2] / grid, k2 < 11 ? "," : "");
            printf("}}\n");
        }
        auto k2 = k_half_pass2<N, G::E, G::P, G::R2>;
        CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, G::thin_lds));
        float ms = time_ms([&] { hipLaunchKernelGGL(k2, dim3(G::thin_grid), dim3(G::thin_threads), G::thin_lds, 0, inter, out, tw, lh); });
        printf("{\"kernel\":\"half_pass2\",\"ms\":%.4f}\n", ms);
    }
    return 0;
}
