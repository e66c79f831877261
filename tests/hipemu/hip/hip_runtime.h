// Host emulation of the small slice of the HIP device language that
// gfx_ocean_amd/csrc/{fft_core,ocean_kernels}.hpp use.  TEST INFRASTRUCTURE ONLY: it lets the
// CPU test-suite execute the *unmodified* kernel sources (index algebra, LDS exchanges, barriers)
// with one OS thread per GPU thread.  It is never part of the product build, which is hipcc/gfx950 only.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct emu_dim3 { unsigned x = 0, y = 0, z = 0; };   // (every launch sets what it uses)
extern thread_local emu_dim3 threadIdx;
extern thread_local emu_dim3 blockIdx;
extern emu_dim3 blockDim;
extern emu_dim3 gridDim;
void __syncthreads();

// ocml's sincospif (sin(pi x), cos(pi x)); host version evaluated in double.
static inline void sincospif(float x, float* s, float* c) {
    const double a = 3.14159265358979323846 * (double)x;
    *s = (float)std::sin(a);
    *c = (float)std::cos(a);
}

static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
// round-to-nearest fp32 add / multiply that the compiler must not contract into an FMA
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
