// Host stand-in for gfx_ocean_amd/csrc/ocean_device_intrinsics.hpp (CPU emulation build only).
#pragma once
#include <cmath>
#include <cstdint>
namespace ocean {
// scalar twin of the device's packed 2-vector complex type and its primitives
struct c32 { float x, y; };
static inline c32 mk(float re, float im) { return c32{re, im}; }
static inline c32 operator+(c32 a, c32 b) { return c32{a.x + b.x, a.y + b.y}; }
static inline c32 operator-(c32 a, c32 b) { return c32{a.x - b.x, a.y - b.y}; }
static inline c32 operator*(c32 a, c32 b) { return c32{a.x * b.x, a.y * b.y}; }
static inline c32 operator*(c32 a, float s) { return c32{a.x * s, a.y * s}; }
static inline c32 operator-(c32 a) { return c32{-a.x, -a.y}; }
static inline c32 xx(c32 a) { return c32{a.x, a.x}; }
static inline c32 yy(c32 a) { return c32{a.y, a.y}; }
static inline c32 yx(c32 a) { return c32{a.y, a.x}; }
static inline c32 vfma(c32 a, c32 b, c32 c) { return c32{std::fmaf(a.x, b.x, c.x), std::fmaf(a.y, b.y, c.y)}; }
struct c32_pair { c32 a, b; };
struct u32_pair { uint32_t a, b; };
struct f32_pair { float a, b; };
static inline int opaque_lane(int x) { return x; }
static inline float ocean_emu_half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int exp = (h >> 10) & 31;
    const uint32_t man = h & 1023u;
    float mag;
    if (exp == 0) mag = std::ldexp((float)man, -24);
    else if (exp == 31) mag = man ? NAN : INFINITY;
    else mag = std::ldexp((float)(man | 1024u), exp - 25);
    return sign ? -mag : mag;
}
static inline c32 unpack_half2(uint32_t bits, float descale) {
    return mk(ocean_emu_half_to_float((uint16_t)(bits & 0xFFFFu)) * descale,
                       ocean_emu_half_to_float((uint16_t)(bits >> 16)) * descale);
}
static inline int wave_uniform(int x) { return x; }
static inline void wave_priority(int) {}
static inline float sin_rev(float x) { return (float)std::sin(6.283185307179586 * (double)x); }
static inline float cos_rev(float x) { return (float)std::cos(6.283185307179586 * (double)x); }
static inline void store_float4_nt(float4* p, float4 v) { *p = v; }
static inline float load_float_nt(const float* p) { return *p; }
static inline int opaque_after(int x, float) { return x; }
template <int T> static inline void line_sync() { __syncthreads(); }   // host threads are not a wave: always the full barrier
static inline void workgroup_publish() { __syncthreads(); }       // the emulation's barrier is a full fence
}  // namespace ocean
#define OCEAN_TL(k)
