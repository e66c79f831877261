// Host stand-in for gfx_ocean_amd/csrc/ocean_device_intrinsics.hpp (CPU emulation build only).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
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
static inline unsigned hw_wave_slot() { return 0u; }
static inline void wave_sleep_127() {}
static inline float sin_rev(float x) { return (float)std::sin(6.283185307179586 * (double)x); }
static inline float cos_rev(float x) { return (float)std::cos(6.283185307179586 * (double)x); }
static inline void store_float4_nt(float4* p, float4 v) { *p = v; }
static inline void pin_here(c32&, c32&) {}
template <int T> static inline void line_sync() { __syncthreads(); }   // host threads are not a wave: always the full barrier
static inline void workgroup_publish() { __syncthreads(); }       // the emulation's barrier is a full fence
// LDS-DMA (half_load_AB_dma): the copy happens at issue, lane by lane; the waits are no-ops and the barrier is the
// emulation's full barrier, so what is checked here is the ring's index algebra and the barrier protocol's ordering.
extern unsigned char smem[];
static inline uint32_t lds_address(const void* p) { return (uint32_t)(reinterpret_cast<const unsigned char*>(p) - smem); }
template <bool NT> static inline void glds16(const void* base_uniform, uint32_t lane_offset, uint32_t lds_dst_uniform) {
    std::memcpy(smem + lds_dst_uniform + (threadIdx.x % 64) * 16, reinterpret_cast<const unsigned char*>(base_uniform) + lane_offset, 16);
}
template <int K> static inline void dma_wait() {}
static inline void dma_barrier() { __syncthreads(); }
// 16-bit block-floating intermediate: the wave reduction through a scratch array and two barriers (every thread of the
// workgroup calls it, uniformly); pack / unpack as the device's (round to nearest even, saturate).
extern float g_emu_wave_scratch[16][64];
static inline float wave_max_nonneg(float v) {
    const unsigned w = threadIdx.x / 64, lane = threadIdx.x % 64;
    g_emu_wave_scratch[w][lane] = v;
    __syncthreads();
    float m = 0.0f;
    const unsigned live = (blockDim.x - w * 64 < 64) ? (blockDim.x - w * 64) : 64;
    for (unsigned i = 0; i < live; ++i) m = std::fmax(m, g_emu_wave_scratch[w][i]);
    __syncthreads();
    return m;
}
static inline void block_scale_i16(float m, float& scale, float& inv) {
    int e;
    std::frexp(m, &e);                                            // m = f * 2^e, f in [0.5, 1): m < 2^e
    int E = (m > 0.0f) ? e + 126 : 0;
    E = (E < 15) ? 15 : E;
    scale = std::ldexp(1.0f, E - 141);
    inv = std::ldexp(1.0f, 141 - E);
}
static inline uint32_t pack_i16x2(c32 v, float inv) {
    auto q = [](float x) { const float r = std::nearbyint(x); return (int)(r > 32767.0f ? 32767.0f : (r < -32768.0f ? -32768.0f : r)); };
    return ((uint32_t)(uint16_t)(int16_t)q(v.x * inv)) | ((uint32_t)(uint16_t)(int16_t)q(v.y * inv) << 16);
}
static inline c32 unpack_i16x2(uint32_t bits, float scale) {
    return mk((float)(int16_t)(bits & 0xFFFFu), (float)(int16_t)(bits >> 16)) * scale;
}
}  // namespace ocean
#define OCEAN_TL(k)
