// ocean_aux_kernels.hpp -- small streaming kernels around the displacement map (device-side consumers of the
// hot path's output; not part of the two fused launches):
//   k_checksum       order-independent 64-bit checksum of a device buffer (reproducibility / race tests, SURVEY 5;
//                    the reference's own discipline is the barrier chain of shader/fft_row.comp:48-59);
//   k_pack_rgb32f    (disp_x, height, disp_z, 0) -> (disp_x, height, disp_z): 12 instead of 16 B/texel for the final
//                    gather of BASELINE config 4 (192 MiB instead of 256 MiB per N = 4096 tile);
//   k_pack_height32f the height channel alone: 4 B/texel (64 MiB per tile).
// shader/correction.comp:31-34 defines the RGBA texel these read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocean {

// sum over 32-bit words w_i of (w_i + golden) * (2 i + 1)  mod 2^64.  Integer adds commute, so the block/atomic
// reduction order does not matter: equal buffers <=> equal sums (up to 64-bit collisions), any launch geometry.
__global__ void __launch_bounds__(256)
k_checksum(const uint4* __restrict__ data, size_t vec_count, unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long part[256];
    unsigned long long h = 0;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < vec_count; i += (size_t)gridDim.x * 256u) {
        const uint4 v = data[i];
        const unsigned long long b = 8ull * i + 1ull;                  // 2 * (4 i + k) + 1
        h += (unsigned long long)(v.x + 0x9E3779B9u) * b;
        h += (unsigned long long)(v.y + 0x9E3779B9u) * (b + 2ull);
        h += (unsigned long long)(v.z + 0x9E3779B9u) * (b + 4ull);
        h += (unsigned long long)(v.w + 0x9E3779B9u) * (b + 6ull);
    }
    part[threadIdx.x] = h;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(acc, part[0]);
}

// One thread per 4 texels: 64 contiguous bytes in, 48 (rgb) or 16 (height) contiguous bytes out.
__global__ void __launch_bounds__(256)
k_pack_rgb32f(const float4* __restrict__ rgba, float4* __restrict__ rgb, size_t quads) {
    const size_t q = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= quads) return;
    const float4 a = rgba[4 * q], b = rgba[4 * q + 1], c = rgba[4 * q + 2], d = rgba[4 * q + 3];
    rgb[3 * q] = make_float4(a.x, a.y, a.z, b.x);
    rgb[3 * q + 1] = make_float4(b.y, b.z, c.x, c.y);
    rgb[3 * q + 2] = make_float4(c.z, d.x, d.y, d.z);
}
__global__ void __launch_bounds__(256)
k_pack_height32f(const float4* __restrict__ rgba, float4* __restrict__ height, size_t quads) {
    const size_t q = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= quads) return;
    height[q] = make_float4(rgba[4 * q].y, rgba[4 * q + 1].y, rgba[4 * q + 2].y, rgba[4 * q + 3].y);
}

// ---- upload-time re-layout (once per ocean_upload_spectrum; the reference's staging copy, src/render.rs:872-924) --------
// The fused path reads the static inputs transposed (h0T[x][y] = h0[y][x], omegaT likewise): 32 x 32 tiles through LDS,
// both sides in contiguous pieces.  The source is a WINDOW of the natural-layout array: rows [y0, y0 + 32 ytiles) and columns
// from x0 on, `pitch` elements per row -- the whole array (pitch n, origins 0), or the slab of rows / band of columns a fused-only
// context stages at a time (ocean_api.hip upload_common).  Of that window the source columns (= destination lines)
// [32 xt0, 32 (xt0 + xtiles)) are transposed.  grid = xtiles * ytiles, 256 threads.
struct NaturalWindow {
    int pitch;   // elements per source row
    int x0, y0;  // the array coordinates of the window's first element
};
template <typename T>
__global__ void __launch_bounds__(256)
k_transpose(const T* __restrict__ src, NaturalWindow w, T* __restrict__ dst, int n, int xt0, int xtiles) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = (xt0 + (int)blockIdx.x % xtiles) * 32, y0 = w.y0 + ((int)blockIdx.x / xtiles) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = src[(size_t)(y0 - w.y0 + ty + 8 * k) * w.pitch + (x0 - w.x0) + tx];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[(size_t)(x0 + ty + 8 * k) * n + y0 + tx] = tile[tx][ty + 8 * k];
}
// BASELINE config 5: h0 * 2^scale_log2 rounded to fp16 (nearest even) -> packed (re, im) pairs, transposed, for the fused
// path; and the natural-layout fp32 source is REPLACED by the dequantised values, so that every kernel (and
// ocean_read_spectrum) uses exactly the numbers the fp16 storage holds.  Same window and grid as k_transpose.
__global__ void __launch_bounds__(256)
k_quantise_f16_transpose(float2* __restrict__ h0, NaturalWindow w, uint32_t* __restrict__ packedT, int n, int xt0, int xtiles, float up, float down) {
    __shared__ uint32_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = (xt0 + (int)blockIdx.x % xtiles) * 32, y0 = w.y0 + ((int)blockIdx.x / xtiles) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)(y0 - w.y0 + ty + 8 * k) * w.pitch + (x0 - w.x0) + tx;
        const float2 v = h0[i];
        const _Float16 re = (_Float16)(v.x * up), im = (_Float16)(v.y * up);       // round to nearest even
        h0[i] = make_float2((float)re * down, (float)im * down);
        tile[ty + 8 * k][tx] = (uint32_t)__builtin_bit_cast(unsigned short, re) | ((uint32_t)__builtin_bit_cast(unsigned short, im) << 16);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) packedT[(size_t)(x0 + ty + 8 * k) * n + y0 + tx] = tile[tx][ty + 8 * k];
}

}  // namespace ocean
