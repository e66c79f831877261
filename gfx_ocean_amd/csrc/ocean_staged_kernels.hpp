// ocean_staged_kernels.hpp -- the gfx950 kernels of the STAGED path and the consumers of the finished map.
//
// One kernel per reference dispatch, every intermediate observable in the reference's natural layout (ocean_read_field):
//   k_propagate / k_propagate_paired  <- shader/propagate.comp:42-72
//   k_fft_lines<ROW>, k_stage_rows    <- shader/fft_row.comp:44-63
//   k_fft_lines<COL>, k_stage_cols    <- shader/fft_col.comp:44-63
//   k_correct / k_correct_chunked     <- shader/correction.comp:24-35
// -- the compatibility path (172 B/texel as the reference; what a consumer calls per frame is the fused frame of
// ocean_kernels.hpp) -- plus
//   k_normals, k_positions            <- shader/ocean.frag:50-66, shader/ocean.vert:21-25
//   k_shard_rows, k_shard_transpose   one tile sharded by row blocks over several GPUs (SURVEY 8f #4, N <= 16384).
// No launches in this header: it is also compiled by the host emulation harness (tests/hipemu).
#pragma once
#include "ocean_kernels.hpp"

namespace ocean {

// ---------------------------------------------------------------------------------------------
// Staged kernels
// ---------------------------------------------------------------------------------------------
// One thread per 2 texels of a block of `rows` rows starting at row `row0` (the whole tile: row0 = 0, rows = N;
// grid = rows*N/2/256).  h0 / omega / outputs point at the block's first texel; h0_partner points at the first texel
// of the rows the "-k" partners live in: the block itself for the whole tile, the opposite row block
// [N - row0 - rows, N - row0) when the tile is sharded by row blocks (Q2 on only).
__global__ void __launch_bounds__(256)
k_propagate(const c32* __restrict__ h0, const c32* __restrict__ h0_partner, const float* __restrict__ omega,
            c32* __restrict__ height, c32* __restrict__ disp_x, c32* __restrict__ disp_z, int n, int row0, int rows,
            float time, float domain_size, uint32_t quirks) {
    const uint32_t un = (uint32_t)n;
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = un * (uint32_t)rows;
    const uint32_t index = pair * 2u;
    if (index >= total) return;
    const uint32_t gx = index % un, gy = index / un + (uint32_t)row0;   // gx even, gx+1 same row
    const float4 own = *reinterpret_cast<const float4*>(h0 + index);
    // index_neg = N*N-1-index (:48): row N-1-gy reversed, i.e. the partner block read backwards; the pair
    // (index, index+1) mirrors to (ineg, ineg-1).  With Q2 off the partners are columns ((N+1-gx)%N, (N-gx)%N) of
    // row (N+1-gy)%N: the same reversed pair, other base (whole-tile calls only).
    const bool q2 = (quirks & OCEAN_QUIRK_Q2) != 0u;
    const uint32_t pbase = q2 ? (total - 2u - index) : (((un + 1u - gy) & (un - 1u)) * un + ((un - gx) & (un - 1u)));
    float4 neg = *reinterpret_cast<const float4*>(h0_partner + pbase);
    if (!q2) { neg.y = -neg.y; neg.w = -neg.w; }                   // conjugated partner
    const c32 om = *reinterpret_cast<const c32*>(omega + index);
    const float ky = OCEAN_PI_F * wave_index(gy, un, quirks) / domain_size;
    const float kx0 = OCEAN_PI_F * wave_index(gx, un, quirks) / domain_size;
    const float kx1 = OCEAN_PI_F * wave_index(gx + 1u, un, quirks) / domain_size;
    const c32 h0v = propagate_height(mk(own.x, own.y), mk(neg.z, neg.w), om.x, time);
    const c32 h1v = propagate_height(mk(own.z, own.w), mk(neg.x, neg.y), om.y, time);
    float knx0, kny0, knx1, kny1;
    k_normalised(kx0, ky, knx0, kny0);
    k_normalised(kx1, ky, knx1, kny1);
    const c32 dx0 = mul_minus_i_kn(knx0, h0v), dx1 = mul_minus_i_kn(knx1, h1v);
    const c32 dz0 = mul_minus_i_kn(kny0, h0v), dz1 = mul_minus_i_kn(kny1, h1v);
    *reinterpret_cast<float4*>(height + index) = make_float4(h0v.x, h0v.y, h1v.x, h1v.y);
    *reinterpret_cast<float4*>(disp_x + index) = make_float4(dx0.x, dx0.y, dx1.x, dx1.y);
    *reinterpret_cast<float4*>(disp_z + index) = make_float4(dz0.x, dz0.y, dz1.x, dz1.y);
}

// The whole tile with the reference quirks: texel i and its "-k" partner N*N-1-i (propagate.comp:48) use the SAME two
// spectrum values with the roles swapped, so one thread does two adjacent texels AND their two partners from one pair of
// 16-byte loads: 12 instead of 20 bytes read per texel (k_propagate re-reads every h0 texel as somebody's partner:
// 738 MB counted at N = 4096 where 604 are the algorithm's).  Same arithmetic per texel as
// k_propagate: bit-identical fields.  grid = N*N/4/256.
__global__ void __launch_bounds__(256)
k_propagate_paired(const c32* __restrict__ h0, const float* __restrict__ omega, c32* __restrict__ height, c32* __restrict__ disp_x,
                   c32* __restrict__ disp_z, int n, float time, float domain_size) {
    const uint32_t un = (uint32_t)n;
    const uint32_t total = un * un;
    const uint32_t index = (blockIdx.x * 256u + threadIdx.x) * 2u;     // texels index, index + 1 of the first half ...
    if (index >= total / 2u) return;
    const uint32_t mindex = total - 2u - index;                        // ... and mindex, mindex + 1 = the partners of index + 1, index
    const float4 lo = *reinterpret_cast<const float4*>(h0 + index);
    const float4 hi = *reinterpret_cast<const float4*>(h0 + mindex);
    const c32 om_lo = *reinterpret_cast<const c32*>(omega + index);
    const c32 om_hi = *reinterpret_cast<const c32*>(omega + mindex);
    const uint32_t quirks = OCEAN_QUIRK_Q1 | OCEAN_QUIRK_Q2;
    auto texels = [&](uint32_t base, float4 own, float4 neg, c32 om) {  // the body of k_propagate for texels base, base + 1
        const uint32_t gx = base % un, gy = base / un;
        const float ky = OCEAN_PI_F * wave_index(gy, un, quirks) / domain_size;
        const float kx0 = OCEAN_PI_F * wave_index(gx, un, quirks) / domain_size;
        const float kx1 = OCEAN_PI_F * wave_index(gx + 1u, un, quirks) / domain_size;
        const c32 h0v = propagate_height(mk(own.x, own.y), mk(neg.z, neg.w), om.x, time);
        const c32 h1v = propagate_height(mk(own.z, own.w), mk(neg.x, neg.y), om.y, time);
        float knx0, kny0, knx1, kny1;
        k_normalised(kx0, ky, knx0, kny0);
        k_normalised(kx1, ky, knx1, kny1);
        const c32 dx0 = mul_minus_i_kn(knx0, h0v), dx1 = mul_minus_i_kn(knx1, h1v);
        const c32 dz0 = mul_minus_i_kn(kny0, h0v), dz1 = mul_minus_i_kn(kny1, h1v);
        *reinterpret_cast<float4*>(height + base) = make_float4(h0v.x, h0v.y, h1v.x, h1v.y);
        *reinterpret_cast<float4*>(disp_x + base) = make_float4(dx0.x, dx0.y, dx1.x, dx1.y);
        *reinterpret_cast<float4*>(disp_z + base) = make_float4(dz0.x, dz0.y, dz1.x, dz1.y);
    };
    texels(index, lo, hi, om_lo);
    texels(mindex, hi, lo, om_hi);
}

// One thread per 2 texels of a block of `lines` lines of N texels starting at line `line0` (whole tile: 0, N).
// The sign depends on the parity of x + y only, so the same kernel serves a block of rows (line = y) and, in the
// sharded transform, a block of columns stored as lines (line = x, position = y).
__global__ void __launch_bounds__(256)
k_correct(const c32* __restrict__ height, const c32* __restrict__ disp_x, const c32* __restrict__ disp_z,
          float4* __restrict__ out, int n, int line0, int lines) {
    const uint32_t un = (uint32_t)n;
    const uint32_t index = (blockIdx.x * 256u + threadIdx.x) * 2u;
    if (index >= un * (uint32_t)lines) return;
    const uint32_t x = index % un, y = index / un + (uint32_t)line0;
    const float4 h = *reinterpret_cast<const float4*>(height + index);
    const float4 dx = *reinterpret_cast<const float4*>(disp_x + index);
    const float4 dz = *reinterpret_cast<const float4*>(disp_z + index);
    const float s0 = (((x + y) & 1u) == 0u) ? -1.0f : 1.0f;        // correction.comp:29
    const float s1 = -s0;
    out[index] = make_float4(dx.x * s0, h.x * s0, dz.x * s0, 0.0f);
    out[index + 1u] = make_float4(dx.z * s1, h.z * s1, dz.z * s1, 0.0f);
}

// SURVEY 8f #1 -- the reference's "normal field" (shader/ocean.frag:50-66) as a compute kernel at
// texel centres: finite differences of one channel of the displacement map with Tile wrap
// (src/render.rs:398), height_scale 180 (:19), diff = 2/N (the shader's literal 512 -> N).
// channel 0 = disp_x is what the reference differentiates (quirk Q5); 1 = height is the
// physically meant source.
//
// The arithmetic.  With ax = x1 - x0, az = z1 - z0, d = 2/N, s = 180:
//     na = normalize(-d, ax/s, 0),  nb = normalize(0, az/s, d)                      :64-65
//     n  = normalize(cross(na, nb)) = (ax/s, d, -az/s) / |(ax/s, d, -az/s)|          :66
// (the two inner normalisations scale the cross product by 1/(|na'| |nb'|) and drop out of the outer one).  Evaluated in
// that closed form -- n = (ax, s d, -az) * rsqrt(ax^2 + az^2 + (s d)^2): one v_rsq_f32, no division -- instead of the
// literal three normalize() with a square root and three IEEE divisions each, which made the kernel VALU-issue-bound
// (9 divisions + 3 square roots = ~180 issue slots per texel: 19 of the 25 us at N = 2048; round 5).  As close
// to the fp64 value as the literal form (1.2e-7 either way on the reference's data, 1.8e-7 from each other: restated in numpy;
// tests/test_gpu_parity.py::test_normal_field holds both to the literal
// fp32 restatement within 1e-5 absolute; GLSL's own normalize() is implementation-defined, usually v * inversesqrt(dot(v, v))).
__device__ __forceinline__ float4 normal_from_differences(float ax, float az, float sd) {
#ifdef OCEAN_NORMALS_LITERAL                                         // A/B only (tools/build_variants.py): the literal evaluation
    const float d = sd / 180.0f;
    const float ay = ax / 180.0f, by = az / 180.0f;
    const float la = sqrtf(d * d + ay * ay), lb = sqrtf(by * by + d * d);
    const float nax = -d / la, nay = ay / la, nby = by / lb, nbz = d / lb;
    const float cx = nay * nbz, cy = -nax * nbz, cz = nax * nby;
    const float lc = sqrtf(cx * cx + cy * cy + cz * cz);
    return make_float4(cx / lc, cy / lc, cz / lc, 0.0f);
#else
    const float r = rsqrtf(fmaf(ax, ax, fmaf(az, az, sd * sd)));
    return make_float4(ax * r, sd * r, -az * r, 0.0f);
#endif
}
// A workgroup owns 256 columns x ROWS rows; a thread walks down its column with the rows above and below in registers,
// so a row is fetched once per workgroup (plus two halo rows per ROWS) instead of three times by three workgroups, and
// the x neighbours are hits in the lines the centre load brought.  Streamed stores.  grid = (N / 256) * (N / ROWS);
// N >= 256.  Reads whole RGBA texels for one channel: 16 + 16 B/texel (the frame with normals uses k_normals_plane).
// ROWS per size (measured 4 / 8 / 16; the small sizes need the workgroups):
constexpr int normals_rows(int n) { return (n >= 8192) ? 8 : ((n >= 2048) ? 4 : ((n >= 1024) ? 2 : 1)); }
template <int ROWS>
__global__ void __launch_bounds__(256)
k_normals(const float4* __restrict__ rgba, float4* __restrict__ normals, int n, int channel) {
    const uint32_t un = (uint32_t)n;
    const uint32_t col_blocks = un / 256u;
    const uint32_t x = (blockIdx.x % col_blocks) * 256u + threadIdx.x, y0 = (blockIdx.x / col_blocks) * ROWS;
    const uint32_t xm = (x + un - 1u) % un, xp = (x + 1u) % un;
    const float* f = reinterpret_cast<const float*>(rgba) + channel;
    const float sd = 360.0f / (float)n;                            // height_scale * diff = 180 * 2 / N   (:19, :52)
    float above = f[((size_t)((y0 + un - 1u) % un) * un + x) * 4], centre = f[((size_t)y0 * un + x) * 4];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t y = y0 + (uint32_t)r;
        const float below = f[((size_t)((y + 1u) % un) * un + x) * 4];
        const float x0 = f[((size_t)y * un + xm) * 4], x1 = f[((size_t)y * un + xp) * 4];
        store_float4_nt(normals + (size_t)y * un + x, normal_from_differences(x1 - x0, below - above, sd));
        above = centre;
        centre = below;
    }
}
// The same field from the dense fp32 plane of the source channel that pass 2 of the fused frame stores next to the map
// (k_half_pass2<.., PLANE>): 4 + 16 instead of 16 + 16 B/texel.  A wave owns 256 adjacent columns x ROWS rows: lane l holds
// columns l, l + 64, l + 128, l + 192 of the segment (every load instruction of the wave is one contiguous 256-byte piece
// of a plane row, every store instruction one contiguous KiB of a normals row) and walks down the rows with the rows above
// and below in registers; the x neighbours are 4-byte hits in the lines the centre loads brought.  Bands of rows go to the
// XCDs in contiguous ranges, so that a band's two halo rows are L2 hits.  grid = (N / 256) * (N / ROWS) / 4 workgroups of
// four waves; N >= 256, N / ROWS >= 4.
// Same arithmetic on the same floats as k_normals: bit-identical normals (tests/test_gpu_parity.py).
// (rows per wave: 16 against 8 at N = 4096: 50.9-51.2 against 52.2-52.4 us, two halo rows per 16 instead of per 8; r05_run7)
constexpr int normals_plane_rows(int n) { return (n >= 4096) ? 16 : ((n >= 2048) ? 8 : ((n >= 1024) ? 4 : 2)); }
template <int ROWS>
__global__ void __launch_bounds__(256)
k_normals_plane(const float* __restrict__ plane, float4* __restrict__ normals, int n) {
    // (a batched launch -- ocean_frame_batch / ocean_frame_tiles with the normal field, N <= 1024: blockIdx.y = frame, planes and
    //  fields N * N elements apart; gridDim.y = 1 and blockIdx.y = 0 otherwise)
    plane += (size_t)blockIdx.y * (size_t)n * (size_t)n;
    normals += (size_t)blockIdx.y * (size_t)n * (size_t)n;
    const uint32_t un = (uint32_t)n, mask = un - 1u, segs = un / 256u;
    const uint32_t gw = (uint32_t)xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * 4u + (threadIdx.x >> 6);   // wave of the grid
    const uint32_t x0 = (gw % segs) * 256u + (threadIdx.x & 63u), y0 = (gw / segs) * ROWS;
    const float sd = 360.0f / (float)n;                            // height_scale * diff = 180 * 2 / N
    float above[4], centre[4];
    {
        const float* pa = plane + (size_t)((y0 + un - 1u) & mask) * un + x0;
        const float* pc = plane + (size_t)y0 * un + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { above[k] = pa[64 * k]; centre[k] = pc[64 * k]; }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t y = y0 + (uint32_t)r;
        const float* pb = plane + (size_t)((y + 1u) & mask) * un + x0;
        const float* prow = plane + (size_t)y * un;
        float below[4], left[4], right[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            below[k] = pb[64 * k];
            left[k] = prow[(x0 + 64u * k + un - 1u) & mask];
            right[k] = prow[(x0 + 64u * k + 1u) & mask];
        }
        float4* o = normals + (size_t)y * un + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            store_float4_nt(o + 64 * k, normal_from_differences(right[k] - left[k], below[k] - above[k], sd));
            above[k] = centre[k];
            centre[k] = below[k];
        }
    }
}

// The same, one workgroup per band of ROWS whole rows (N >= 8192, where the field's N*N*16 bytes of stores -- 1 GiB at 8192 -- are
// the bound and go to HBM one way or the other): the workgroup walks along the rows, 1024 columns per step, so that what it writes
// is ONE contiguous span of ROWS * N * 16 bytes (contiguous spans per workgroup stream at 6.2 TB/s on this part, fine-interleaved
// stores at 4.5: tools/membench2.hip).  grid = N / ROWS workgroups of 256 threads.  Measured at N = 8192 (r05_run14, one box, two
// repetitions): k_normals_plane<16> 312 us; bands of 4 / 8 / 16 rows 291 / 266 / 303 us; whole rows one after the other with the rows
// above and below re-read from the caches: 348-350 us.  Bands of 8: 1342 MB at 5.0 TB/s.  Same arithmetic: the same bits.
constexpr int NORMALS_BAND_ROWS = 8;
template <int ROWS>
__global__ void __launch_bounds__(256)
k_normals_plane_bands(const float* __restrict__ plane, float4* __restrict__ normals, int n) {
    const uint32_t un = (uint32_t)n, mask = un - 1u;
    const uint32_t y0 = (uint32_t)xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * ROWS;
    const float sd = 360.0f / (float)n;
    for (uint32_t xb = 0; xb < un; xb += 1024u) {
        const uint32_t x0 = xb + (threadIdx.x >> 6) * 256u + (threadIdx.x & 63u);
        float above[4], centre[4];
        const float* pa = plane + (size_t)((y0 + un - 1u) & mask) * un + x0;
        const float* pc = plane + (size_t)y0 * un + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { above[k] = pa[64 * k]; centre[k] = pc[64 * k]; }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t y = y0 + (uint32_t)r;
            const float* pb = plane + (size_t)((y + 1u) & mask) * un + x0;
            const float* prow = plane + (size_t)y * un;
            float below[4], left[4], right[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                below[k] = pb[64 * k];
                left[k] = prow[(x0 + 64u * k + un - 1u) & mask];
                right[k] = prow[(x0 + 64u * k + 1u) & mask];
            }
            float4* o = normals + (size_t)y * un + x0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                store_float4_nt(o + 64 * k, normal_from_differences(right[k] - left[k], below[k] - above[k], sd));
                above[k] = centre[k];
                centre[k] = below[k];
            }
        }
    }
}

// SURVEY 8f #2 -- the vertex stage's consumer of the map (shader/ocean.vert:21-25) as a compute kernel: the
// V x V patch grid of src/render.rs:494-508 (a_Pos = (x, 0, z), a_Uv = (x, z) / (V - 1); V = HALF_RESOLUTION
// = 128 in the reference), the displacement sampled with the reference's sampler (Filter::Linear,
// WrapMode::Tile, src/render.rs:398: bilinear at texel coordinates uv * N - 0.5 with wrap, weights in fp32),
// displacement.y /= 3.0, .xz /= 3.5 (:22-23), pos = a_Pos + displacement + (offset.x, 0, offset.y) (:25).
// One thread per vertex; positions[z * V + x] = (pos.x, pos.y, pos.z, 1).
__global__ void __launch_bounds__(256)
k_positions(const float4* __restrict__ rgba, float4* __restrict__ positions, int n, int verts, float offset_x,
            float offset_z) {
    const uint32_t index = blockIdx.x * 256u + threadIdx.x;
    const uint32_t uv_ = (uint32_t)verts;
    if (index >= uv_ * uv_) return;
    const uint32_t vx = index % uv_, vz = index / uv_;
    const float den = (float)(verts - 1);
    const float u = (float)vx / den, v = (float)vz / den;          // `(x as f32) / (V - 1) as f32`, src/render.rs:503-504
    const float tx = u * (float)n - 0.5f, ty = v * (float)n - 0.5f;
    const float fx = floorf(tx), fy = floorf(ty);
    const float wx = tx - fx, wy = ty - fy;
    const uint32_t mask = (uint32_t)n - 1u;                        // N is a power of two: Tile wrap
    const uint32_t x0 = (uint32_t)(int32_t)fx & mask, x1 = (x0 + 1u) & mask;
    const uint32_t y0 = (uint32_t)(int32_t)fy & mask, y1 = (y0 + 1u) & mask;
    const float4 a = rgba[(size_t)y0 * n + x0], b = rgba[(size_t)y0 * n + x1];
    const float4 c = rgba[(size_t)y1 * n + x0], d = rgba[(size_t)y1 * n + x1];
    const float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy), w01 = (1.0f - wx) * wy, w11 = wx * wy;
    const float dx = a.x * w00 + b.x * w10 + c.x * w01 + d.x * w11;
    const float dy = a.y * w00 + b.y * w10 + c.y * w01 + d.y * w11;
    const float dz = a.z * w00 + b.z * w10 + c.z * w01 + d.z * w11;
    positions[index] = make_float4((float)vx + dx / 3.5f + offset_x, dy / 3.0f, (float)vz + dz / 3.5f + offset_z, 1.0f);
}

// In-place line FFT on the natural layout.  LPW lines per workgroup (across threads).
// ROW: line = row y, element pos at data[y*N + pos], threads of a line are contiguous lanes.
// COL: line = column x, element pos at data[pos*N + x]; the LPW columns of a workgroup are
//      adjacent and the line index is the fastest thread coordinate (8*LPW-byte pieces).
template <int N, int E, int LPW, bool COL>
__global__ void __launch_bounds__((N / E) * LPW)
k_fft_lines(c32* __restrict__ data, const c32* __restrict__ tw) {
    constexpr int T = N / E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = COL ? (tid % LPW) : ((T >= 64) ? wave_uniform(tid / T) : (tid / T));
    const int j = COL ? (tid / LPW) : (tid % T);
    const int group = COL ? xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int line = group * LPW + ll;
    c32* lds_line = lds + ll * LinePitch<N>::elems;
    c32 reg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int pos = j + e * T;
        reg[e] = COL ? data[(size_t)pos * N + line] : data[(size_t)line * N + pos];
    }
    fft_line<N, E>(reg, j, tw, lds_line);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int pos = j + e * T;
        if (COL) data[(size_t)pos * N + line] = reg[e];
        else data[(size_t)line * N + pos] = reg[e];
    }
}

// ---------------------------------------------------------------------------------------------
// Staged column pass at N >= 8192 in two steps ("four-step" FFT; shader/fft_col.comp:44-63 for lines that do not fit the LDS
// sixteen at a time)
// ---------------------------------------------------------------------------------------------
// k_fft_lines<COL> keeps whole columns in LDS: two 8192-point columns fill it, so a workgroup owns 16-byte pieces of every row
// and the pass runs at 1.26 TB/s (0.85 ms per field at N = 8192).  With NF = S * M, row n = r + S m and
// output row k = k1 + M k2:
//     X[k1 + M k2] = sum_{r < S} W_S^{r k2} * ( W_NF^{r k1} * sum_{m < M} x[r + S m] W_M^{m k1} ),      W_n = e^{+2 pi i / n}.
// Step A (k_cols4_a, in place): per residue r the M-point transforms along m of LPW = 16 adjacent columns -- a workgroup moves
//   whole 128-byte pieces (16 columns x 8 bytes of a row) -- times W_NF^{r k1}; Y[r][k1] stays at row r + S k1.  M = 512 (S = 16
//   at 8192, 32 at 16384): 16 lines are 70 KiB of LDS and 512 threads, so TWO workgroups share a CU and one loads while the
//   other transforms (M = 1024, one 1024-thread workgroup per CU in lock-step phases: 0.56-0.58 ms per field at 8192 instead of
//   0.44; 32 columns of 512 points: 0.55; 8 columns of 1024: 0.50 -- r05_run12).
// Step B (k_cols4_b, out of place): for every k1 the S values Y[0..S)[k1] are S CONSECUTIVE rows; an S-point DFT in registers,
//   results to rows k1 + M k2 of the destination: whole rows in, whole rows out, no LDS.
// 32 instead of 16 B/texel per field, all of it in full lines.  The destination of step B becomes the field (the API swaps the
// two buffers), so nothing is copied back.
template <int NF, int S, int E, int LPW>
__global__ void __launch_bounds__((NF / S / E) * LPW)
k_cols4_a(c32* __restrict__ data, const c32* __restrict__ tw) {
    constexpr int M = NF / S, T = M / E;
    static_assert(NF == S * M && M % E == 0 && T % 16 == 0, "four-step geometry (fft_line: threads per line a multiple of 16)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = tid % LPW, j = tid / LPW;                       // the line (column) is the fastest thread coordinate
    const int groups = NF / LPW;
    const int b = (int)blockIdx.x;
    const int r = b / groups;                                      // residue of the rows this workgroup transforms
    const int x = xcd_contiguous(b % groups, groups) * LPW + ll;
    c32* lds_line = lds + ll * LinePitch<M>::elems;
    c32* col = data + (size_t)r * NF + x;                          // row r + S m, column x
    c32 reg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) reg[e] = col[(size_t)(j + e * T) * S * NF];
    fft_line<M, E, S>(reg, j, tw, lds_line);                       // (table of NF points: stride S)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k1 = j + e * T;
        col[(size_t)k1 * S * NF] = (r == 0) ? reg[e] : cmul(reg[e], tw[r * k1]);   // W_NF^{r k1}, r k1 < NF
    }
}
// One thread per CPT adjacent columns (2; 1 at S = 32: 4 S registers per column) and one k1; grid = (NF / CPT / 256) * M
// workgroups of 256 threads.
template <int NF, int S>
__global__ void __launch_bounds__(256)
k_cols4_b(const c32* __restrict__ src, c32* __restrict__ dst) {
    constexpr int M = NF / S;
    constexpr int CPT = (S >= 32) ? 1 : 2;
    constexpr int PER_ROW = NF / CPT / 256;                        // workgroups per row
    const int k1 = (int)blockIdx.x / PER_ROW;
    const int x = (((int)blockIdx.x % PER_ROW) * 256 + (int)threadIdx.x) * CPT;
    const c32* in = src + (size_t)k1 * S * NF + x;
    c32* out = dst + (size_t)k1 * NF + x;
    if constexpr (CPT == 2) {
        c32 a[S], b[S], oa[S], ob[S];
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(in + (size_t)r * NF);
            a[r] = mk(v.x, v.y);
            b[r] = mk(v.z, v.w);
        }
        Dft<S>::run(a, oa);
        Dft<S>::run(b, ob);
#pragma unroll
        for (int k2 = 0; k2 < S; ++k2)
            store_float4_nt(reinterpret_cast<float4*>(out + (size_t)k2 * M * NF), make_float4(oa[k2].x, oa[k2].y, ob[k2].x, ob[k2].y));
    } else {
        c32 a[S], oa[S];
#pragma unroll
        for (int r = 0; r < S; ++r) a[r] = in[(size_t)r * NF];
        Dft<S>::run(a, oa);
#pragma unroll
        for (int k2 = 0; k2 < S; ++k2) out[(size_t)k2 * M * NF] = oa[k2];
    }
}

// Step B of the three fields AND the correction (shader/correction.comp:24-35) in one kernel: when ocean_correct follows the
// three column passes (the reference's order, src/render.rs:1210-1287) the S-point step has not run yet (the API defers it: see
// launch_cols / settle_field in ocean_api.hip) and its only consumer wants the real parts with the sign.  24 R + 16 W instead of
// 3 x (8 R + 8 W) + 24 R + 16 W bytes per texel; the step-A fields stay as they are, so ocean_read_field afterwards still runs
// k_cols4_b on them.  One thread per column and k1; rows k1 + M k2 have the parity of k1 (M is even).
template <int NF, int S>
__global__ void __launch_bounds__(256)
k_cols4_b_correct(const c32* __restrict__ height, const c32* __restrict__ disp_x, const c32* __restrict__ disp_z,
                  float4* __restrict__ out) {
    constexpr int M = NF / S;
    constexpr int PER_ROW = NF / 256;
    static_assert(M % 2 == 0, "the sign of a row follows k1");
    const int k1 = (int)blockIdx.x / PER_ROW;
    const int x = ((int)blockIdx.x % PER_ROW) * 256 + (int)threadIdx.x;
    const size_t in = (size_t)k1 * S * NF + x;
    float re[3][S];
    const c32* const src[3] = {disp_x, height, disp_z};            // the map's channel order
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        c32 a[S], oa[S];
#pragma unroll
        for (int r = 0; r < S; ++r) a[r] = src[f][in + (size_t)r * NF];
        Dft<S>::run(a, oa);
#pragma unroll
        for (int k2 = 0; k2 < S; ++k2) re[f][k2] = oa[k2].x;
    }
    const float s = (((x + k1) & 1) == 0) ? -1.0f : 1.0f;          // correction.comp:29
    float4* o = out + (size_t)k1 * NF + x;
#pragma unroll
    for (int k2 = 0; k2 < S; ++k2)
        store_float4_nt(o + (size_t)k2 * M * NF, make_float4(re[0][k2] * s, re[1][k2] * s, re[2][k2] * s, 0.0f));
}

// ---------------------------------------------------------------------------------------------
// Staged path, chunked hand-off (N <= 4096, 4 x 4 chunks)
// ---------------------------------------------------------------------------------------------
// The reference's column pass reads its lines with a 4 KiB lane stride (shader/fft_col.comp:45-47) and leaves it to
// the L2 to merge neighbouring columns; k_fft_lines<COL> does the same with 32-byte pieces and reaches a quarter
// of the HBM roofline.  The staged calls keep the reference's dispatch structure (ocean_fft_rows, then
// ocean_fft_cols, per field) but hand the field over in the 4 x 4-chunk layout of the fused path, so that BOTH passes
// move whole 128-byte lines:
//   k_stage_rows:   4 natural rows in (contiguous) -> row FFT -> chunk row Y out: one contiguous 4 N-element span;
//   k_stage_cols:   the 4 columns of chunk column X, in place on the chunked field (whole chunks in and out);
//   k_correct_chunked / k_unchunk: the consumers (correction.comp:24-35, ocean_read_field) read the chunked field.
// Results in the natural layout are available after every call through ocean_read_field (include/ocean_hip.h).
template <int N, int E>
__global__ void __launch_bounds__((N / E) * 4)
k_stage_rows(const c32* __restrict__ nat, c32* __restrict__ chk, const c32* __restrict__ tw, InterLayout lay) {
    constexpr int T = N / E, THREADS = 4 * T;
    static_assert(CHUNK_W == 4 && CHUNK_R == 4, "the chunked staged path is written for 4 x 4 chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int r = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    const int Y = blockIdx.x;
    c32* lds_line = lds + r * LinePitch<N>::elems;
    c32 reg[E];
    const c32* src = nat + (size_t)(Y * 4 + r) * N + j;
#pragma unroll
    for (int e = 0; e < E; ++e) reg[e] = src[e * T];
    fft_line_to_lds<N, E>(reg, j, tw, lds_line);
    // the chunk row is 2 N pieces of 16 bytes (chunk X, piece p: row p / 2, columns 2 (p % 2) and + 1), contiguous in
    // this order: consecutive lanes store consecutive pieces
    float4* dst = reinterpret_cast<float4*>(chk + (size_t)Y * lay.sy);
#pragma unroll
    for (int q = 0; q < (2 * N) / THREADS; ++q) {
        const int idx = tid + q * THREADS;
        const int p = idx & 7;
        const int x = (idx >> 3) * 4 + 2 * (p & 1);
        const c32* l = lds + (p >> 1) * LinePitch<N>::elems;
        const c32 v0 = l[lds_pad(x)], v1 = l[lds_pad(x + 1)];
        store_float4_nt(dst + idx, make_float4(v0.x, v0.y, v1.x, v1.y));
    }
}

template <int N, int E>
__global__ void __launch_bounds__((N / E) * 4)
k_stage_cols(c32* __restrict__ chk, const c32* __restrict__ tw, InterLayout lay) {
    constexpr int T = N / E, THREADS = 4 * T;
    static_assert(CHUNK_W == 4 && CHUNK_R == 4, "the chunked staged path is written for 4 x 4 chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int X = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    c32* base = chk + (size_t)X * lay.sx;                          // chunk (X, Y) at + Y * lay.sy
    // whole chunks in: piece idx -> chunk Y = idx / 8, piece p: row y = 4 Y + p / 2, columns 2 (p % 2) and + 1
#pragma unroll
    for (int q = 0; q < (2 * N) / THREADS; ++q) {
        const int idx = tid + q * THREADS;
        const int p = idx & 7;
        const int y = (idx >> 3) * 4 + (p >> 1);
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(idx >> 3) * lay.sy + 2 * p);
        c32* l = lds + (2 * (p & 1)) * LinePitch<N>::elems + lds_pad(y);
        l[0] = mk(v.x, v.y);
        l[LinePitch<N>::elems] = mk(v.z, v.w);
    }
    __syncthreads();
    const int c = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    c32* lds_line = lds + c * LinePitch<N>::elems;
    c32 reg[E];
    {
        const c32* g = lds_line + lds_pad(j);
#pragma unroll
        for (int e = 0; e < E; ++e) reg[e] = g[e * (T + T / 16)];
    }
    __syncthreads();
    fft_line_to_lds<N, E>(reg, j, tw, lds_line);
    // whole chunks out, same addresses (every chunk of this column group was read above): thread -> (column pair h, row)
    const int h = tid & 1;
    const int i = tid >> 1;
    const c32* l0 = lds + (2 * h) * LinePitch<N>::elems;
    const c32* l1 = l0 + LinePitch<N>::elems;
    c32* dst = base + (size_t)(i / 4) * lay.sy + (i % 4) * 4 + 2 * h;
#pragma unroll
    for (int q = 0; q < E / 2; ++q) {
        const int y = i + q * (2 * T);
        const c32 v0 = l0[lds_pad(y)], v1 = l1[lds_pad(y)];
        store_float4_nt(reinterpret_cast<float4*>(dst + (size_t)q * ((2 * T) / 4) * lay.sy), make_float4(v0.x, v0.y, v1.x, v1.y));
    }
}

// One workgroup per chunk row Y (4 rows of the field): wave w owns row 4 Y + w, lanes run along x, so the RGBA /
// natural stores are whole rows; the four waves read the same 128-byte lines at the same time (one HBM fetch).
__device__ __forceinline__ size_t chunked_index(InterLayout lay, int Y, int r, int x) {
    return (size_t)Y * lay.sy + (size_t)(x >> 2) * lay.sx + r * 4 + (x & 3);
}
__global__ void __launch_bounds__(256)
k_correct_chunked(const c32* __restrict__ height, const c32* __restrict__ disp_x, const c32* __restrict__ disp_z,
                  float4* __restrict__ out, int n, InterLayout lay) {
    const int Y = blockIdx.x, r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y = 4 * Y + r;
    for (int x = lane; x < n; x += 64) {
        const size_t i = chunked_index(lay, Y, r, x);
        const float s = (((x + y) & 1) == 0) ? -1.0f : 1.0f;       // correction.comp:29
        store_float4_nt(out + (size_t)y * n + x, make_float4(disp_x[i].x * s, height[i].x * s, disp_z[i].x * s, 0.0f));
    }
}
__global__ void __launch_bounds__(256)
k_unchunk(const c32* __restrict__ chk, c32* __restrict__ nat, int n, InterLayout lay) {
    const int Y = blockIdx.x, r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int x = lane; x < n; x += 64) nat[(size_t)(4 * Y + r) * n + x] = chk[chunked_index(lay, Y, r, x)];
}

// ---------------------------------------------------------------------------------------------
// One N x N transform sharded by row blocks over `world` GPUs (N up to 16384)
// ---------------------------------------------------------------------------------------------
// Rank r owns rows [r N/world, (r+1) N/world) of the three spectra: k_propagate on its block, the row pass below,
// ONE all-to-all, then the column pass on the N/world columns it receives (k_shard_transpose + k_fft_lines<ROW> +
// k_correct).  The row pass stores straight into the all-to-all send buffer
//     send[dest][field][row][column of dest]      (dest = x / cols; pieces of `cols` contiguous elements),
// and the receive buffer recv[src][field][row of src][my column] is the column block in row-major order.
template <int N, int E, int LPW>
__global__ void __launch_bounds__((N / E) * LPW)
k_shard_rows(const c32* __restrict__ block, c32* __restrict__ send, const c32* __restrict__ tw, int field, int rows,
             int cols_log2) {
    constexpr int T = N / E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int ll = (T >= 64) ? wave_uniform(tid / T) : (tid / T);
    const int j = tid % T;
    const int line = (int)blockIdx.x * LPW + ll;                   // local row
    c32* lds_line = lds + ll * LinePitch<N>::elems;
    c32 reg[E];
    const c32* src = block + (size_t)line * N + j;
#pragma unroll
    for (int e = 0; e < E; ++e) reg[e] = src[e * T];
    fft_line<N, E>(reg, j, tw, lds_line);
    const int cols = 1 << cols_log2;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int x = j + e * T;
        const int dest = x >> cols_log2;
        send[(((size_t)dest * 3 + field) * rows + line) * cols + (x & (cols - 1))] = reg[e];
    }
}

// recv[src][field][rows][cols] (field f of the column block, row-major: y = src * rows + row) -> out[column][y],
// 32 x 32 tiles through LDS (32 * 33 * 8 bytes, dynamic), both sides in 256-byte pieces.
// grid = (N / 32) * (cols / 32), 256 threads.
__global__ void __launch_bounds__(256)
k_shard_transpose(const c32* __restrict__ recv, c32* __restrict__ out, int n, int field, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32 (*tile)[33] = reinterpret_cast<c32 (*)[33]>(smem);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_y = n / 32;
    const int y0 = ((int)blockIdx.x % tiles_y) * 32, c0 = ((int)blockIdx.x / tiles_y) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + ty + 8 * k;
        const int src = y / rows, ry = y - src * rows;
        tile[ty + 8 * k][tx] = recv[(((size_t)src * 3 + field) * rows + ry) * cols + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(size_t)(c0 + ty + 8 * k) * n + y0 + tx] = tile[tx][ty + 8 * k];
}

}  // namespace ocean
