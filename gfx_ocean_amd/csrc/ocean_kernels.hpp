// ocean_kernels.hpp -- the gfx950 kernels of the FUSED frame (ocean_frame), the hot path of gfx-ocean.
//
// 2 launches, the half-spectrum "real-output" algorithm, 52-54 B/texel of HBM traffic instead of the reference's 172:
//   k_half_pass1 / k_half_pass1_split (N = 8192): propagate (shader/propagate.comp:42-72) + symmetrise + FFT along y
//                  (shader/fft_col.comp:44-63) of the half spectrum's columns, reading the *transposed* static inputs (h0T,
//                  omegaT; made once at upload) so that every line is contiguous -- at N >= 2048 streamed through LDS by
//                  LDS-DMA (half_load_AB_dma); writes the intermediate as 4 x 4 chunks (128 bytes);
//   k_half_pass2 (N <= 4096): rebuild full rows from the half spectrum, two complex FFTs along x
//                  (shader/fft_row.comp:44-63) for the three real channels, sign correction and RGBA pack
//                  (shader/correction.comp:24-35), whole-row stores;
//   k_half_pass2_real (N >= 8192): the same rows as three real-output transforms of N/2 complex points each.
// The column transform runs first (a separable 2-D DFT commutes), so that the pass that owns whole rows is the one that
// writes the row-major RGBA image.  One tile over several GPUs: the same kernels on column / row blocks (x_group0, SHARD).
// The staged 1:1 kernels, the map's consumers and the row-block sharding kernels: ocean_staged_kernels.hpp.
//
// Every kernel in this file ships; measured dead ends live in git history and DESIGN.md 4.3-4.6, not here.
// No launches in this header: it is also compiled by the host emulation harness (tests/hipemu) that checks the index
// algebra on the CPU.
#pragma once
#include <utility>

#include "fft_core.hpp"

namespace ocean {

// The fused kernels take the base twiddle of every pass from v_sin / v_cos instead of the table (fft_core.hpp base_twiddle;
// the reference evaluates cos/sin per butterfly, shader/fft_row.comp:32-33): their workgroups run in lock-step phases, and a
// dependent table read behind every barrier of a transform is exposed latency.  First the latency-bound sizes (round 3:
// N <= 1024); measured again in round 4 after the other such reads were gone (r04_run28/29, one box each, two repetitions):
// N = 2048 +2.5-3 % frames/s, 4096 +1.4 % (pass 2 89.2-89.5 -> 86.0-86.5 us), 8192 pass 1 -3 %, 16384 pass 2 -8 %.
// (The staged kernels keep the table: they are the 1:1 restatement.)
constexpr bool FUSED_HWTW = true;

// shader/propagate.comp:6 -- `const float pi = 3.1415926;` (fp32 0x40490FDA)
#define OCEAN_PI_F 3.1415926f

// Observed dispatch rule: block b runs on XCD b % 8.  Give each XCD a contiguous range of
// logical work items so neighbours share an L2 (speed only; any mapping is correct).
__device__ __forceinline__ int xcd_contiguous(int b, int nblocks) {
    if ((nblocks & 7) != 0) return b;
    return (b & 7) * (nblocks >> 3) + (b >> 3);
}

// Q1: float(uint(2*g - N - 1))  (shader/propagate.comp:45-46, wraps in uint32)
__device__ __forceinline__ float wave_index_q1(uint32_t g, uint32_t n) {
    return (float)(uint32_t)(2u * g - n - 1u);
}
// Quirk switches of the staged path (include/ocean_hip.h OCEAN_QUIRK_*).  Both set = the reference.
//   Q1 off: the wave index 2g - N - 1 is evaluated signed (what the shader's author meant).
//   Q2 off: the "-k" partner of texel g is (N + 1 - g) % N on both axes (k is antisymmetric about (N+1)/2;
//           the two texels without a partner, g = 0 and 1, pair with each other) and enters conjugated.
#define OCEAN_QUIRK_Q1 1u
#define OCEAN_QUIRK_Q2 2u
__device__ __forceinline__ float wave_index(uint32_t g, uint32_t n, uint32_t quirks) {
    return (quirks & OCEAN_QUIRK_Q1) ? wave_index_q1(g, n) : (float)((int32_t)(2u * g) - (int32_t)n - 1);
}

// The per-texel math of propagate.comp:55-71, shared by the staged and the fused kernel.
// h = h0 * e^{+i w t} + h0[index_neg] * e^{-i w t}   (:55-62; no conjugate, quirk Q2)
__device__ __forceinline__ c32 propagate_height(c32 h0, c32 h0_neg, float omega, float time) {
    const float disp = omega * time;                               // :55 (one fp32 multiply, as the shader)
    // cos/sin of the fp32 phase, full range (phases reach 1e4..1e5 rad): reduce to revolutions in
    // [-0.5, 0.5] with a two-constant fp32 product (1/(2 pi) = HI + LO, both FMAs exact in the
    // product: fraction error < 6e-8 revolutions for |disp| < 1e5), then the hardware sin/cos.
    // Total abs error <= 4e-7 against the correctly rounded value the oracle uses.
    constexpr float INV_2PI_HI = 0.15915494f;                      // fl32(1 / (2 pi)) = 0x3E22F983
    constexpr float INV_2PI_LO = 6.4206382e-09f;                   // 1 / (2 pi) - INV_2PI_HI
    const float turns = rintf(disp * INV_2PI_HI);
    const float frac = fmaf(disp, INV_2PI_LO, fmaf(disp, INV_2PI_HI, -turns));
    const float s = sin_rev(frac), c = cos_rev(frac);
    // h0 (c + i s) + h0_neg (c - i s) = c (h0 + h0_neg) + s * i (h0 - h0_neg): five packed instructions
    const c32 p = h0 + h0_neg, q = h0 - h0_neg;
    return vfma(yx(q) * s, mk(-1.0f, 1.0f), p * c);
}
// k_norm = k / length(k) if length(k) > 1e-10 else 0   (:64-67)
__device__ __forceinline__ void k_normalised(float kx, float ky, float& knx, float& kny) {
    const float len = sqrtf(kx * kx + ky * ky);
    knx = 0.0f; kny = 0.0f;
    if (len > 1.0e-10f) { knx = kx / len; kny = ky / len; }
}
// Same value to <= 2 ulp with one v_rsq_f32 instead of a square root and two IEEE divisions
// (used by the fused kernel, where pass 1 is VALU-limited): kn = k * rsqrt(|k|^2).
__device__ __forceinline__ void k_normalised_fast(float kx, float ky, float& knx, float& kny) {
    const float l2 = kx * kx + ky * ky;
    const float r = (l2 > 1.0e-20f) ? rsqrtf(l2) : 0.0f;          // length(k) > 1e-10
    knx = kx * r;
    kny = ky * r;
}
// complex_mul(vec2(0, -kn), h) = (kn*h.y, -kn*h.x)   (:70-71)
__device__ __forceinline__ c32 mul_minus_i_kn(float kn, c32 h) { return yx(h) * mk(kn, -kn); }

// LDS pitch of one line buffer: padded line + 4 elements so that P adjacent lines do not alias.
template <int N> struct LinePitch { static constexpr int elems = LdsLine<N>::elems + 4; };

// ---------------------------------------------------------------------------------------------
// Fused frame
// ---------------------------------------------------------------------------------------------
// Intermediate (between the two passes): 4 x 4 chunks of c32 (128 bytes).  Chunk (X, Y) holds columns
// x = 4 X + c and rows y = 4 Y + r at offset 4 r + c.  Field order: 0 = disp_x, 1 = height, 2 = disp_z (OCEAN_FIELD_*).
struct InterLayout {
    size_t sx, sy, fs;
    int bshift;
    // One tile sharded over several GPUs (ocean_tile_pass1 / ocean_tile_pass2, SHARD kernels only): the receive buffer
    // of the all-to-all holds the chunk columns of source rank s in its own slab, chunk column X = s * 2^xs_shift + Xl at
    //     s * src_stride + (the layout above with X = Xl).   Unsharded: xs_shift = 31 (one slab).
    // With the exchange cut into 2^part_bits parts per rank (one all-to-all per part, overlapped with the next part's
    // pass 1), block v = X >> xs_shift of the half spectrum is part v % parts of source rank v / parts and sits in slab
    // part * world + rank:  slab = ((v & (parts - 1)) << rank_bits) | (v >> part_bits).
    int xs_shift = 31;
    size_t src_stride = 0;
    int part_bits = 0, rank_bits = 0;
};
__device__ __forceinline__ size_t tile_slab_offset(const InterLayout& lay, int Xc) {
    const int v = Xc >> lay.xs_shift;
    const int slab = ((v & ((1 << lay.part_bits) - 1)) << lay.rank_bits) | (v >> lay.part_bits);
    return (size_t)slab * lay.src_stride + (size_t)((uint32_t)Xc & ((1u << lay.xs_shift) - 1u)) * lay.sx;
}
// Rows of chunks are grouped in blocks of B = 2^bshift: chunk (X, Y) sits at
//     field * fs + (Y / B) * sy + X * sx + (Y % B) * 16        (elements; 16 = one 128-byte chunk).
// B = 1 is pass-2-contiguous (the chunks of one chunk row adjacent, sx = 16: pass 2 streams 64 KiB runs while pass 1
// scatters single chunks 64 KiB apart); B = N / 4 is pass-1-contiguous (all the chunks of one chunk column adjacent);
// in between a pass-1 workgroup writes B * 128 contiguous bytes and a pass-2 workgroup finds its row's lines B * 128
// bytes apart.  Shipped: B = 4 at N >= 2048, B = 1 below (Geo::inter_bshift, measurements there and in DESIGN 4.4).
// The staged hand-off always uses B = 1.
__device__ __forceinline__ size_t chunk_row_offset(const InterLayout& lay, int Y) {
    return (size_t)(Y >> lay.bshift) * lay.sy + (size_t)(Y & ((1 << lay.bshift) - 1)) * (size_t)16;   // one chunk = 16 elements
}
// K time steps of one tile in ONE launch pair (ocean_frame_batch; the latency-bound sizes N <= 1024, where a frame's two
// launches fill an eighth of the chip and the reference itself keeps 3 frames in flight, src/lib.rs:86,150): gridDim.y = K
// and frame y = blockIdx.y of the batch runs at time t0 + y dt with its own intermediate, Nyquist scratch and output map.
// The kernels of N > 1024 ignore these (their launches are never batched: one frame fills the chip).
struct FrameBatch {
    float dt = 0.0f;                // frame y: time = t0 + dt * (float)y, rounded like the host's `t0 + dt * (float)y` (no FMA)
    uint32_t inter_stride = 0;      // elements between the intermediates of consecutive frames
    size_t out_stride = 0;          // texels (float4) between their output maps
    // K TILES instead of K time steps (ocean_frame_tiles: dt = 0): frame y reads its own static inputs
    size_t spec_stride_bytes = 0;   // between the transposed spectra of consecutive tiles
    uint32_t omega_stride = 0;      // elements between their transposed dispersion arrays
};
template <int N> constexpr bool batched_launches = (N <= 1024);

// A chunk is 4 columns x 4 rows of complex = 128 bytes.  A pass-1 workgroup owns P = 4 lines (whole
// chunks, non-temporal stores) or P = 2 lines (the left or right 16 bytes of every chunk row; its
// neighbour, dispatched in the adjacent slot of the same XCD, writes the other half and the XCD L2
// merges them into full lines -- plain stores, so that the lines stay in L2 until complete).
// (2 x 8 chunks, so that a 2-line workgroup owns whole lines, cost pass 2 more than they gave pass 1: DESIGN 4.4.)
constexpr int CHUNK_W = 4, CHUNK_R = 4;

// ---------------------------------------------------------------------------------------------
// Fused frame, half-spectrum variant ("real-output" algorithm)
// ---------------------------------------------------------------------------------------------
// correction.comp:31 keeps only the real part of each inverse transform, and for any complex F
//     Re(IDFT2(F)) = IDFT2(S(F)),   S(F)[ky][kx] = (F[ky][kx] + conj(F[(-ky)%N][(-kx)%N])) / 2.
// S(F) is Hermitian, so after the transform along y the columns kx and N-kx are conjugates: pass 1
// only has to produce columns kx = 0 .. N/2-1 (the self-paired Nyquist column N/2, real after the
// transform, rides in the imaginary part of column 0), i.e. HALF the column FFTs and 12 instead of
// 24 bytes/texel of intermediate.
// Pass 2 rebuilds the full rows in LDS (C[kx] = A + iB, C[N-kx] = conj(A) + i conj(B)) and gets
// two real channels per complex FFT: (disp_x, disp_z) from one, height from the other.
// Same result as the three complex transforms up to fp32 rounding; the factors 1/2 are applied
// once in the epilogue.  HBM traffic per texel: pass 1 14 R + 12 W, pass 2 12 R + 16 W.
//
// Per output column x < N/2 and row y the symmetrised spectrum needs the propagated height at
// (y, x) and at ((-y)%N, (-x)%N):   H1 from h0T[x][y], h0T[N-1-x][N-1-y], omegaT[x][y];
//                                    H2 from h0T[x2][y2], h0T[(x-1)%N][(y-1)%N], omegaT[x2][y2],
// x2 = (N-x)%N, y2 = (N-y)%N  (the second mirror index is N-1-x2 = (x-1)%N).  With A = H1,
// B = conj(H2) and k1 = k_norm(x, y), k2 = k_norm(x2, y2)   (all quirks Q1/Q2 kept):
//     2 S(H)  = A + B,     2 S(Dx) = i (k2.x B - k1.x A),     2 S(Dz) = i (k2.y B - k1.y A).

// One field's symmetrised column spectrum (times 2) at row y of a column pair; kxv = (k(x), k((N-x)%N)).
// 2 S(H) = A + B,  2 S(Dx) = i (k2.x B - k1.x A),  2 S(Dz) = i (k2.y B - k1.y A),  k1 = k_norm(x, y),
// k2 = k_norm((N-x)%N, (N-y)%N) with the Q1 wrap and k * rsqrt(|k|^2) (k_normalised_fast), evaluated for
// the pair (k1, k2) in the two lanes of packed instructions.
template <int N>
__device__ __forceinline__ c32 half_spectrum_at(int f, c32 A, c32 B, c32 kxv, float kscale, int y) {
    if (f == 1) return A + B;
    const int y2 = (N - y) & (N - 1);
    const c32 kyv = mk(wave_index_q1((uint32_t)y, N), wave_index_q1((uint32_t)y2, N)) * kscale;
    const c32 l2 = vfma(kxv, kxv, kyv * kyv);
    const c32 r = mk((l2.x > 1.0e-20f) ? rsqrtf(l2.x) : 0.0f, (l2.y > 1.0e-20f) ? rsqrtf(l2.y) : 0.0f);
    const c32 kn = ((f == 0) ? kxv : kyv) * r;                     // (k1, k2) of this field
    const c32 v = vfma(A, -xx(kn), B * yy(kn));                    // k2 B - k1 A
    return crot(v);                                                // i * v
}
// ... for the E positions of a thread.
// (S, par): the thread's positions are y = S * (j + e * T) + par, T = N / (E * S) -- S = 1 for a whole line,
// S = 2 for one parity of a line split in two half-length transforms (k_half_pass1_split).
template <int N, int E, int S = 1>
__device__ __forceinline__ void half_spectrum(int f, const c32 (&A)[E], const c32 (&B)[E], c32 kxv, float kscale, int j,
                                              c32 (&reg)[E], int par = 0) {
    constexpr int T = N / (E * S);
#pragma unroll
    for (int e = 0; e < E; ++e) reg[e] = half_spectrum_at<N>(f, A[e], B[e], kxv, kscale, S * (j + e * T) + par);
}

// The same three spectra with the wave-vector normalisation applied ONCE (round 4; the kernels that transform the three fields
// one after the other): height first (2 S(H) = A + B from the unscaled values), then A <- A / |k1|, B <- B / |k2| in place
// (half_scale_AB: the one place the two quarter-rate v_rsq_f32 per element are paid), after which
//     2 S(Dx) = i (k2.x B' - k1.x A'),   2 S(Dz) = i (k2.y B' - k1.y A')
// are three packed instructions per element (k.x is the column's, wave-uniform; k.y a conversion and a multiply).  The
// transform phases of pass 1 are VALU-issue-bound and the two normalisations were 2 x 2.3 of their 22 us per round at
// N = 4096: frame 0.186 -> 0.181-0.182 ms, 5366-5374 -> 5488-5538 frames/s (r04_run18, A/B on one box); and the kernel drops
// from 128 VGPRs with 6 spilled to 124 with none.  (A / |k| * k.x instead of A * (k.x / |k|): one rounding moved, 1 ulp.)
template <int N>
__device__ __forceinline__ c32 wave_vector_y(int y, float kscale) {
    const int y2 = (N - y) & (N - 1);
    return mk(wave_index_q1((uint32_t)y, N), wave_index_q1((uint32_t)y2, N)) * kscale;
}
template <int N, int E, int S = 1>
__device__ __forceinline__ void half_scale_AB(c32 (&A)[E], c32 (&B)[E], c32 kxv, float kscale, int j, int par = 0) {
    constexpr int T = N / (E * S);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const c32 kyv = wave_vector_y<N>(S * (j + e * T) + par, kscale);
        const c32 l2 = vfma(kxv, kxv, kyv * kyv);
        const c32 r = mk((l2.x > 1.0e-20f) ? rsqrtf(l2.x) : 0.0f, (l2.y > 1.0e-20f) ? rsqrtf(l2.y) : 0.0f);
        A[e] = A[e] * xx(r);
        B[e] = B[e] * yy(r);
    }
}
template <int N, int E, int S = 1>
__device__ __forceinline__ void half_spectrum_scaled(int f, const c32 (&A)[E], const c32 (&B)[E], c32 kxv, float kscale, int j,
                                                     c32 (&reg)[E], int par = 0) {
    constexpr int T = N / (E * S);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const c32 kv = (f == 0) ? kxv : wave_vector_y<N>(S * (j + e * T) + par, kscale);
        reg[e] = crot(vfma(A[e], -xx(kv), B[e] * yy(kv)));          // i (k2 B' - k1 A')
    }
}

// The initial spectrum in HBM: fp32 complex (8 B/texel), or -- BASELINE config 5 -- two fp16 with
// a power-of-two scale (4 B/texel); arithmetic is fp32 either way.
template <bool H16> struct Spec;
template <> struct Spec<false> {
    typedef c32 elem;
    static __device__ __forceinline__ c32 load(const elem* p, float) { return *p; }
};
template <> struct Spec<true> {
    typedef uint32_t elem;
    static __device__ __forceinline__ c32 load(const elem* p, float descale) { return unpack_half2(*p, descale); }
};

// A = H(y, x), B = conj(H((-y)%N, (-x)%N)) for the E positions y = j + e T of a thread on column x, straight into
// registers: the loader of the latency-bound sizes (N <= 1024; Geo::dma selects half_load_AB_dma above that), where the
// working set is cache-resident, a thread has 8 elements and all 6 x 8 loads are issued at once (four dependent batches
// were 1.1 us of a 3.7 us workgroup at N = 512, timeline r03).  Every address is written as (uniform base + e-dependent
// constant)[small lane index], so that the six streams share three lane offsets and the bases stay in SGPRs; only
// e == 0 can hit the wrap of y2 = (N - y) % N and ym = (y - 1) % N (at y == 0).
template <int N, int E, bool H16>
__device__ __forceinline__ void half_load_AB(const void* __restrict__ h0T_, float descale, const float* __restrict__ omegaT,
                                             uint32_t x, int j, float time, c32 (&A)[E], c32 (&B)[E]) {
    typedef typename Spec<H16>::elem Sp;
    constexpr int T = N / E;
    const Sp* h0T = reinterpret_cast<const Sp*>(h0T_);
    const uint32_t x2 = (N - x) & (N - 1);
    const uint32_t xm = (x - 1u) & (N - 1);
    const Sp* own = h0T + (size_t)x * N;
    const Sp* mir = h0T + (size_t)(N - 1 - x) * N;
    const Sp* own2 = h0T + (size_t)x2 * N;
    const Sp* mir2 = h0T + (size_t)xm * N;
    const float* om = omegaT + (size_t)x * N;
    const float* om2 = omegaT + (size_t)x2 * N;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const c32 a = Spec<H16>::load((own + e * T) + j, descale);
        const c32 m = Spec<H16>::load((mir + (N - (e + 1) * T)) + (T - 1 - j), descale);   // mir[N - 1 - y]
        const float w = ((om + e * T))[j];
        c32 a2, m2;
        float w2;
        if (e == 0) {
            const int y2 = (N - j) & (N - 1);
            const int ym = (j - 1) & (N - 1);
            a2 = Spec<H16>::load(own2 + y2, descale);
            m2 = Spec<H16>::load(mir2 + ym, descale);
            w2 = om2[y2];
        } else {
            a2 = Spec<H16>::load((own2 + (N - (e + 1) * T)) + (T - j), descale);           // own2[N - y]
            m2 = Spec<H16>::load((mir2 + (e * T - 1)) + j, descale);                       // mir2[y - 1]
            w2 = ((om2 + (N - (e + 1) * T)))[T - j];
        }
        A[e] = propagate_height(a, m, w, time);
        B[e] = cconj(propagate_height(a2, m2, w2, time));
    }
}

// ---------------------------------------------------------------------------------------------
// Pass-1 loader through LDS-DMA (half_load_AB_dma)
// ---------------------------------------------------------------------------------------------
// half_load_AB stages its inputs in VGPRs: four dependent batches of 6 streams x 4 elements, at the 128-register budget
// of a 1024-thread workgroup, while the 136 KiB of line buffers sit idle until the first transform.  Here the workgroup
// streams its inputs through that idle LDS instead, in E / Q pieces of L = S * TS * Q positions of every line, with
// global_load_lds_dwordx4 (glds16: no VGPR destination, 1 KiB per wave instruction), a ring of D pieces so that
// D - 2 .. D - 1 pieces are in flight while one is propagated out of LDS:
//   * P + 1 "own-type" lines  OWN[k] = h0T[(x0 - 1 + k) % N][y0 .. y0 + L)            k = 0 is the left neighbour's column
//   * P + 1 "mirror-type"     MIR[k] = h0T[(N - x0 - k) % N][N - y0 - L .. N - y0)
//   * P dispersion lines      OM[c]  = omegaT[x0 + c][y0 .. y0 + L),   OM2[c] = omegaT[(N - x0 - c) % N][N - y0 - L .. N - y0)
// Column c of the workgroup (x = x0 + c) at position y = y0 + o reads (half_load_AB's six streams):
//     own[y] = OWN[c+1][o]      mirror[N-1-y] = MIR[c+1][L-1-o]     omega  = OM[c][o]
//     mirror2[y-1] = OWN[c][o-1]   own2[N-y]  = MIR[c][L-o]          omega2 = OM2[c][L-o]
// -- the duplicate streams (own2 / mirror2 of column c are mirror / own of column c - 1) come out of the SAME staged
// lines instead of a second request, so a workgroup asks the memory system for each of its P + 1 + P + 1 lines once
// (half_load_AB: 4 P requests, of which the L2 merged most but not all: 267 MB fetched at N = 4096 where 235 MB are
// the workgroups' distinct lines).  Position o = 0 needs index -1 / L of the second row: the previous
// piece's last (first) element -- every thread keeps those three values of its column from the piece before
// (broadcast reads); for the first piece they are the line's other end, which arrives with the LAST piece: that one
// element of B is computed after the loop.
// Synchronisation per piece: every wave waits for ITS instructions of piece b (counted vmcnt, the younger pieces stay in
// flight), one barrier -- which also says that everybody has finished reading piece b - 1 -- then piece b + D - 1 is
// issued into the slot of piece b - 1 and piece b is consumed.  No register staging, no compiler-visible global load
// in the whole load phase.
template <int N, int E, int P, bool H16, int S>
struct DmaRing {
    static constexpr int TS = N / (S * E);                         // threads per (sub-)line
    static constexpr int Q = (S * TS >= 256) ? 1 : 256 / (S * TS); // elements per thread and piece (a dispersion segment is >= one 1 KiB instruction)
    static constexpr int L = S * TS * Q;                           // positions of a line per piece
    static constexpr int NP = E / Q;                               // pieces
    static constexpr int SPB = H16 ? 4 : 8;                        // bytes per spectrum texel
    static constexpr int SEG_H = L * SPB, SEG_W = L * 4;           // one line segment: spectrum, dispersion
    static constexpr int OFF_MIR = 0;
    static constexpr int OFF_OWN = (P + 1) * SEG_H;
    static constexpr int OFF_OM2 = 2 * (P + 1) * SEG_H;
    static constexpr int OFF_OM = OFF_OM2 + P * SEG_W;
    static constexpr int PIECE = OFF_OM + P * SEG_W;               // bytes of one piece
    static constexpr int SLOTS = PIECE / 1024;                     // wave instructions per piece
    static constexpr int WAVES = P * S * TS / 64;
    static constexpr int SPW = (SLOTS + WAVES - 1) / WAVES;        // ... per wave (the first SLOTS % WAVES waves when uneven)
    static constexpr int REM = SLOTS % WAVES;
    static constexpr int BASE = SLOTS / WAVES;
    // ring slots: 4 (2-3 pieces in flight while one is consumed; measured r04_run1 at N = 4096 / 8192: 3, 4, 5 slots and
    // 2 slots of double pieces within the noise of each other, so the depth is not what bounds the load phase)
    static constexpr int DFIT = (144 * 1024) / PIECE;              // (N = 16384: 40 KiB pieces -> 3 slots)
    static constexpr int DMAX = (DFIT < 4) ? DFIT : 4;
    static constexpr int D = (NP < DMAX) ? (NP < 2 ? 2 : NP) : DMAX;
    static constexpr int bytes = D * PIECE;
    static_assert(E % Q == 0 && SEG_H % 1024 == 0 && SEG_W % 1024 == 0 && (P * S * TS) % 64 == 0, "DMA ring geometry");
};

// LDS bytes of the ring (fp32 spectrum: the larger one), 0 when the loader is not used for this geometry.
template <bool ON, int N, int E, int P, int S> struct DmaRingBytes { static constexpr int value = 0; };
template <int N, int E, int P, int S> struct DmaRingBytes<true, N, E, P, S> { static constexpr int value = DmaRing<N, E, P, false, S>::bytes; };

template <int N, int E, int P, bool H16, int S>
__device__ __forceinline__ void half_load_AB_dma(const void* __restrict__ h0T_, float descale, const float* __restrict__ omegaT_,
                                                 uint32_t x0, int c, int par, int j, int tid, float time, unsigned char* ring,
                                                 c32 (&A)[E], c32 (&B)[E]) {
    typedef DmaRing<N, E, P, H16, S> R;
    typedef typename Spec<H16>::elem Sp;
    constexpr int TS = R::TS, Q = R::Q, L = R::L, NP = R::NP, D = R::D;
    const char* h0T = reinterpret_cast<const char*>(h0T_);
    const char* omT = reinterpret_cast<const char*>(omegaT_);
    const int wave = wave_uniform(tid >> 6);
    const uint32_t lane16 = (uint32_t)(tid & 63) * 16u;
    const uint32_t ring_lds = lds_address(ring);

    // This wave's slots (wave + k * WAVES): source of piece 0 and the step from piece to piece (scalar registers).
    const char* src0[R::SPW];
    int step[R::SPW];
    bool stream_once[R::SPW];
#pragma unroll
    for (int k = 0; k < R::SPW; ++k) {
        const int byte = (wave + k * R::WAVES) * 1024;             // offset within the piece = offset within its LDS image
        if (byte < R::OFF_OWN) {
            const uint32_t line = (uint32_t)(N - (int)x0 - byte / R::SEG_H) & (uint32_t)(N - 1);
            src0[k] = h0T + ((size_t)line * N + (N - L)) * R::SPB + (byte % R::SEG_H);
            step[k] = -R::SEG_H; stream_once[k] = false;
        } else if (byte < R::OFF_OM2) {
            const int bb = byte - R::OFF_OWN;
            const uint32_t line = (uint32_t)((int)x0 - 1 + bb / R::SEG_H) & (uint32_t)(N - 1);
            src0[k] = h0T + (size_t)line * N * R::SPB + (bb % R::SEG_H);
            step[k] = R::SEG_H; stream_once[k] = false;
        } else if (byte < R::OFF_OM) {
            const int bb = byte - R::OFF_OM2;
            const uint32_t line = (uint32_t)(N - (int)x0 - bb / R::SEG_W) & (uint32_t)(N - 1);
            src0[k] = omT + ((size_t)line * N + (N - L)) * 4 + (bb % R::SEG_W);
            step[k] = -R::SEG_W; stream_once[k] = true;
        } else {
            const int bb = byte - R::OFF_OM;
            src0[k] = omT + (size_t)(x0 + (uint32_t)(bb / R::SEG_W)) * N * 4 + (bb % R::SEG_W);
            step[k] = R::SEG_W; stream_once[k] = true;
        }
    }
    auto issue = [&](int b) {
#pragma unroll
        for (int k = 0; k < R::SPW; ++k) {
            if ((k + 1) * R::WAVES <= R::SLOTS || wave + k * R::WAVES < R::SLOTS) {   // wave-uniform
                const char* src = src0[k] + (ptrdiff_t)b * step[k];
                const uint32_t dst = ring_lds + (uint32_t)((b % D) * R::PIECE + (wave + k * R::WAVES) * 1024);
                // the dispersion lines are read once per frame by one workgroup: non-temporal where the working set exceeds the caches
                // (measured r04_run1, pass 1 at N = 4096: the hint on nothing +6 us, on everything +6 us; run 22 of round 1
                //  found the same for plain loads, and -2.5 % at 2048 where everything stays cache-resident)
                if (N >= 4096 && stream_once[k]) glds16<true>(src, lane16, dst);
                else glds16<false>(src, lane16, dst);
            }
        }
    };
    // "my instructions of piece b have landed": the pieces issued after it may stay in flight
    auto wait_piece = [&](int b, int issued_upto) {
        const int after = issued_upto - b;                         // compile-time after unrolling: 0 .. D - 2
        if constexpr (R::REM == 0) {
            if (after <= 0) dma_wait<0>(); else if (after == 1) dma_wait<R::BASE>(); else if (after == 2) dma_wait<2 * R::BASE>(); else dma_wait<3 * R::BASE>();
        } else if (wave < R::REM) {
            if (after <= 0) dma_wait<0>(); else if (after == 1) dma_wait<R::BASE + 1>(); else if (after == 2) dma_wait<2 * (R::BASE + 1)>(); else dma_wait<3 * (R::BASE + 1)>();
        } else {
            if (after <= 0) dma_wait<0>(); else if (after == 1) dma_wait<R::BASE>(); else if (after == 2) dma_wait<2 * R::BASE>(); else dma_wait<3 * R::BASE>();
        }
    };
    static_assert(D <= 5, "wait_piece covers rings of up to five pieces");

#pragma unroll
    for (int b = 0; b < D - 1 && b < NP; ++b) issue(b);

    const int o0 = S * j + par;                                    // this thread's offset within a piece (element q: + q S TS)
    const bool first = (o0 == 0);
    const Sp* own1 = reinterpret_cast<const Sp*>(ring + R::OFF_OWN + (c + 1) * R::SEG_H) + o0;        // OWN[c+1][o]
    const Sp* own0 = reinterpret_cast<const Sp*>(ring + R::OFF_OWN + c * R::SEG_H) + o0 - 1;          // OWN[c][o-1]
    const Sp* mir1 = reinterpret_cast<const Sp*>(ring + R::OFF_MIR + (c + 1) * R::SEG_H) + (L - 1 - o0);   // MIR[c+1][L-1-o]
    const Sp* mir0 = reinterpret_cast<const Sp*>(ring + R::OFF_MIR + c * R::SEG_H) + (L - o0);        // MIR[c][L-o]
    const float* om1 = reinterpret_cast<const float*>(ring + R::OFF_OM + c * R::SEG_W) + o0;          // OM[c][o]
    const float* om0 = reinterpret_cast<const float*>(ring + R::OFF_OM2 + c * R::SEG_W) + (L - o0);   // OM2[c][L-o]
    // the piece's last / first elements of the second rows: position o = 0 of the NEXT piece (wave-uniform addresses)
    const Sp* own0_last = reinterpret_cast<const Sp*>(ring + R::OFF_OWN + c * R::SEG_H) + (L - 1);
    const Sp* mir0_first = reinterpret_cast<const Sp*>(ring + R::OFF_MIR + c * R::SEG_H);
    const float* om0_first = reinterpret_cast<const float*>(ring + R::OFF_OM2 + c * R::SEG_W);
    c32 edge_m2 = mk(0.0f, 0.0f), edge_a2 = mk(0.0f, 0.0f);
    float edge_w2 = 0.0f;
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        constexpr int PE = R::PIECE / (int)sizeof(Sp), PF = R::PIECE / 4;
        const int issued = (b + D - 2 < NP - 1) ? (b + D - 2) : (NP - 1);
        wait_piece(b, issued);
        dma_barrier();
        if (b + D - 1 < NP) issue(b + D - 1);                      // into the slot of piece b - 1: everybody is past it
        const int slot = b % D;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = b * Q + q;
            const int oq = q * S * TS;
            const c32 a = Spec<H16>::load(own1 + slot * PE + oq, descale);
            const c32 m = Spec<H16>::load(mir1 + slot * PE - oq, descale);
            const float w = om1[slot * PF + oq];
            c32 m2 = Spec<H16>::load(own0 + slot * PE + oq, descale);
            c32 a2 = Spec<H16>::load(mir0 + slot * PE - oq, descale);
            float w2 = om0[slot * PF - oq];
            if (q == 0) {                                          // o == 0: the previous piece's values (b == 0: fixed below)
                m2 = first ? edge_m2 : m2;
                a2 = first ? edge_a2 : a2;
                w2 = first ? edge_w2 : w2;
            }
            A[e] = propagate_height(a, m, w, time);
            B[e] = cconj(propagate_height(a2, m2, w2, time));
            pin_here(A[e], B[e]);                                  // (else the arithmetic sinks to the first use, behind the whole
                                                                   //  load phase, and the raw inputs of every piece stay live: 117 spills)
        }
        edge_m2 = Spec<H16>::load(own0_last + slot * PE, descale);
        edge_a2 = Spec<H16>::load(mir0_first + slot * PE, descale);
        edge_w2 = om0_first[slot * PF];
    }
    // position y = 0 (piece 0, o = 0) pairs with the lines' other ends, which came with the last piece
    {
        const c32 b0 = cconj(propagate_height(edge_a2, edge_m2, edge_w2, time));
        B[0] = first ? b0 : B[0];
    }
}

// The same for ONE position (the Nyquist column is done element-wise by a whole workgroup).
template <int N, bool H16>
__device__ __forceinline__ void half_AB_at(const void* __restrict__ h0T_, float descale, const float* __restrict__ omegaT,
                                           uint32_t x, uint32_t y, float time, c32& A, c32& B) {
    typedef typename Spec<H16>::elem S;
    const S* h0T = reinterpret_cast<const S*>(h0T_);
    const uint32_t x2 = (N - x) & (N - 1), y2 = (N - y) & (N - 1);
    const uint32_t xm = (x - 1u) & (N - 1), ym = (y - 1u) & (N - 1);
    A = propagate_height(Spec<H16>::load(h0T + (size_t)x * N + y, descale),
                         Spec<H16>::load(h0T + (size_t)(N - 1 - x) * N + (N - 1 - y), descale), omegaT[(size_t)x * N + y], time);
    const c32 h2 = propagate_height(Spec<H16>::load(h0T + (size_t)x2 * N + y2, descale),
                                    Spec<H16>::load(h0T + (size_t)xm * N + ym, descale), omegaT[(size_t)x2 * N + y2], time);
    B = cconj(h2);
}

// The Nyquist column's three symmetrised spectra, element-wise by all THREADS threads of ONE workgroup, into the
// scratch `nyq_spec` (3 N complex), then published to the other waves of the workgroup.
template <int N, bool H16, int THREADS>
__device__ __forceinline__ void nyquist_spectra(const void* __restrict__ h0T, float descale, const float* __restrict__ omegaT,
                                                c32* nyq_spec, int tid, float time, float kscale) {
    const c32 kxn = xx(mk(wave_index_q1((uint32_t)(N / 2), N) * kscale, 0.0f));
#pragma unroll 1
    for (int y = tid; y < N; y += THREADS) {
        c32 An, Bn;
        half_AB_at<N, H16>(h0T, descale, omegaT, (uint32_t)(N / 2), (uint32_t)y, time, An, Bn);
#pragma unroll
        for (int f = 0; f < 3; ++f) nyq_spec[(size_t)f * N + y] = half_spectrum_at<N>(f, An, Bn, kxn, kscale, y);
    }
    workgroup_publish();
}

// grid = (N/2)/P blocks of P columns.  The half spectrum has N/2 + 1 distinct columns; the odd one out, the
// self-paired Nyquist column kx = N/2, is like column 0 Hermitian along y, so both have REAL column transforms
// and share one complex FFT: line 0 of the workgroup that owns column 0 transforms S0 + i Sn and the
// intermediate's column 0 holds (column 0, Nyquist column) as (re, im).  That workgroup first evaluates the
// Nyquist column's three symmetrised spectra element-wise with all its threads (3 N complex in `nyq_spec`,
// an L2-resident scratch) and line 0 adds them to its inputs field by field.
// Why not a workgroup of its own: at N = 4096 the 512 column groups are exactly two dispatch rounds of 256
// one-per-CU workgroups; a 513th ran alone after everything else (tools/timeline.hip: bulk done at 102 us,
// kernel end at 133 us; a lone workgroup is latency-bound and takes 40 us for a third of the work).
// __launch_bounds__(.., 4 waves/SIMD when the workgroup is >= 512 threads): 1024 threads per CU, i.e. one
// 4-line or two 2-line workgroups co-resident (128 VGPRs each).
// Waves per SIMD the register allocation must leave room for: with E = 16 a 512- or 1024-thread workgroup (or two of
// 512) fills the CU's 1024 thread slots at 128 VGPRs; with more elements per thread (A/B knob OCEAN_E1) the workgroup
// is smaller and each thread gets the registers of the threads that are not there.
template <int E> constexpr int pass1_waves_per_simd(int threads) {
    return (E == 16) ? ((threads >= 512) ? 4 : 1) : ((threads / 256) > 1 ? (threads / 256) : 1);
}
// FPAR ("field-parallel", the launch- and latency-bound sizes N <= 1024): three wave groups per workgroup, one per
// field -- each loads and propagates the workgroup's columns (the same cache-resident lines, three times), builds ITS
// field's symmetrised spectrum, transforms and stores it -- instead of one group doing the three fields one after the
// other.  At N = 512 a workgroup is two waves on a CU with four SIMDs and every instruction's latency is exposed
// (timeline r03_run4: load 1.0 us, then 1.45 + 1.0 + 1.3 us of transforms, 3 x 0.2 us of stores): the three
// transforms now run side by side.
// DMA: the inputs are streamed through the idle line buffers by LDS-DMA (half_load_AB_dma, N >= 2048) instead of loaded
// into registers (half_load_AB).
// SHR: the wave-vector normalisation applied once, to A and B in place (half_scale_AB; the kernels that transform the three
// fields one after the other).  The batched launches of a size whose single frame is field-parallel switch it off: a frame of a
// batch must equal ocean_frame's bit for bit, and the shared normalisation moves one rounding.
template <int N, int E, int P, bool H16, bool DMA = false, bool FPAR = false, bool SHR = !FPAR>
__global__ void __launch_bounds__((N / E) * P * (FPAR ? 3 : 1), pass1_waves_per_simd<E>((N / E) * P * (FPAR ? 3 : 1)))
k_half_pass1(const void* __restrict__ h0T, float descale, const float* __restrict__ omegaT, c32* __restrict__ inter,
             c32* nyq_spec, const c32* __restrict__ tw, InterLayout lay, float time, float domain_size, int x_group0, FrameBatch batch) {
    if constexpr (batched_launches<N>) {                           // frame blockIdx.y of a batch (ocean_frame_batch; 0 for a plain frame)
        const uint32_t fi = blockIdx.y;
        time = __fadd_rn(time, __fmul_rn(batch.dt, (float)fi));
        inter += (size_t)fi * batch.inter_stride;
        nyq_spec += (size_t)fi * (3 * N);
        h0T = reinterpret_cast<const char*>(h0T) + (size_t)fi * batch.spec_stride_bytes;
        omegaT += (size_t)fi * batch.omega_stride;
    }
    constexpr int T = N / E;
    constexpr int GT = T * P;                                      // threads of one field group (= the workgroup without FPAR)
    constexpr int H2 = P / 2;
    constexpr int CR = CHUNK_R, CW = CHUNK_W;                    // row y of a chunk is y % CR, column x % CW
    static_assert((2 * T) % CR == 0 && CW % P == 0, "chunk geometry");
    static_assert(!FPAR || (GT % 64 == 0 && !DMA), "a field group is whole waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int fg = FPAR ? wave_uniform(tid / GT) : 0;              // field group = the field this thread transforms
    const int gt = FPAR ? (tid - fg * GT) : tid;                   // thread within the group
    const int c = (T >= 64) ? wave_uniform(gt / T) : (gt / T);
    const int j = gt % T;
    c32* lds_grp = lds + fg * (P * LinePitch<N>::elems);           // the group's P line buffers
    c32* lds_line = lds_grp + c * LinePitch<N>::elems;
    const float kscale = OCEAN_PI_F / domain_size;

    // X: the workgroup's column group within this launch (= within the intermediate it writes); Xg: within the tile.
    // They differ only when the tile is sharded over several GPUs and this rank transforms the column groups
    // [x_group0, x_group0 + gridDim.x) (ocean_tile_pass1).
    const int X = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int Xg = X + x_group0;
    if (Xg == 0) nyquist_spectra<N, H16, GT * (FPAR ? 3 : 1)>(h0T, descale, omegaT, nyq_spec, tid, time, kscale);   // uniform branch
    const bool packs_nyquist = (Xg == 0) && (c == 0);              // line 0 of that workgroup: column 0 + i * Nyquist
    const uint32_t x = (uint32_t)(Xg * P + c);                     // kx in [0, N/2)
    const uint32_t x2 = (N - x) & (N - 1);
    c32 A[E], B[E];
    OCEAN_TL(0);
    // N = 2048 is ONE dispatch round of 512 workgroups, two per CU, that start together: the whole chip loads, then the whole
    // chip transforms.  The second workgroup of a CU -- known by its waves' slots on their SIMDs (HW_ID.WAVE_ID: a 512-thread
    // workgroup is two waves per SIMD, the first arrival gets slots 0 and 1), not by the dispatch order -- starts 3.4 us late
    // (s_sleep 127 = 8128 clocks), so that one loads while the other transforms.  Timing only: the same bits.  A launch with
    // at most one workgroup per CU (a part of a sharded tile) has no wave in a slot >= 2 and pays nothing; a wave that lands
    // elsewhere only delays its own workgroup.  Measured (r05_run8, one box, three interleaved repetitions): pass 1
    // 28.2-28.5 -> 27.3-27.6 us, 21.17-21.20k -> 21.60-21.77k frames/s; the same 3.4 us for the FIRST arrival instead:
    // 21.6-21.7k as well (what pays is the offset); 2 us: 21.1-21.65k.  A second box (r05_run9): pass 1 26.8-26.9 -> 25.4-25.6 us,
    // frame 45.9-46.4 -> 44.7 us (+3.3 %), with the normal field 61.5-61.7 -> 60.4 us.  [Round 4 had the offset by block index:
    // +3 %, not shipped because it rested on the dispatch order.]
    if constexpr (N == 2048 && DMA && P == 2 && !FPAR) {
        if (hw_wave_slot() >= 2u) wave_sleep_127();
    }
    if constexpr (DMA) {
        static_assert(DmaRing<N, E, P, H16, 1>::bytes <= 160 * 1024, "the DMA ring fits the LDS");
        half_load_AB_dma<N, E, P, H16, 1>(h0T, descale, omegaT, (uint32_t)(Xg * P), c, 0, j, tid, time, smem, A, B);
    } else half_load_AB<N, E, H16>(h0T, descale, omegaT, x, j, time, A, B);
    OCEAN_TL(1);
    const c32 kxv = mk(wave_index_q1(x, N), wave_index_q1(x2, N)) * kscale;

    constexpr int H2S = (H2 > 0) ? H2 : 1;                        // (P == 1 stores straight from registers, below)
    const int h = gt % H2S;
    const int i = gt / H2S;
    const c32* l0 = lds_grp + (2 * h) * LinePitch<N>::elems;
    const c32* l1 = lds_grp + (2 * h + 1) * LinePitch<N>::elems;
#pragma unroll
    for (int ff = 0; ff < (FPAR ? 1 : 3); ++ff) {
        // One field group per field (FPAR), or the three fields one after the other: height first, then A and B are
        // normalised in place and disp_x, disp_z cost three packed instructions per element (half_scale_AB).
        constexpr bool SHARED_R = SHR;
        static_assert(!(FPAR && SHR), "one field per wave group: nothing to share");
        const int f = FPAR ? fg : ((ff == 0) ? 1 : ((ff == 1) ? 0 : 2));
        c32 reg[E];
        const int jf = FPAR ? j : opaque_lane(j);                  // (one field per thread: nothing to keep apart, and the twiddle loads may move up)
        if constexpr (SHARED_R) {
            if (ff == 0) half_spectrum<N, E>(1, A, B, kxv, kscale, jf, reg);
            else {
                if (ff == 1) half_scale_AB<N, E>(A, B, kxv, kscale, jf);
                half_spectrum_scaled<N, E>(f, A, B, kxv, kscale, jf, reg);
            }
        } else half_spectrum<N, E>(f, A, B, kxv, kscale, jf, reg);
        if (packs_nyquist) {
            const c32* z = nyq_spec + (size_t)f * N + jf;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                reg[e] = cadd_i(reg[e], z[e * T]);                 // + i * Sn
            }
        }
        if (DMA || ff > 0) __syncthreads();                       // the line buffers' previous readers (loader ring / chunk stores) are done
        if constexpr (P == 1) {
            // One column per workgroup (the latency-bound sizes, where a column per CU is the whole grid): the transform
            // ends in registers, X[j + e T], and every lane stores its 8-byte elements straight into the chunks (a
            // quarter of a chunk row each; the four column workgroups of a chunk column run on one XCD and the
            // cache-resident intermediate merges there).
            fft_line<N, E, 1, true, FUSED_HWTW>(reg, jf, tw, lds_line);   // the threads of a line are consecutive lanes
            OCEAN_TL(2 + 2 * (FPAR ? f : ff));
            c32* dcol = inter + (size_t)f * lay.fs + (size_t)(X / CW) * lay.sx + (X % CW);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int y = jf + e * T;
                dcol[chunk_row_offset(lay, y / CR) + (y % CR) * CW] = reg[e];
            }
            OCEAN_TL(3 + 2 * (FPAR ? f : ff));
            continue;
        }
        fft_line_to_lds<N, E, 1, true, FUSED_HWTW>(reg, jf, tw, lds_line);
        OCEAN_TL(2 + 2 * (FPAR ? f : ff));
        // chunk row Y = i / CR + q * (2T / CR): the thread's part and the (wave-uniform, scalar) part of the address add
        // up because 2T / CR is a power of two > i / CR (no carry between them in chunk_row_offset)
        c32* dst = inter + (size_t)f * lay.fs + (size_t)((X * P) / CW) * lay.sx + chunk_row_offset(lay, i / CR) + (i % CR) * CW +
                   ((X * P) % CW) + 2 * h;
#pragma unroll
        for (int q0 = 0; q0 < E / 2; ++q0) {
            const int q = q0;
            const int y = i + q * (2 * T);
            const c32 v0 = l0[lds_pad(y)];
            const c32 v1 = l1[lds_pad(y)];
            float4* o = reinterpret_cast<float4*>(dst + chunk_row_offset(lay, q * ((2 * T) / CR)));
            if constexpr (P == CW) store_float4_nt(o, make_float4(v0.x, v0.y, v1.x, v1.y));
            else *o = make_float4(v0.x, v0.y, v1.x, v1.y);        // half a chunk row: must meet its other half in L2
        }
        OCEAN_TL(3 + 2 * (FPAR ? f : ff));
    }
}

// The same pass for lines too long for a three-pass plan (N = 8192 = 2 * 16^3): every column is transformed
// as TWO interleaved half-length lines (decimation in time: E = DFT_{N/2}(x[2m]), O = DFT_{N/2}(x[2m+1])), which
// have the geometry of the N/2 kernel (2P sub-lines of N/(2E) threads, three passes, same LDS), and the last
// radix-2 step X[k] = E[k] + W^k O[k], X[k + N/2] = E[k] - W^k O[k] (W = e^{+2 pi i / N}) is done by the
// threads that read the lines out of LDS for the chunk stores.  Replaces a leading radix-2 pass with its own
// LDS exchange and barriers (run 16: FFT phases 14 / 10 / 13 us per field at 8192 against 6 / 6 / 7.5 for the
// same amount of data at 4096).
// I16 (opt-in precision mode, SURVEY 8d "B_frame16"; ocean_set_intermediate): the intermediate is stored as int16 (re, im)
// pairs -- 4 bytes per element at the SAME element offsets -- with one power-of-two scale per wave store, i.e. per block
// of 64 rows x 2 columns (rows k .. k+63 and their partners k+N/2 ..), kept in a side array
//     inter_scale[(field * N/64 + row block) * N/4 + column pair].
// The scale is the wave maximum of |re|, |im| (DPP reduction) rounded up to a power of two: values are exact multiples of
// 2^(e-15).  Numerics (tools/inter16_numerics.py, N = 8192, fp16-quantised spectrum): 2.7-3.1e-5 normalised max against
// the unquantised result, tolerance 1e-4.
// The inputs always arrive through the LDS-DMA ring (half_load_AB_dma with S = 2: sub-line thread (par, j) reads positions
// y = 2 (j + e TS) + par out of the staged lines; rounds 1-3 loaded pairs into registers and exchanged parities and
// duplicate lines through LDS: 454 -> 420 us, r04_run1).
template <int N, int E, int P, bool H16, bool I16 = false>
__global__ void __launch_bounds__((N / E) * P, pass1_waves_per_simd<E>((N / E) * P))
k_half_pass1_split(const void* __restrict__ h0T, float descale, const float* __restrict__ omegaT, c32* __restrict__ inter,
                   c32* nyq_spec, const c32* __restrict__ tw, InterLayout lay, float time, float domain_size, int x_group0,
                   float* __restrict__ inter_scale) {
    constexpr int M = N / 2;                                       // sub-transform length
    constexpr int TS = M / E;                                      // threads per sub-line
    constexpr int THREADS = 2 * P * TS;
    constexpr int CR = CHUNK_R, CW = CHUNK_W;
    static_assert((P == 1 || P == 2) && CW % P == 0 && M % THREADS == 0 && !(I16 && P == 1), "split geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int l = (TS >= 64) ? wave_uniform(tid / TS) : (tid / TS);   // sub-line: column l / 2, parity l % 2
    const int c = l >> 1, par = l & 1;
    const int j = tid % TS;
    c32* lds_line = lds + l * LinePitch<M>::elems;
    const float kscale = OCEAN_PI_F / domain_size;

    const int X = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // within this launch; Xg within the tile (k_half_pass1)
    const int Xg = X + x_group0;
    if (Xg == 0) nyquist_spectra<N, H16, THREADS>(h0T, descale, omegaT, nyq_spec, tid, time, kscale);   // uniform branch
    const bool packs_nyquist = (Xg == 0) && (c == 0);              // both parities of column 0 carry the Nyquist column
    const uint32_t x = (uint32_t)(Xg * P + c);
    const uint32_t x2 = (N - x) & (N - 1);
    c32 A[E], B[E];
    OCEAN_TL(0);
    static_assert(DmaRing<N, E, P, H16, 2>::bytes <= 160 * 1024, "the DMA ring fits the LDS");
    half_load_AB_dma<N, E, P, H16, 2>(h0T, descale, omegaT, (uint32_t)(Xg * P), c, par, j, tid, time, smem, A, B);
    __syncthreads();                                               // the ring is the transforms' line buffers next
    OCEAN_TL(1);
    const c32 kxv = mk(wave_index_q1(x, N), wave_index_q1(x2, N)) * kscale;

#pragma unroll
    for (int ff = 0; ff < 3; ++ff) {
        const int f = (ff == 0) ? 1 : ((ff == 1) ? 0 : 2);          // height, then (A, B normalised in place) disp_x, disp_z
        c32 reg[E];
        const int jf = opaque_lane(j);
        if (ff == 0) half_spectrum<N, E, 2>(1, A, B, kxv, kscale, jf, reg, par);
        else {
            if (ff == 1) half_scale_AB<N, E, 2>(A, B, kxv, kscale, jf, par);
            half_spectrum_scaled<N, E, 2>(f, A, B, kxv, kscale, jf, reg, par);
        }
        if (packs_nyquist) {
            const c32* z = nyq_spec + (size_t)f * N + 2 * jf + par;
#pragma unroll
            for (int e = 0; e < E; ++e) reg[e] = cadd_i(reg[e], z[2 * e * TS]);   // + i * Sn
        }
        if (ff > 0) __syncthreads();
        const int tf = opaque_lane(tid);
        // The combine's twiddles e^{2 pi i k / N}, k = k0 + q ROWS (+ 1): ONE table read per field, in flight under the transform,
        // the rest by tw[1] and the constants e^{2 pi i q ROWS / N} = e^{2 pi i q / 8} (a read per q behind the transform's last
        // barrier: pass 1 401-407 -> 386-387 us at 8192, 1719-1916 -> 1643-1673 us at 16384, r04_run27).  Not with the 16-bit
        // intermediate, whose kernel has no two registers left (8 spilled).
        constexpr bool TWROT = !I16;
        float4 w2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        c32 w0 = mk(0.0f, 0.0f);                                   // e^{2 pi i k0 / N}; its neighbour k0 + 1 by tw[1] (a scalar)
        if constexpr (TWROT) w0 = tw[CR * (tf / (2 * P)) + 2 * (tf & 1)];
        // length N/2 inside an N-point context: twiddle stride 2
        fft_line_to_lds<M, E, 2, false, FUSED_HWTW>(reg, jf, tw, lds_line);
        OCEAN_TL(2 + 2 * ff);
        // The chunks of the split geometry are COLUMN-major inside (element (row r, column c) at c CR + r; SPLIT_CMAJOR):
        // a workgroup owns P < CW columns of every chunk, and its part of a chunk is then P * 32 contiguous bytes -- a thread
        // combines and stores two consecutive rows of one column (16 bytes), 2P adjacent lanes one piece -- instead of
        // CR pieces of P * 8 bytes 32 bytes apart.  Measured (r04_run25, one box, two repetitions): pass 1 at 16384
        // 1996-2001 -> 1766-1823 us (8-byte -> 32-byte pieces), at 8192 413-420 -> 414-428 us (16 -> 64: nothing); pass 2,
        // whose lanes now read 8 bytes 32 bytes apart inside the same 128-byte lines, +1 %.
        // Thread tf: chunk row g = tf / (2P) (+ the wave-uniform q THREADS / (2P) below: no carry between them in
        // chunk_row_offset, see k_half_pass1), column cc = (tf / 2) % P, rows 2 rp, 2 rp + 1 of the chunk, rp = tf % 2.
        const int rp = tf & 1, cc = (tf >> 1) & (P - 1), g = tf / (2 * P);
        const c32* ev = lds + (2 * cc) * LinePitch<M>::elems;      // the column's even and odd sub-lines
        const c32* od = ev + LinePitch<M>::elems;
        c32* dst = inter + (size_t)f * lay.fs + (size_t)((X * P) / CW) * lay.sx + chunk_row_offset(lay, g) + (((X * P) % CW) + cc) * CR + 2 * rp;
        constexpr int ROWS = (2 * THREADS) / P;                    // rows of every column one iteration covers
#pragma unroll
        for (int q0 = 0; q0 < M / ROWS; ++q0) {
            const int q = q0;
            const int k = CR * g + 2 * rp + q * ROWS;              // rows k, k + 1 and k + M, k + M + 1
            if constexpr (!TWROT) w2 = *reinterpret_cast<const float4*>(tw + k);   // e^{2 pi i k / N}, e^{2 pi i (k + 1) / N}
            c32 wa = TWROT ? w0 : mk(w2.x, w2.y), wb = TWROT ? cmul(w0, tw[1]) : mk(w2.z, w2.w);
            if constexpr (TWROT) {
                static_assert(8 * ROWS == N, "the rotation constants are eighth roots");
                constexpr float h = 0.70710678118654752f;
                if (q == 1) { wa = vfma(yy(wa), mk(-h, h), xx(wa) * mk(h, h)); wb = vfma(yy(wb), mk(-h, h), xx(wb) * mk(h, h)); }        // e^{i pi/4}
                if (q == 2) { wa = crot(wa); wb = crot(wb); }                                                                     // i
                if (q == 3) { wa = vfma(yy(wa), mk(-h, -h), xx(wa) * mk(-h, h)); wb = vfma(yy(wb), mk(-h, -h), xx(wb) * mk(-h, h)); }   // e^{3 i pi/4}
            }
            const int pk = lds_pad(k);                             // k is even: k + 1 has the same k >> 4
            const c32 ta = cmul_r(od[pk], wa, crot(wa)), tb = cmul_r(od[pk + 1], wb, crot(wb));
            const c32 ua = ev[pk], ub = ev[pk + 1];
            const c32 loa = ua + ta, hia = ua - ta, lob = ub + tb, hib = ub - tb;
            float4* olo = reinterpret_cast<float4*>(dst + chunk_row_offset(lay, q * (ROWS / CR)));
            float4* ohi = reinterpret_cast<float4*>(dst + chunk_row_offset(lay, q * (ROWS / CR) + M / CR));
            if constexpr (I16) {
                // one scale for this wave's two store blocks (64 rows x 2 columns: rows k .. k+63 and k+M .., the same columns,
                // the same magnitudes); int16 pairs at the element offsets of the fp32 layout, 4 bytes per element
                const c32 a0 = mk(fabsf(loa.x), fabsf(loa.y)), a1 = mk(fabsf(lob.x), fabsf(lob.y));
                const c32 b0 = mk(fabsf(hia.x), fabsf(hia.y)), b1 = mk(fabsf(hib.x), fabsf(hib.y));
                const float m = wave_max_nonneg(fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y)), fmaxf(fmaxf(b0.x, b0.y), fmaxf(b1.x, b1.y))));
                float scale, inv;
                block_scale_i16(m, scale, inv);
                uint32_t* base32 = reinterpret_cast<uint32_t*>(inter);
                uint2* qlo = reinterpret_cast<uint2*>(base32 + (reinterpret_cast<c32*>(olo) - inter));
                uint2* qhi = reinterpret_cast<uint2*>(base32 + (reinterpret_cast<c32*>(ohi) - inter));
                *qlo = make_uint2(pack_i16x2(loa, inv), pack_i16x2(lob, inv));
                *qhi = make_uint2(pack_i16x2(hia, inv), pack_i16x2(hib, inv));
                if ((tf & 63) == 0) {                              // one lane per wave: the two blocks' entries
                    float* sc = inter_scale + ((size_t)f * (N / 64) + (size_t)(k >> 6)) * (N / 4) + Xg;
                    sc[0] = scale;
                    sc[(size_t)(M / 64) * (N / 4)] = scale;
                }
                continue;
            }
            *olo = make_float4(loa.x, loa.y, lob.x, lob.y);        // P of a chunk's CW columns: meets the others in L2 (plain
            *ohi = make_float4(hia.x, hia.y, hib.x, hib.y);        // stores; streamed 32-byte pieces: pass 1 x 2.3-3 at 16384)
        }
        OCEAN_TL(3 + 2 * ff);
    }
}

// LDS pitch of pass 2's line buffers: with two rows per workgroup the loader's 16-lane store groups alternate
// between the two lines (see below), and a pitch of 16 dwords mod 32 banks keeps them on disjoint banks.
template <int N, int R2> struct Pitch2 { static constexpr int elems = LdsLine<N>::elems + ((R2 == 2) ? 8 : 4); };

// Pass 2 of the half-spectrum path: R2 rows per workgroup (T threads each); two complex FFTs per row.
// Two thread mappings:
//   loader  (lr, lk): the rows of a workgroup lie in one chunk row (R2 divides CHUNK_R), so consecutive lanes read
//           a chunk's CHUNK_W columns of row 0, then of row 1, ...: R2 * CHUNK_W * 8 contiguous bytes per 128-byte
//           line and per load instruction (32 B at R2 = 1 with 4 x 4 chunks, 64 B at R2 = 2) -- fewer distinct
//           lines per wave instruction and fewer sharers of a line in L2.  R2 = 1: lk = tid, the FFT mapping.
//   FFT     (ll, j):  a wave stays inside one row; thread j holds x[j + e T].
// The two meet in LDS, where the full row is rebuilt from the half spectrum anyway.
// __launch_bounds__(.., 4 waves/SIMD): 1024 threads per CU (LDS: 4 x 35 KiB or 2 x 70 KiB), i.e. at most 128 VGPRs.
// GRP: this many consecutive chunk rows run in adjacent dispatch slots of one XCD (with the pass-1-contiguous layout the
// chunks (X, Y), (X, Y + 1), ... are adjacent in memory: the workgroups that read one DRAM page run together).
// PPAR (the latency-bound sizes N <= 1024): two wave groups per workgroup, one per transform -- group 0 gathers and
// transforms the height rows, group 1 the (disp_x, disp_z) rows, side by side instead of one after the other; group 0
// hands its real parts over through LDS and group 1 writes the RGBA rows.  (Timeline r03_run4 at N = 512: gather
// 0.9 + transform 0.9 us, then gather 0.4 + transform 1.65 us, one after the other, on a CU with two waves.)
// SHARD: this launch transforms the gridDim.x * R2 rows of ONE rank of a tile sharded over several GPUs; `inter` is the
// receive buffer of the all-to-all (InterLayout::xs_shift / src_stride), `out` the rank's block of rows.
// PLANE (the frame with the normal field, ocean_set_frame_normals): the epilogue also stores channel `plane_channel` of the
// finished map -- the very floats it puts into the RGBA texels -- as a dense fp32 plane, plane[y N + x], which is what the
// normal-field kernel differentiates (k_normals_plane: 4 instead of 16 bytes read per texel; ocean_staged_kernels.hpp).
template <int N, int E, int P1, int R2, int GRP = 1, bool PPAR = false, bool SHARD = false, bool PLANE = false>
__global__ void __launch_bounds__((N / E) * R2 * (PPAR ? 2 : 1), (E == 16) ? 4 : 2)
k_half_pass2(const c32* __restrict__ inter, float4* __restrict__ out, const c32* __restrict__ tw, InterLayout lay,
             float* __restrict__ plane, int plane_channel, FrameBatch batch) {
    if constexpr (batched_launches<N> && !SHARD) {                 // frame blockIdx.y of a batch (ocean_frame_batch)
        const uint32_t fi = blockIdx.y;
        inter += (size_t)fi * batch.inter_stride;
        out += (size_t)fi * batch.out_stride;
        if constexpr (PLANE) plane += (size_t)fi * ((size_t)N * N);    // (one plane per frame of the batch)
    }
    constexpr int T = N / E;
    constexpr int GT = T * R2;                                     // threads of one transform group (= the workgroup without PPAR)
    constexpr int EH = E / 2;                                      // elements of the half spectrum per thread
    static_assert(T % P1 == 0 && (E % 2) == 0, "geometry");
    static_assert(!PPAR || GT % 64 == 0, "a transform group is whole waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* lds = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int pg = PPAR ? wave_uniform(tid / GT) : 0;              // 0: height, 1: (disp_x, disp_z)
    const int gt = PPAR ? (tid - pg * GT) : tid;
    const int ll = (T >= 64) ? wave_uniform(gt / T) : (gt / T);
    const int j = gt % T;
    constexpr int CR = CHUNK_R;
    constexpr int LP = Pitch2<N, R2>::elems;
    static_assert(P1 == CHUNK_W, "pass 2 reads whole chunk rows");
    static_assert(CR % R2 == 0 || R2 % CR == 0, "the rows of a workgroup tile chunk rows");
    constexpr int S = ((CR > R2) ? (CR / R2) : 1) * GRP;           // workgroups per XCD group: sharers of a chunk row x GRP
    int rb = blockIdx.x;
    if (S > 1 && (gridDim.x % (8 * S)) == 0) {
        const int xcd = rb & 7, slot = rb >> 3;
        rb = ((slot / S) * 8 + xcd) * S + (slot % S);
    }
    const int y = rb * R2 + ll;                                    // the row this thread transforms and stores
    c32* lds_grp = lds + pg * (R2 * LP);                           // the group's R2 line buffers
    c32* lds_line = lds_grp + ll * LP;
    float* hbuf = reinterpret_cast<float*>(lds + 2 * R2 * LP);     // PPAR: R2 rows of N height values
    // loader coordinates
    constexpr int RL = (R2 < CR) ? R2 : CR;                        // rows of one chunk this workgroup owns
    const int lr = (R2 == 1) ? 0 : ((gt / P1) % RL + (gt / (T * RL)) * RL);   // row within the workgroup
    const int lk0 = (R2 == 1) ? gt : (((gt % (T * RL)) / (P1 * RL)) * P1 + gt % P1);   // kx of element e = 0
    const int ly = rb * R2 + lr;
    c32* load_line = lds_grp + lr * LP;

    float keep_h[E];
    OCEAN_TL(0);
    // N <= 2048: all three fields' loads are issued before the first transform (a workgroup has 8, then 16, 8-byte loads
    // per thread in flight otherwise).  Run 34: N = 2048 20.07-20.24k frames/s against 19.84-19.94k; at N = 4096 the same
    // costs 2-3 us (pass 2 91-94 us against 88-91): more lines in flight than the L2 keeps for the four sharers.
    constexpr bool PREFETCH = (N <= 2048) && !PPAR && !SHARD;
    constexpr bool PREFETCH_Z = true;
    c32 pre_x[EH], pre_z[EH];
    if constexpr (PREFETCH) {
        const size_t off0 = chunk_row_offset(lay, ly / CR) + (size_t)(lk0 / P1) * lay.sx + (ly % CR) * P1 + (lk0 % P1);
        const c32* sx_ = inter + off0;
        const c32* sz_ = inter + (size_t)2 * lay.fs + off0;
#pragma unroll
        for (int e = 0; e < EH; ++e) { pre_x[e] = sx_[(size_t)e * (T / P1) * lay.sx]; if constexpr (PREFETCH_Z) pre_z[e] = sz_[(size_t)e * (T / P1) * lay.sx]; }
    }
#pragma unroll
    for (int it = 0; it < (PPAR ? 1 : 2); ++it) {                  // pass 0: height, 1: (disp_x, disp_z)
        const int pass = PPAR ? pg : it;
        const int jf = PPAR ? j : opaque_lane(j);
        const int lk = (R2 == 1) ? jf : opaque_lane(lk0);
        const size_t off = chunk_row_offset(lay, ly / CR) + (size_t)(lk / P1) * lay.sx + (ly % CR) * P1 + (lk % P1);
        c32 a[EH], b[EH];
        if constexpr (SHARD) {
            // chunk column X of the tile sits in the slab of source rank X >> xs_shift
            const size_t offy = chunk_row_offset(lay, ly / CR) + (ly % CR) * P1 + (lk % P1);
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                const int Xc = lk / P1 + e * (T / P1);
                const size_t o = offy + tile_slab_offset(lay, Xc);
                if (pass == 0) { a[e] = inter[(size_t)1 * lay.fs + o]; b[e] = mk(0.0f, 0.0f); }
                else { a[e] = inter[o]; b[e] = inter[(size_t)2 * lay.fs + o]; }
            }
        } else if (pass == 0) {
            const c32* src = inter + (size_t)1 * lay.fs + off;
#pragma unroll
            for (int e = 0; e < EH; ++e) { a[e] = src[(size_t)e * (T / P1) * lay.sx]; b[e] = mk(0.0f, 0.0f); }
        } else if constexpr (PREFETCH) {
            const c32* sz_ = inter + (size_t)2 * lay.fs + off;
#pragma unroll
            for (int e = 0; e < EH; ++e) { a[e] = pre_x[e]; b[e] = PREFETCH_Z ? pre_z[e] : sz_[(size_t)e * (T / P1) * lay.sx]; }
        } else {
            const c32* sx_ = inter + off;
            const c32* sz_ = inter + (size_t)2 * lay.fs + off;
#pragma unroll
            for (int e = 0; e < EH; ++e) { a[e] = sx_[(size_t)e * (T / P1) * lay.sx]; b[e] = sz_[(size_t)e * (T / P1) * lay.sx]; }
        }
        if (it > 0) __syncthreads();                               // previous FFT's LDS reads done
        // rebuild the full row: C[k] = A + iB, C[N-k] = conj(A) + i conj(B).  Column 0 of the intermediate
        // holds two real columns, (kx = 0, kx = N/2) as (re, im): C[0] = re(A) + i re(B), C[N/2] = im(A) + i im(B)
        c32* lo = load_line + lds_pad(lk);                         // lds_pad(lk + e*T) = lds_pad(lk) + e*(T + T/16)
        c32* hi = load_line + lds_pad(N - lk);                     // lds_pad(N - lk - e*T) = lds_pad(N - lk) - e*(T + T/16)
#pragma unroll
        for (int e = 0; e < EH; ++e) {
            c32 ck, cm;
            if (pass == 0) {
                ck = a[e];
                cm = cconj(a[e]);
            } else {
                ck = cadd_i(a[e], b[e]);                           // A + i B
                cm = vfma(yx(b[e]), mk(1.0f, 1.0f), cconj(a[e]));   // conj(A) + i conj(B)
            }
            if (e == 0) {
                const bool dc = (lk == 0);                         // kx = 0; its mirror slot is the Nyquist bin
                if (dc) {
                    const c32 bb = (pass == 0) ? mk(0.0f, 0.0f) : b[e];
                    ck = mk(a[e].x, bb.x);
                    cm = mk(a[e].y, bb.y);
                }
                lo[0] = ck;
                (dc ? (load_line + lds_pad(N / 2)) : hi)[0] = cm;
            } else {
                lo[e * (T + T / 16)] = ck;
                hi[-e * (T + T / 16)] = cm;
            }
        }
        __syncthreads();
        OCEAN_TL(1 + 3 * pass);
        c32 reg[E];
        const c32* g = lds_line + lds_pad(jf);
#pragma unroll
        for (int e = 0; e < E; ++e) reg[e] = g[e * (T + T / 16)];
        __syncthreads();
        fft_line<N, E, 1, true, FUSED_HWTW>(reg, jf, tw, lds_line);   // (ll, j): the threads of a row are consecutive lanes
        OCEAN_TL(2 + 3 * pass);
        if constexpr (PPAR) {                                      // group 0 hands its row of heights to group 1
            if (pass == 0) {
                float* hrow = hbuf + ll * N + j;
#pragma unroll
                for (int e = 0; e < E; ++e) hrow[e * T] = reg[e].x;
            }
            __syncthreads();
            if (pass == 1) {
                const float* hrow = hbuf + ll * N + j;
#pragma unroll
                for (int e = 0; e < E; ++e) keep_h[e] = hrow[e * T];
            }
        }
        if (pass == 0) {
            if constexpr (!PPAR) {
#pragma unroll
                for (int e = 0; e < E; ++e) keep_h[e] = reg[e].x;
            }
        } else {
            float4* orow = out + (size_t)y * N;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int xo = j + e * T;
                const float s = (((xo + y) & 1) == 0) ? -0.5f : 0.5f;   // correction.comp:29 and the 1/2 of S(F)
                const c32 d = reg[e] * s;
                const float hs = keep_h[e] * s;
                store_float4_nt(orow + xo, make_float4(d.x, hs, d.y, 0.0f));
                // (plain stores: the plane is read back by the next kernel out of the caches -- as non-temporal stores k_normals_plane
                //  takes 16.8-17.5 instead of 12.8 us at N = 2048 and 78-83 instead of 52 at 4096, r05_run7)
                if constexpr (PLANE) plane[(size_t)y * N + xo] = (plane_channel == 0) ? d.x : ((plane_channel == 1) ? hs : d.y);
            }
            OCEAN_TL(6);
        }
    }
}

// Pass 2 with real-output rows.  Every field's row is real, so its N values are ONE complex transform of M = N/2 points
// (x[2m] + i x[2m+1] = sum_{k<M} Z[k] e^{2 pi i k m / M}) of
//     Z[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) e^{2 pi i k / N},      k < M,
// where X[0 .. M] is the half spectrum the intermediate holds (column 0 = the real pair X[0], X[M]).  Against
// k_half_pass2 (two N-point transforms per row: the height's wastes its imaginary half): three M-point transforms, 3/4
// of the butterflies, half the LDS per row (two rows share a CU at N = 16384, four at 8192) and half the threads per row;
// a thread keeps 2E reals of two fields until the third arrives, which is what the 128-VGPR budget allows.  Shipped at
// N >= 8192 (Launch<N>::REAL2 with the measurements; rows of 128 or 64 threads at N <= 4096 are slower than k_half_pass2).
//   thread j holds X[j + e T] (T = M / E): its partner M - j - e T is element E-1-e of thread T - j -- one LDS round
//   trip (write own, read own and mirrored back); e^{2 pi i (j + e T)/N} = tw[j] * e^{2 pi i e/(2E)}: one table entry per thread,
//   the rest compile-time constants;
//   the transform's last pass lands in LDS (fft_line_to_lds), from where thread t takes texels t + s T, s < 2E (the
//   float (t & 1) of element (t >> 1) + s T/2): a wave's store instruction writes one contiguous KiB of the row, as
//   the other pass-2 kernels do, instead of 16-byte pieces 32 bytes apart.
// The loads of the next field are in flight under the current transform (__syncthreads() waits for LDS only).
template <int E, int e>
__device__ __forceinline__ c32 real_row_input(c32 x, c32 p, c32 wji, c32 wjn) {
    const c32 ev = vfma(p, mk(1.0f, -1.0f), x);                    // X[k] + conj X[M-k]
    const c32 dv = vfma(p, mk(-1.0f, 1.0f), x);                    // X[k] - conj X[M-k]
    const c32 r = cmul_r(dv, wji, wjn);                            // i (..) e^{2 pi i j / N}
    if constexpr (e == 0) return ev + r;
    else {
        constexpr float c = W64<e * (32 / E)>::c, s = W64<e * (32 / E)>::s;   // e^{2 pi i e T / N} = e^{2 pi i e / (2E)}
        return vfma(yy(r), mk(-s, c), vfma(xx(r), mk(c, s), ev));
    }
}
template <int E, int... I>
__device__ __forceinline__ void real_row_inputs(const c32* lo, const c32* hi, int k1, bool dc, c32 wji, c32 wjn, c32 (&reg)[E],
                                                std::integer_sequence<int, I...>) {
    // own and mirrored element both come back from LDS: nothing but the transform's registers lives across the round trip
    // kx = 0: the slot holds the real pair (X[0], X[M]); its mirror index M is not in the buffer
    const c32 a0 = lo[0], p0 = hi[0];
    reg[0] = real_row_input<E, 0>(dc ? mk(a0.x, 0.0f) : a0, dc ? mk(a0.y, 0.0f) : p0, wji, wjn);
    ((I == 0 ? (void)0 : (void)(reg[I] = real_row_input<E, I>(lo[I * k1], hi[-I * k1], wji, wjn))), ...);
}
template <int N, int E, int P1, int GRP = 1, bool SHARD = false, bool I16 = false, int WPS = 4, bool PLANE = false>
__global__ void __launch_bounds__((N / 2) / E, WPS)
k_half_pass2_real(const c32* __restrict__ inter, float4* __restrict__ out, const c32* __restrict__ tw, InterLayout lay,
                  const float* __restrict__ inter_scale, float* __restrict__ plane, int plane_channel) {
    static_assert(!(SHARD && I16), "the 16-bit intermediate is not combined with the sharded tile");
    constexpr int M = N / 2;
    constexpr int T = M / E;                                       // threads per row
    constexpr int CR = CHUNK_R;
    constexpr int K1 = T + T / 16;                                 // lds_pad(i + T) - lds_pad(i)
    static_assert(P1 == CHUNK_W && T % P1 == 0 && T % 16 == 0 && 32 % E == 0, "geometry");
    // lds_pad(i + s T/2) - lds_pad(i) for the i < T/2 of the store mapping (T/2 a multiple of 16, or 8 with i < 8)
    auto k2 = [](int s) { return s * (T / 2) + ((s * (T / 2)) >> 4); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* line = reinterpret_cast<c32*>(smem);                      // LinePitch<M>::elems
    const int tid = threadIdx.x;
    constexpr int S = CR * GRP;                                    // workgroups sharing a chunk line (x GRP chunk rows): same XCD
    int rb = blockIdx.x;
    if ((gridDim.x % (8 * S)) == 0) {
        const int xcd = rb & 7, slot = rb >> 3;
        rb = ((slot / S) * 8 + xcd) * S + (slot % S);
    }
    const int y = rb;
    // Element offsets in 32 bits (the whole intermediate of N = 16384 is 3 * 2^27 elements of 8 bytes: byte offsets fit too),
    // so that an address is one VGPR on top of the scalar base.
    static_assert((uint64_t)3 * (N / 2) * N * sizeof(c32) + ((uint64_t)1 << 28) < ((uint64_t)1 << 32), "32-bit byte offsets");
    const uint32_t sx = (uint32_t)lay.sx, fs = (uint32_t)lay.fs;
    const uint32_t offy = (uint32_t)chunk_row_offset(lay, y / CR) + (uint32_t)((tid % P1) * CR + (y % CR));   // column-major chunks (k_half_pass1_split)
    constexpr uint32_t SF = (uint32_t)(N / 64) * (N / 4);          // I16: one field's scales
    // the E half-spectrum values kx = tid + e T of field f (I16: the raw int16 pair in .x, its scale in .y)
    auto issue = [&](int f, c32 (&raw)[E]) {
        const char* base = reinterpret_cast<const char*>(inter);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t Xc = (uint32_t)(tid / P1 + e * (T / P1));
            if constexpr (SHARD) {                                 // chunk column Xc of the tile sits in a source rank's slab: tile_slab_offset
                const uint32_t v = Xc >> lay.xs_shift;
                const uint32_t slab = ((v & ((1u << lay.part_bits) - 1u)) << lay.rank_bits) | (v >> lay.part_bits);
                const uint32_t o = (uint32_t)f * fs + offy + slab * (uint32_t)lay.src_stride + (Xc & ((1u << lay.xs_shift) - 1u)) * sx;
                raw[e] = *reinterpret_cast<const c32*>(base + (size_t)(o * 8u));
            } else if constexpr (I16) {
                const uint32_t o = (uint32_t)f * fs + offy + Xc * sx;
                const float* sc = inter_scale + ((uint32_t)f * SF + (uint32_t)(y >> 6) * (N / 4) + (uint32_t)(tid >> 1) + e * (T / 2));
                raw[e] = mk(*reinterpret_cast<const float*>(base + (size_t)(o * 4u)), sc[0]);
            } else {
                const uint32_t o = (uint32_t)f * fs + offy + Xc * sx;
                raw[e] = *reinterpret_cast<const c32*>(base + (size_t)(o * 8u));
            }
        }
    };
    float keep_h[2 * E], keep_x[2 * E];
    const c32 wj = tw[tid];                                        // e^{2 pi i j / N}: once, for the three fields
    const float sgn = (((tid + y) & 1) == 0) ? -0.5f : 0.5f;       // correction.comp:29 and the 1/2 of S(F); T is even
    float4* orow = out + (size_t)y * N;
    // one field: pair, transform, take the store mapping.  `after_pairing` issues the loads that replace raw's registers.
    auto field = [&](int it, const c32 (&raw)[E], auto&& after_pairing) {
        const int jf = opaque_lane(tid);
        if (it > 0) __syncthreads();                               // the previous field's store-mapped reads are done
        c32* lo = line + lds_pad(jf);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (I16) lo[e * K1] = unpack_i16x2(__builtin_bit_cast(uint32_t, raw[e].x), raw[e].y);
            else lo[e * K1] = raw[e];
        }
        line_sync<T>();
        c32 reg[E];
        real_row_inputs<E>(lo, line + lds_pad(M - jf), K1, jf == 0, crot(wj), -wj, reg, std::make_integer_sequence<int, E>{});
        line_sync<T>();                                            // the mirrored reads are done before the transform scatters
        after_pairing();
        if (it == 0) OCEAN_TL(1);
        fft_line_to_lds<M, E, 2, true, FUSED_HWTW>(reg, jf, tw, line);
        if (it == 2) OCEAN_TL(5);
        const float* src = reinterpret_cast<const float*>(line + lds_pad(jf >> 1)) + (jf & 1);
        if (it == 0) {
#pragma unroll
            for (int s = 0; s < 2 * E; ++s) keep_h[s] = src[2 * k2(s)];
        } else if (it == 1) {
#pragma unroll
            for (int s = 0; s < 2 * E; ++s) keep_x[s] = src[2 * k2(s)];
        } else {
#pragma unroll
            for (int s = 0; s < 2 * E; ++s) {
                const float vx = keep_x[s] * sgn, vh = keep_h[s] * sgn, vz = src[2 * k2(s)] * sgn;
                store_float4_nt(orow + jf + s * T, make_float4(vx, vh, vz, 0.0f));
                if constexpr (PLANE) plane[(size_t)y * N + jf + s * T] = (plane_channel == 0) ? vx : ((plane_channel == 1) ? vh : vz);   // (k_half_pass2)
            }
        }
    };
    OCEAN_TL(0);
    c32 raw_h[E], raw_x[E], raw_z[E];
    issue(1, raw_h);
    issue(0, raw_x);
    field(0, raw_h, [&] { issue(2, raw_z); });
    OCEAN_TL(2);
    field(1, raw_x, [] {});
    OCEAN_TL(4);
    field(2, raw_z, [] {});
    OCEAN_TL(6);
}

// ---------------------------------------------------------------------------------------------
// Launch geometry per resolution -- the single source for the API (ocean_api.hip) and for the
// host emulation harness (tests/hipemu).
// ---------------------------------------------------------------------------------------------
// THROUGHPUT = the geometry of a BATCHED launch (ocean_frame_batch, N <= 1024): K frames fill the chip, so what counts is work
// per frame, not the serial chain of one workgroup -- no field-parallel wave groups (which load and propagate every line three
// times).  Measured at N = 512 (r05_run3, A/B on one box, K = 8 / 16 time steps per launch pair): 256-259k / 322-324k frames/s
// with the single-frame geometry (P = 1, FPAR), 300-306k / 386-399k with two columns per workgroup and no FPAR; the single
// frame itself is slower that way (91k against 105k), so ocean_frame keeps its own.
template <int N, int PSEL = 0, bool THROUGHPUT = false> struct Geo {
    static constexpr int E = 16;                                   // elements per thread
    // Elements per thread of fused pass 1.  At N <= 1024 a frame is launch- and latency-bound (a 512-point line with 16
    // elements per thread is half a wave): 8 elements per thread double the waves that share a workgroup's serial chain
    // of loads and transforms (run r02_run12: N = 512 51.0k -> 67.1k frames/s, 256 63k -> 77k, 1024 38.9k -> 41.3k).
    // N = 2048 is ONE dispatch round of 512 co-resident workgroups, as long as its slowest one: with the LDS-DMA loader
    // (no staging registers: 8 waves per workgroup fit) 8 elements per thread take pass 1 from 29.7-31.0 to 28.5-28.9 us,
    // 19.6-20.4k -> 20.7-20.8k frames/s (r04_run9-11; round 3 with the register loader: +3 %).  4096 and up: 16 (a line
    // of 512 threads leaves two lines per workgroup; 32 and 64 elements: slower, EXPERIMENTS 4.4).
    static constexpr int E1 = (N <= 2048) ? 8 : 16;
    static constexpr int E1S = 16;                                 // split kernels: lines of N / 2 points
    static constexpr int T = N / E;                                // threads per line
    static constexpr int ROW_LPW = (256 / T) > 1 ? (256 / T) : 1;  // rows per workgroup (staged)
    static constexpr int COL_LPW = (N > 8192) ? 1 : ((N > 4096) ? 2 : ((256 / T) > 4 ? (256 / T) : 4));  // columns per workgroup (staged)
    // Lines per workgroup of fused pass 1.  4 lines = 1024 threads at N = 4096 (one workgroup per CU, whole 4 x 4 chunks);
    // 2 lines = 512 threads and 70 KiB LDS (two co-resident workgroups, each writing half of every chunk row).
    // PSEL = 0 = this default; the API picks per size (Launch<N>::default_psel).
    static constexpr int P = PSEL ? PSEL : ((N > 8192) ? 1 : ((N > 4096) ? 2 : 4));
    static constexpr int row_threads = T * ROW_LPW;
    static constexpr int col_threads = T * COL_LPW;
    // LDS-DMA loader of fused pass 1 (half_load_AB_dma): the inputs streamed through the idle line buffers, no register
    // staging, every line of the workgroup requested once -- N >= 2048; the latency-bound sizes load straight into
    // registers (half_load_AB).  Measured (r04_run1/2): pass 1 at N = 8192 454 -> 405-424 us, 4096 97.8 -> 93-95 us (fetched
    // 267 -> 216 MB), 2048 unchanged.  OCEAN_DMA_MIN_N exists for the CPU emulation, which builds with 256 to run the ring's
    // geometry at the sizes it can afford (tests/test_emu_kernels.py).
#ifndef OCEAN_DMA_MIN_N
#define OCEAN_DMA_MIN_N 2048
#endif
    static constexpr bool dma = (N >= OCEAN_DMA_MIN_N) && (((N / E1) * P) % 64 == 0);
    // Field-parallel pass 1 (k_half_pass1<.., FPAR>): three wave groups per workgroup, one per field, at N <= 512.
        static constexpr bool fpar = !THROUGHPUT && (N <= 512) && (((N / E1) * P) % 64 == 0) && (3 * (N / E1) * P <= 1024) && !dma;
    static constexpr int half_threads1 = (N / E1) * P * (fpar ? 3 : 1);   // fused pass 1
    static constexpr int split_threads1 = (N / E1S) * P;
    static constexpr int line_bytes = LinePitch<N>::elems * (int)sizeof(c32);
    static constexpr int row_lds = ROW_LPW * line_bytes;
    static constexpr int col_lds = COL_LPW * line_bytes;
    static constexpr int max_i(int a, int b) { return a > b ? a : b; }
    static constexpr int half_lds1 = max_i(P * line_bytes * (fpar ? 3 : 1), DmaRingBytes<dma, N, E1, P, 1>::value);
    static constexpr int row_grid = N / ROW_LPW;
    static constexpr int col_grid = N / COL_LPW;
    // Rows per workgroup of fused pass 2: 256 threads' worth at N >= 2048; at most two at the launch-bound sizes, where
    // 8 or 16 rows per workgroup leave most CUs without one (N = 512: 64 workgroups; run r02_run13: 67.0k -> 72-73k
    // frames/s at 512, 76.8k -> 80-85k at 256; more rows at N >= 4096: slower).
    static constexpr int R2 = (N <= 1024 && ROW_LPW > 2) ? ((T * 2 >= 64) ? 2 : (64 / T)) : ROW_LPW;   // >= one wave per transform group
    // Elements per thread of fused pass 2: 8 at N <= 512 (a 512-point row is then one wave instead of half of one, and
    // a workgroup is one row: twice the workgroups, four waves per CU).
    static constexpr int E2 = (N <= 512) ? 8 : E;
    static constexpr int T2 = N / E2;                              // threads per row of fused pass 2
    static constexpr int R2h = (E2 == E) ? R2 : ((T2 >= 64) ? 1 : (64 / T2));   // rows per workgroup (whole waves per transform group)
    // Transform-parallel pass 2 (k_half_pass2<.., PPAR>): two wave groups per workgroup, height and (disp_x, disp_z), at
    // the latency-bound sizes N <= 1024.
    static constexpr bool ppar = (N <= 1024) && ((T2 * R2h) % 64 == 0) && (2 * T2 * R2h <= 1024);
    // Intermediate layout of the fused frame (InterLayout): blocks of B = 2^inter_bshift chunk rows.
    // B = 1 is pass-2-contiguous (pass 2 streams, pass 1 scatters single 128-byte chunks), B = N / 4 pass-1-contiguous.
    // At N >= 2048, where pass 1 is bound by its scattered stores, B = 4: a pass-1 wave stores 512-byte pieces, and the
    // workgroups of pass 2 that read one block (4 chunk rows = 16 rows) plus the next one run in adjacent slots of
    // one XCD (p2_group = 8 chunk rows).  Measured (runs r02_run20-22, four repetitions per box): N = 4096 5440-5500
    // frames/s against 4905-5270 with B = 1 and 5170-5220 with B = N / 4 (pass 1 93-95 us against 101-115 and 92-98,
    // pass 2 88-91 us against 88-92 and 99-100); N = 8192 1047-1065 against 1034-1054; N = 2048 19.6-19.9k against
    // 19.3-19.5k (run 31; four lines per pass-1 workgroup there: 17.2k).
    static constexpr int inter_bshift = (N >= 2048) ? 2 : 0;
    static constexpr int p2_group = (inter_bshift > 0) ? 8 : 1;
    static constexpr int half_threads2 = T2 * R2h * (ppar ? 2 : 1);  // fused pass 2
    static constexpr int half_grid2 = N / R2h;
    static constexpr int half_lines2 = R2h * Pitch2<N, R2h>::elems * (int)sizeof(c32);
    static constexpr int half_lds2 = ppar ? (2 * half_lines2 + R2h * N * (int)sizeof(float)) : half_lines2;
    static constexpr int half_grid1 = (N / 2) / P;                 // column groups
    // staged path with the chunked hand-off (k_stage_rows / k_stage_cols): 4 lines per workgroup
    static constexpr bool stage_chunked = (N <= 4096);
    static constexpr int stage_threads = 4 * T;
    static constexpr int stage_lds = 4 * line_bytes;
    static constexpr int stage_grid = N / 4;
    // split geometry (lines as two interleaved N/2 transforms; k_half_pass1_split; pass 2 is k_half_pass2_real): N >= 8192 in the
    // product, any N >= 512 in the emulation
    static constexpr bool can_split = (P <= 2) && (N >= 512);
    static constexpr int split_lds1 = max_i(2 * P * LinePitch<N / 2>::elems * (int)sizeof(c32), DmaRingBytes<can_split, N, E1S, P, 2>::value);
    // real-output pass 2 (k_half_pass2_real): one row of N/2 complex points per workgroup
    static constexpr int real_threads2 = (N / 2) / E;
    static constexpr int real_lds2 = LinePitch<N / 2>::elems * (int)sizeof(c32);
    // One tile sharded over `world` GPUs (ocean_tile_pass1 / ocean_tile_pass2): rank r owns the half-spectrum columns
    // [r N/(2 world), ..) in pass 1 and the rows [r N/world, ..) in pass 2.  The all-to-all buffers are
    //     [dest or src][block of B chunk rows][field][chunk column][chunk row in block][16 elements]
    // -- the intermediate's layout with the peer outermost, so that every (src, dest) message is contiguous.
    // `parts` (a power of two): the rank's column block is transformed and shipped in that many pieces, one all-to-all
    // each, so that the exchange of piece k runs under pass 1 of piece k + 1 (the reference's barrier
    // between its row and column dispatches, src/render.rs:1181-1208, becomes `parts` overlapped collectives).
    static constexpr bool tile_supported(int world, int parts = 1) {
        return world >= 1 && (world & (world - 1)) == 0 && parts >= 1 && (parts & (parts - 1)) == 0 && (N / world) >= 32 &&
               (N / 8 / world / parts) >= 1 && ((N / 2 / world / parts) % P) == 0;
    }
    static InterLayout tile_layout(int world, int parts = 1) {
        const size_t gxp = (size_t)N / 8 / world / parts, gyl = (size_t)N / 4 / world;   // chunk columns per piece / chunk rows per rank
        int bs = inter_bshift;
        while (((size_t)1 << bs) > gyl) --bs;
        const size_t B = (size_t)1 << bs;
        InterLayout l{0, 0, 0, bs};
        l.sx = B * 16;
        l.fs = gxp * l.sx;                                         // field stride inside a block of chunk rows
        l.sy = 3 * l.fs;                                           // block of chunk rows
        l.xs_shift = 0;
        while (((size_t)1 << l.xs_shift) < gxp) ++l.xs_shift;
        l.src_stride = (gyl / B) * l.sy;                           // one (src, dest, part) message
        while ((1 << l.part_bits) < parts) ++l.part_bits;
        while ((1 << l.rank_bits) < world) ++l.rank_bits;
        return l;
    }
    static_assert(row_threads <= 1024 && col_threads <= 1024 && half_threads1 <= 1024, "workgroup too large");
    static_assert(col_lds <= 160 * 1024 && half_lds1 <= 160 * 1024, "LDS budget (gfx950: 160 KiB)");
};

}  // namespace ocean
