// fft_core.hpp -- register-resident mixed-radix Stockham line FFT for gfx950 (wave64).
//
// Computes, per line, the reference's transform (shader/fft_row.comp:25-63):
//     X[n] = sum_m x[m] * e^{+2 pi i m n / N}      (unnormalised inverse DFT, natural order)
// but not with the reference's radix-2 / one-butterfly-per-thread schedule: each thread keeps
// E elements of a line in VGPRs, does radix-R butterflies (R <= E) entirely in registers and
// only crosses threads through LDS between passes (2 exchanges for N = 4096 instead of the
// reference's 12 barrier-separated LDS round trips).  Twiddles come from one fp64-computed table
// lookup per thread per pass (w) and a two-level rotation (apply_twiddles), instead of
// the reference's cos/sin per butterfly (fft_row.comp:32-33).
//
// Index algebra (verified against numpy in oracle prototype, see DESIGN.md "FFT schedule"):
//   T = N/E threads per line; at the start of every pass thread j holds x[j + e*T], e in [0,E).
//   Pass with radix R and Ns = product of earlier radices; for u in [0, E/R):
//     jv = j + u*T;  k = jv % Ns;  inputs a_t = reg[u + t*E/R] * w^{t}, w = e^{+2 pi i k/(Ns*R)}
//     b = DFT_R(a);  b_s goes to position (jv/Ns)*Ns*R + k + s*Ns.
//   The last pass' positions are again j + e'*T, so global stores stay lane-contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <ocean_device_intrinsics.hpp>

// The exchanges of a line synchronise the threads OF THAT LINE: a workgroup barrier in general; a wave-level fence when
// the caller guarantees that the T <= 64 threads of a line are T consecutive lanes of one wave (template flag CONTIG of
// fft_line / fft_line_to_lds; line_sync<T>, ocean_device_intrinsics.hpp).  SYNC_T below is T with CONTIG, "many" without.

namespace ocean {

// c32 (ocean_device_intrinsics.hpp) is a packed (re, im) 2-vector; everything here is written in whole-vector
// operations so that each line below is ONE v_pk_* instruction (swizzles ride on op_sel, constants in SGPR pairs).
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return a + b; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return a - b; }
__device__ __forceinline__ c32 cadd_i(c32 a, c32 b) { return vfma(yx(b), mk(-1.0f, 1.0f), a); }   // a + i b
__device__ __forceinline__ c32 csub_i(c32 a, c32 b) { return vfma(yx(b), mk(1.0f, -1.0f), a); }   // a - i b
__device__ __forceinline__ c32 crot(c32 a) { return yx(a) * mk(-1.0f, 1.0f); }                    // i a
__device__ __forceinline__ c32 cconj(c32 a) { return a * mk(1.0f, -1.0f); }
// a * w given w and wr = i w: two instructions; without wr: three, no temporary twiddle
__device__ __forceinline__ c32 cmul_r(c32 a, c32 w, c32 wr) { return vfma(yy(a), wr, xx(a) * w); }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return vfma(yy(a) * yx(w), mk(-1.0f, 1.0f), xx(a) * w); }

// e^{+2 pi i s / 64}, s in [0,32): exact trivial entries, others rounded from fp64.
template <int S64> struct W64 {};
#define OCEAN_W64(S, C, SN) \
    template <> struct W64<S> { static constexpr float c = C; static constexpr float s = SN; };
OCEAN_W64(0, 1.0f, 0.0f)
OCEAN_W64(1, 0.9951847266721969f, 0.0980171403295606f)
OCEAN_W64(2, 0.9807852804032304f, 0.19509032201612825f)
OCEAN_W64(3, 0.9569403357322088f, 0.29028467725446233f)
OCEAN_W64(4, 0.9238795325112867f, 0.3826834323650898f)
OCEAN_W64(5, 0.881921264348355f, 0.47139673682599764f)
OCEAN_W64(6, 0.8314696123025452f, 0.5555702330196022f)
OCEAN_W64(7, 0.773010453362737f, 0.6343932841636455f)
OCEAN_W64(8, 0.7071067811865476f, 0.7071067811865475f)
OCEAN_W64(9, 0.6343932841636455f, 0.773010453362737f)
OCEAN_W64(10, 0.5555702330196023f, 0.8314696123025452f)
OCEAN_W64(11, 0.4713967368259978f, 0.8819212643483549f)
OCEAN_W64(12, 0.38268343236508984f, 0.9238795325112867f)
OCEAN_W64(13, 0.29028467725446233f, 0.9569403357322089f)
OCEAN_W64(14, 0.19509032201612833f, 0.9807852804032304f)
OCEAN_W64(15, 0.09801714032956077f, 0.9951847266721968f)
OCEAN_W64(16, 0.0f, 1.0f)
OCEAN_W64(17, -0.09801714032956065f, 0.9951847266721969f)
OCEAN_W64(18, -0.1950903220161282f, 0.9807852804032304f)
OCEAN_W64(19, -0.29028467725446216f, 0.9569403357322089f)
OCEAN_W64(20, -0.3826834323650897f, 0.9238795325112867f)
OCEAN_W64(21, -0.4713967368259977f, 0.881921264348355f)
OCEAN_W64(22, -0.555570233019602f, 0.8314696123025455f)
OCEAN_W64(23, -0.6343932841636454f, 0.7730104533627371f)
OCEAN_W64(24, -0.7071067811865475f, 0.7071067811865476f)
OCEAN_W64(25, -0.773010453362737f, 0.6343932841636455f)
OCEAN_W64(26, -0.8314696123025453f, 0.5555702330196022f)
OCEAN_W64(27, -0.8819212643483549f, 0.47139673682599786f)
OCEAN_W64(28, -0.9238795325112867f, 0.3826834323650899f)
OCEAN_W64(29, -0.9569403357322088f, 0.2902846772544624f)
OCEAN_W64(30, -0.9807852804032304f, 0.1950903220161286f)
OCEAN_W64(31, -0.9951847266721968f, 0.09801714032956083f)
#undef OCEAN_W64

// In-register R-point unnormalised inverse DFT, natural order in and out (DIT recursion).
template <int R> struct Dft;
template <> struct Dft<1> {
    static __device__ __forceinline__ void run(const c32 (&in)[1], c32 (&out)[1]) { out[0] = in[0]; }
};
template <> struct Dft<2> {
    static __device__ __forceinline__ void run(const c32 (&in)[2], c32 (&out)[2]) {
        out[0] = cadd(in[0], in[1]);
        out[1] = csub(in[0], in[1]);
    }
};
template <> struct Dft<4> {
    static __device__ __forceinline__ void run(const c32 (&in)[4], c32 (&out)[4]) {
        const c32 t0 = cadd(in[0], in[2]), t1 = csub(in[0], in[2]);
        const c32 t2 = cadd(in[1], in[3]), t3 = csub(in[1], in[3]);
        out[0] = cadd(t0, t2);
        out[2] = csub(t0, t2);
        out[1] = cadd_i(t1, t3);
        out[3] = csub_i(t1, t3);
    }
};
template <int R, int S> struct Combine {
    static __device__ __forceinline__ void run(const c32 (&ev)[R / 2], const c32 (&od)[R / 2], c32 (&out)[R]) {
        // out[S] = ev + od * W, out[S + R/2] = ev - od * W,  W = e^{+2 pi i S / R}
        static_assert(R <= 64, "twiddle constants exist up to a 64-point butterfly");
        constexpr int s64 = S * (64 / R);
        if constexpr (s64 == 0) {
            out[S] = cadd(ev[S], od[S]);
            out[S + R / 2] = csub(ev[S], od[S]);
        } else if constexpr (s64 == 16) {                                       // W = i
            out[S] = cadd_i(ev[S], od[S]);
            out[S + R / 2] = csub_i(ev[S], od[S]);
        } else {
            const c32 t = cmul_r(od[S], mk(W64<s64>::c, W64<s64>::s), mk(-W64<s64>::s, W64<s64>::c));
            out[S] = cadd(ev[S], t);
            out[S + R / 2] = csub(ev[S], t);
        }
        if constexpr (S + 1 < R / 2) Combine<R, S + 1>::run(ev, od, out);
    }
};
template <int R> struct Dft {
    static __device__ __forceinline__ void run(const c32 (&in)[R], c32 (&out)[R]) {
        c32 e[R / 2], o[R / 2], ev[R / 2], od[R / 2];
#pragma unroll
        for (int t = 0; t < R / 2; ++t) { e[t] = in[2 * t]; o[t] = in[2 * t + 1]; }
        Dft<R / 2>::run(e, ev);
        Dft<R / 2>::run(o, od);
        Combine<R, 0>::run(ev, od, out);
    }
};

// ---------------------------------------------------------------------------------------------
// Compile-time plan: radices = [N / E^q (if > 1), E, E, ...]   (small radix first: no twiddles)
// ---------------------------------------------------------------------------------------------
template <int N, int E> struct Plan {
    static constexpr int full_passes() { int n = N, c = 0; while (n >= E && n % E == 0) { n /= E; ++c; } return c; }
    static constexpr int first_radix() { int n = N; while (n >= E && n % E == 0) n /= E; return n; }  // 1 if none
    static constexpr int T = N / E;
    static_assert(N % E == 0, "N must be a multiple of E");
};

// LDS padding: one element per 16 keeps every pass' scatter conflict-free for ds_write_b64
// (16-lane groups, 32 banks) -- see DESIGN.md "LDS exchange".
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 4); }
template <int N> struct LdsLine { static constexpr int elems = N + (N >> 4); };

// a[t] *= w^t, t in [0, R).  For R > 4 the rotation is split in two levels, t = 4*t1 + t2:
// w^t = (w^4)^t1 * w^t2, so only {w, w^2, w^3} and the powers of w^4 are ever live instead of an (R-1)-entry power
// table.  The three low powers are kept together with their rotated copies i w^t2 (each is used by R/4
// products, which then cost two packed instructions); the high powers use the three-instruction product.
// R <= 16: the high powers w^4, w^8, w^12 are products (depth <= 5, a few ulp on the twiddle).
// R >= 32: they are read from the twiddle table (tw[(4 t1 step) mod table]: exact entries, L1-resident) -- a
// product chain would be 7 or 15 deep.
template <int R>
__device__ __forceinline__ void apply_twiddles(c32 (&a)[R], c32 w, const c32* __restrict__ tw, int step, int mask) {
    if constexpr (R == 2) {
        a[1] = cmul(a[1], w);
    } else if constexpr (R == 4) {
        const c32 w2 = cmul(w, w);
        a[1] = cmul(a[1], w);
        a[2] = cmul(a[2], w2);
        a[3] = cmul(a[3], cmul(w2, w));
    } else {
        static_assert(R % 4 == 0 && R <= 64, "two-level twiddle split supports R = 8 .. 64");
        constexpr int R1 = R / 4;
        const c32 wr = crot(w);
        const c32 w2 = cmul_r(w, w, wr), w2r = crot(w2);
        const c32 w3 = cmul_r(w2, w, wr), w3r = crot(w3);
        if constexpr (R <= 16) {
            const c32 w4 = cmul_r(w2, w2, w2r);
            c32 hi[R1];
            hi[0] = mk(1.0f, 0.0f);
            hi[1] = w4;
            if constexpr (R1 > 2) { hi[2] = cmul(w4, w4); hi[3] = cmul(hi[2], w4); }
#pragma unroll
            for (int t = 1; t < R; ++t) {
                const int t1 = t / 4, t2 = t % 4;
                c32 v = a[t];
                if (t2 == 1) v = cmul_r(v, w, wr);
                if (t2 == 2) v = cmul_r(v, w2, w2r);
                if (t2 == 3) v = cmul_r(v, w3, w3r);
                if (t1 > 0) v = cmul(v, hi[t1]);
                a[t] = v;
            }
        } else {
#pragma unroll
            for (int t1 = 0; t1 < R1; ++t1) {
                c32 h = mk(1.0f, 0.0f);
                if (t1 > 0) h = tw[(4 * t1 * step) & mask];
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const int t = 4 * t1 + t2;
                    if (t == 0) continue;
                    c32 v = a[t];
                    if (t2 == 1) v = cmul_r(v, w, wr);
                    if (t2 == 2) v = cmul_r(v, w2, w2r);
                    if (t2 == 3) v = cmul_r(v, w3, w3r);
                    if (t1 > 0) v = cmul(v, h);
                    a[t] = v;
                }
            }
        }
    }
}

// One pass of radix R over the E registers of thread j.  `emit(pos, padded_pos, value, slot)`
// receives every output with its line position, its padded LDS index and the register slot it
// would occupy after the exchange of the last pass.
// TWS: stride of the twiddle table (the table holds e^{+2 pi i k / (N*TWS)}: a length-N transform inside a
// context whose table was made for N*TWS points, see the split kernels of ocean_kernels.hpp).
// HWTW: the pass' base twiddle e^{+2 pi i step / (N TWS)} comes from the transcendental unit (v_cos_f32 / v_sin_f32 take
// revolutions, step / (N TWS) is exact in fp32; max abs error 1.25e-7, tools/sincos_acc.hip) instead of the table in
// memory: at the latency-bound sizes the dependent table load of every pass (~0.3 us of L2 latency with one wave per
// SIMD) is the longest single item of a transform.  R <= 16 only (larger radices read high powers from the table).
template <int TABLE, bool HWTW>
__device__ __forceinline__ c32 base_twiddle(const c32* __restrict__ tw, int step) {
    if constexpr (HWTW) {
        const float rev = (float)step * (1.0f / (float)TABLE);
        return mk(cos_rev(rev), sin_rev(rev));
    } else {
        return tw[step];
    }
}
template <int N, int E, int R, int NS, int TWS = 1, bool HWTW = false, class Emit>
__device__ __forceinline__ void fft_pass(c32 (&reg)[E], int j, const c32* __restrict__ tw, Emit&& emit) {
    constexpr int T = N / E;
    constexpr int U = E / R;
    static_assert(!HWTW || R <= 16, "hardware twiddles: radix <= 16");
    c32 w = mk(1.0f, 0.0f);
    constexpr int TW_MASK = N * TWS - 1;                               // the table holds e^{+2 pi i k / (N TWS)}, k < N TWS
    int step = 0;
    if constexpr (NS > 1 && U == 1) { step = (j & (NS - 1)) * (TWS * (N / (NS * R))); w = base_twiddle<N * TWS, HWTW>(tw, step); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int jv = j + u * T;
        const int k = jv & (NS - 1);
        if constexpr (NS > 1 && U > 1) { step = k * (TWS * (N / (NS * R))); w = base_twiddle<N * TWS, HWTW>(tw, step); }
        c32 a[R], b[R];
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = reg[u + t * U];
        if constexpr (NS > 1) apply_twiddles<R>(a, w, tw, step, TW_MASK);
        Dft<R>::run(a, b);
        const int base = (jv / NS) * (NS * R) + k;
        // lds_pad(base + s*NS) == lds_pad(base) + s*NS + ((s*NS) >> 4) for power-of-two NS, R
        // (one base VGPR + immediates instead of R address registers)
        const int pbase = lds_pad(base);
#pragma unroll
        for (int s = 0; s < R; ++s) emit(base + s * NS, pbase + s * NS + ((s * NS) >> 4), b[s], u + s * U);
    }
}

// Exchange through one padded LDS line buffer: scatter `reg` outputs of a pass, then gather
// positions j + e*T.  `bar()` is the workgroup barrier (all threads of the WG call it).
template <int N, int E, int R, int NS, int TWS = 1, int SYNC_T = 1024, bool HWTW = false>
__device__ __forceinline__ void fft_pass_exchange(c32 (&reg)[E], int j, const c32* __restrict__ tw, c32* lds_line) {
    constexpr int T = N / E;
    fft_pass<N, E, R, NS, TWS, HWTW>(reg, j, tw, [&](int, int ppos, c32 v, int) { lds_line[ppos] = v; });
    line_sync<SYNC_T>();   // RAW: the scatter above is visible
    // lds_pad(j + e*T) == lds_pad(j) + e*(T + T/16)   (T is a multiple of 16)
    const c32* g = lds_line + lds_pad(j);
#pragma unroll
    for (int e = 0; e < E; ++e) reg[e] = g[e * (T + T / 16)];
}

// Whole line transform.  On entry reg[e] = x[j + e*T]; on exit reg[e] = X[j + e*T].
// lds_line: LdsLine<N>::elems c32 owned by this line.  Every thread of the workgroup must call
// this the same number of times (with T > 64 it contains __syncthreads()).
template <int N, int E, int TWS = 1, bool CONTIG = false, bool HWTW = false>
__device__ __forceinline__ void fft_line(c32 (&reg)[E], int j, const c32* __restrict__ tw, c32* lds_line) {
    constexpr int SYNC_T = (CONTIG && N / E <= 64) ? N / E : 1024;
    constexpr int R0 = Plan<N, E>::first_radix();
    constexpr int Q = Plan<N, E>::full_passes();
    static_assert(Q >= 1 && Q <= 3, "unsupported N/E combination");
    int ns = 1;
    (void)ns;
    if constexpr (R0 > 1) {
        fft_pass_exchange<N, E, R0, 1, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
        line_sync<SYNC_T>();   // WAR: next scatter reuses the buffer
    }
    constexpr int NS1 = R0;                  // after the optional small pass
    if constexpr (Q == 1) {
        c32 out[E];
        fft_pass<N, E, E, NS1, TWS, HWTW>(reg, j, tw, [&](int, int, c32 v, int slot) { out[slot] = v; });
#pragma unroll
        for (int e = 0; e < E; ++e) reg[e] = out[e];
    } else {
        fft_pass_exchange<N, E, E, NS1, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
        constexpr int NS2 = NS1 * E;
        if constexpr (Q == 2) {
            c32 out[E];
            fft_pass<N, E, E, NS2, TWS, HWTW>(reg, j, tw, [&](int, int, c32 v, int slot) { out[slot] = v; });
#pragma unroll
            for (int e = 0; e < E; ++e) reg[e] = out[e];
        } else {
            line_sync<SYNC_T>();
            fft_pass_exchange<N, E, E, NS2, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
            constexpr int NS3 = NS2 * E;
            c32 out[E];
            fft_pass<N, E, E, NS3, TWS, HWTW>(reg, j, tw, [&](int, int, c32 v, int slot) { out[slot] = v; });
#pragma unroll
            for (int e = 0; e < E; ++e) reg[e] = out[e];
        }
    }
}

// Same transform, but the final pass scatters into the LDS line (padded positions) and the
// function returns after a barrier: lds_line[lds_pad(n)] = X[n] for the whole line.  Used when
// the global store wants a different thread->element mapping than the FFT's (chunked layouts).
template <int N, int E, int TWS = 1, bool CONTIG = false, bool HWTW = false>
__device__ __forceinline__ void fft_line_to_lds(c32 (&reg)[E], int j, const c32* __restrict__ tw, c32* lds_line) {
    constexpr int SYNC_T = (CONTIG && N / E <= 64) ? N / E : 1024;
    constexpr int R0 = Plan<N, E>::first_radix();
    constexpr int Q = Plan<N, E>::full_passes();
    static_assert(Q >= 2 && Q <= 3, "unsupported N/E combination");
    if constexpr (R0 > 1) {
        fft_pass_exchange<N, E, R0, 1, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
        line_sync<SYNC_T>();
    }
    constexpr int NS1 = R0;
    fft_pass_exchange<N, E, E, NS1, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
    line_sync<SYNC_T>();
    constexpr int NS2 = NS1 * E;
    if constexpr (Q == 2) {
        fft_pass<N, E, E, NS2, TWS, HWTW>(reg, j, tw, [&](int, int ppos, c32 v, int) { lds_line[ppos] = v; });
    } else {
        fft_pass_exchange<N, E, E, NS2, TWS, SYNC_T, HWTW>(reg, j, tw, lds_line);
        line_sync<SYNC_T>();
        constexpr int NS3 = NS2 * E;
        fft_pass<N, E, E, NS3, TWS, HWTW>(reg, j, tw, [&](int, int ppos, c32 v, int) { lds_line[ppos] = v; });
    }
    __syncthreads();
}

}  // namespace ocean
