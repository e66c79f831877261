// ocean_device_intrinsics.hpp -- the two gfx950-specific scheduling/regalloc helpers the
// kernels use.  (tests/hipemu/ has a host version of this header for the CPU emulation build.)
#pragma once
#include <hip/hip_runtime.h>

namespace ocean {

// Race hunting (-DOCEAN_RACE_JITTER): every workgroup barrier of the kernels is wrapped in pseudo-random
// wave-uniform sleeps (0 .. ~2.7 us, from the low bits of the shader clock), before and after.  The arithmetic is
// untouched, so a frame of this build must be BIT-identical to the product build's; a missing or misplaced barrier --
// two waves touching the same LDS words with only "they usually arrive in this order" between them -- shows up as a
// difference within a few hundred frames (tests/test_gpu_race.py::test_barrier_jitter_build_is_bit_identical).
// AddressSanitizer for the device is not available on this pool (xnack-, no instrumented runtime:
// profiles/r03_run16_asan_attempt_log.txt).
#ifdef OCEAN_RACE_JITTER   // (a test build: tests/test_gpu_race.py compiles it next to the product and compares checksums)
__device__ __forceinline__ void race_jitter() {
    const unsigned t = (unsigned)__builtin_readcyclecounter();
    switch ((t >> 2) & 7u) {                                       // wave-uniform (scalar clock)
        case 0: break;
        case 1: __builtin_amdgcn_s_sleep(1); break;
        case 2: __builtin_amdgcn_s_sleep(3); break;
        case 3: __builtin_amdgcn_s_sleep(7); break;
        case 4: __builtin_amdgcn_s_sleep(15); break;
        case 5: __builtin_amdgcn_s_sleep(31); break;
        case 6: __builtin_amdgcn_s_sleep(63); break;
        default: __builtin_amdgcn_s_sleep(100); break;
    }
}
__device__ __forceinline__ void jittered_syncthreads() {
    race_jitter();
    __syncthreads();
    race_jitter();
}
#define __syncthreads() ::ocean::jittered_syncthreads()
#endif

// A complex number is a 2-vector of floats living in an even-aligned VGPR pair, so that complex arithmetic
// maps onto gfx950's packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two lanes per
// instruction, swizzles through op_sel): pass 1's phases are VALU-issue-bound (tools/timeline.hip), and the
// packed form halves the instruction count of the butterflies.  The vector primitives below are the whole
// interface; tests/hipemu has scalar twins.
typedef float c32 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ c32 mk(float re, float im) { return c32{re, im}; }
__host__ __device__ __forceinline__ c32 xx(c32 a) { return __builtin_shufflevector(a, a, 0, 0); }
__host__ __device__ __forceinline__ c32 yy(c32 a) { return __builtin_shufflevector(a, a, 1, 1); }
__host__ __device__ __forceinline__ c32 yx(c32 a) { return __builtin_shufflevector(a, a, 1, 0); }
__host__ __device__ __forceinline__ c32 vfma(c32 a, c32 b, c32 c) { return __builtin_elementwise_fma(a, b, c); }

// Returns x unchanged but opaque to GVN/LICM.  The three per-field FFTs of a fused kernel use
// identical twiddles; without this the compiler keeps ~60 VGPRs of twiddle powers alive across
// the fields (measured: 195 -> 92 VGPRs for k_frame_pass2<4096>), which spills at the
// 128-VGPR budget of a 1024-thread workgroup.
__device__ __forceinline__ int opaque_lane(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// The two values are computed HERE: an opaque use that keeps the compiler from sinking their arithmetic to a later
// block (and their inputs alive until then).
__device__ __forceinline__ void pin_here(c32& a, c32& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// 16-byte store with the non-temporal hint (global_store_dwordx4 ... nt): the intermediate and the
// displacement map are written once and not re-read by the writing kernel; keeping them from
// allocating in the XCD L2 was measured -23% on pass 1 in isolation and -3..5% per frame at
// N = 4096.  Compile-time only: a run-time switch around the 16 stores of an epilogue made the
// compiler keep all 64 output VGPRs live across one branch (pass 2: 104 -> 177 VGPRs).
typedef float ocean_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_float4_nt(float4* p, float4 v) {
    ocean_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<ocean_v4f*>(p));
}

// Global-memory hand-over inside ONE workgroup: every thread's earlier global stores become visible to its
// other waves after the call.  Workgroup scope is enough and cheap (a wait for the stores; the waves of a
// workgroup share the CU's write-through L1); a device-scope __threadfence() here compiles to
// buffer_wbl2 + buffer_inv, i.e. writes back and invalidates the whole L2 of the XCD (measured: +20 us on
// pass 1 at N = 4096).
__device__ __forceinline__ void workgroup_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Synchronisation among the T threads that share one line of an LDS exchange (fft_core.hpp).  T > 64: the line is
// spread over several waves -> workgroup barrier.  T <= 64 (T divides 64, lines are T consecutive lanes): the whole
// line lives in ONE wave, whose LDS instructions execute in program order -- a compiler-level fence at wavefront scope
// is all that is needed, no s_barrier (at N = 512 a pass-1 workgroup is three one-wave lines that would otherwise
// wait for each other five times per transform).
template <int T> __device__ __forceinline__ void line_sync() {
    if constexpr (T <= 64) {
        static_assert(64 % T == 0, "lines must not straddle waves");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Timeline probes for tools/timeline.hip (compiled in only with -DOCEAN_TIMELINE; the product build has none):
// lane 0 of a workgroup records the 100 MHz wall clock at a few points, slot 15 the hardware id.
#ifdef OCEAN_TIMELINE
__device__ unsigned long long* ocean_tl;
// scalar-only probe (no lane branch, no VGPRs: a lane-0 branch moved the register allocation of pass 1 from 9 to
// 112 spills); every wave of the workgroup writes the slot, the last one wins
__device__ __forceinline__ unsigned long long* ocean_tl_uniform(unsigned long long* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (unsigned long long*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void ocean_tl_probe(unsigned long long* slot) {
    unsigned long long t;
    slot = ocean_tl_uniform(slot);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0\n\ts_dcache_wb" : "=&s"(t) : "s"(slot) : "memory");
}
__device__ __forceinline__ void ocean_tl_hwid(unsigned long long* slot) {
    unsigned lo, hi;
    slot = ocean_tl_uniform(slot);
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(lo), "=s"(hi));
    const unsigned long long id = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_dcache_wb" : : "s"(id), "s"(slot) : "memory");
}
#define OCEAN_TL(k)                                                            \
    do {                                                                       \
        ocean_tl_probe(ocean_tl + (size_t)blockIdx.x * 16 + (k));              \
        if ((k) == 0) ocean_tl_hwid(ocean_tl + (size_t)blockIdx.x * 16 + 15);  \
    } while (0)
#else
#define OCEAN_TL(k)
#endif

// The wave's slot on its SIMD (HW_ID.WAVE_ID, bits 3:0).
__device__ __forceinline__ unsigned hw_wave_slot() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(id));
    return id;
}

__device__ __forceinline__ void wave_sleep_127() { __builtin_amdgcn_s_sleep(127); }   // 127 x 64 clocks = 3.4 us at 2.4 GHz

// sin / cos of 2*pi*x for x in [-0.5, 0.5] revolutions: the gfx950 transcendental unit
// (v_sin_f32 / v_cos_f32 take their argument in revolutions).  Measured max abs error on that
// interval: 1.25e-7 (tools/sincos_acc.hip; ocml's sincospif: 5.2e-8) at 2 instructions instead of ~45.
__device__ __forceinline__ float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }

// BASELINE config 5: the initial spectrum stored as two fp16 (re, im) per texel, scaled by a power
// of two; all arithmetic stays fp32.  v_cvt_f32_f16 x2.
typedef _Float16 ocean_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c32 unpack_half2(uint32_t bits, float descale) {
    const ocean_h2 h = __builtin_bit_cast(ocean_h2, bits);
    return mk((float)h.x * descale, (float)h.y * descale);
}

// ---- 16-bit block-floating intermediate (opt-in precision mode, SURVEY 8d "B_frame16") ---------------------------------
// Maximum of a NON-NEGATIVE value over the 64 lanes of the wave, returned in every lane: four DPP steps inside the
// rows of 16 lanes (quad permutes, half-row and row mirrors: one v_max with a DPP operand each), then the four rows
// through readlane.  (North_star's "wavefront shuffle reduction": this is the one reduction the path has.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int b = __builtin_bit_cast(int, v);                            // non-negative floats order like their bit patterns
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, false));   // row_half_mirror
    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, false));   // row_mirror
    const int m = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
                      max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
    return __builtin_bit_cast(float, m);
}
// Block scale for int16 mantissas: the power of two 2^(e-15) with |v| < 2^e for every v of the block (maximum m), and
// its reciprocal; q = rint(v * inv) lies in [-32768, 32768] and the pack saturates.  m = 0 or denormal: scale 2^-126.
__device__ __forceinline__ void block_scale_i16(float m, float& scale, float& inv) {
    int E = (__builtin_bit_cast(int, m) >> 23) & 255;              // m in [2^(E-127), 2^(E-126))
    E = (E < 15) ? 15 : E;
    scale = __builtin_bit_cast(float, (E - 14) << 23);             // 2^(E - 141)
    inv = __builtin_bit_cast(float, (268 - E) << 23);              // 2^(141 - E)
}
typedef short ocean_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_i16x2(c32 v, float inv) {
    const ocean_s2 p = __builtin_amdgcn_cvt_pk_i16((int)rintf(v.x * inv), (int)rintf(v.y * inv));   // saturating
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ c32 unpack_i16x2(uint32_t bits, float scale) {
    return mk((float)(short)(bits & 0xFFFFu), (float)((int)bits >> 16)) * scale;
}

// ---- LDS-DMA (global -> LDS without VGPRs): the loader of fused pass 1 at N >= 2048 (half_load_AB_dma) ------------------
// One wave instruction moves 64 x 16 bytes: lane l reads 16 bytes at (wave-uniform base + its 32-bit offset) and the data
// lands in LDS at (wave-uniform lds_dst + 16 l) -- lane-linear, so the LDS image of a contiguous kilobyte is the kilobyte.
// Issued through inline asm on purpose: hipcc counts the builtin (__builtin_amdgcn_global_load_lds) as a pending LDS
// write and waits vmcnt(0) in front of every ds_read that may alias it (measured on ROCm 7.2), which drains a ring of
// pieces in flight; an asm statement is outside its bookkeeping, the waits below are ours.  M0 (the LDS destination)
// is saved and restored inside the statement; `s_nop 1` + the two s_mov + `s_nop 0` are the five wait states between a
// readfirstlane-written SGPR and the memory instruction that reads it.
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <bool NT>
__device__ __forceinline__ void glds16(const void* base_uniform, uint32_t lane_offset, uint32_t lds_dst_uniform) {
    const uint64_t b = (uint64_t)base_uniform;
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_nop 1\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_offset), "s"(sb), "s"(dst) : "memory");
    else
        asm volatile("s_nop 1\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_offset), "s"(sb), "s"(dst) : "memory");
}
// "At most K of this wave's DMA instructions are still in flight, and its LDS reads have returned."  The data of the
// others is in LDS for THIS wave; for the rest of the workgroup after the barrier that follows (dma_barrier).
template <int K> __device__ __forceinline__ void dma_wait() {
    static_assert(K >= 0 && K < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(K) : "memory");
}
// Workgroup barrier that does NOT wait for DMA in flight (__syncthreads() carries a fence that would).
__device__ __forceinline__ void dma_barrier() {
#ifdef OCEAN_RACE_JITTER
    race_jitter();
#endif
    asm volatile("s_barrier" : : : "memory");
#ifdef OCEAN_RACE_JITTER
    race_jitter();
#endif
}

// x is known to be identical in every lane of the wave: move it to an SGPR so that addresses
// derived from it become scalar bases (one VGPR offset + SGPR base instead of E 64-bit VGPR pairs).
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

}  // namespace ocean
