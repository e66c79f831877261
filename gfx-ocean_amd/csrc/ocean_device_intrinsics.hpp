// ocean_device_intrinsics.hpp -- the two gfx950-specific scheduling/regalloc helpers the
// kernels use.  (tests/hipemu/ has a host version of this header for the CPU emulation build.)
#pragma once
#include <hip/hip_runtime.h>

namespace ocean {

// Returns x unchanged but opaque to GVN/LICM.  The three per-field FFTs of a fused kernel use
// identical twiddles; without this the compiler keeps ~60 VGPRs of twiddle powers alive across
// the fields (measured: 195 -> 92 VGPRs for k_frame_pass2<4096>), which spills at the
// 128-VGPR budget of a 1024-thread workgroup.
__device__ __forceinline__ int opaque_lane(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Returns x, but only after `dep` has been computed: orders the loads whose addresses derive from
// the result behind the arithmetic that produced `dep` (splits a long load phase in two so that
// the first half's input registers are free before the second half's loads are issued).
__device__ __forceinline__ int opaque_after(int x, float dep) {
    asm volatile("" : "+v"(x) : "v"(dep));
    return x;
}

// x is known to be identical in every lane of the wave: move it to an SGPR so that addresses
// derived from it become scalar bases (one VGPR offset + SGPR base instead of E 64-bit VGPR pairs).
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

}  // namespace ocean
