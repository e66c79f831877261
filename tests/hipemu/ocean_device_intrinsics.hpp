// Host stand-in for gfx-ocean_amd/csrc/ocean_device_intrinsics.hpp (CPU emulation build only).
#pragma once
namespace ocean {
static inline int opaque_lane(int x) { return x; }
static inline int wave_uniform(int x) { return x; }
static inline int opaque_after(int x, float) { return x; }
}  // namespace ocean
