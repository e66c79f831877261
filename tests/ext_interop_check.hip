// GPU tier, native: the displacement map in memory ANOTHER owner exported as a file descriptor.
// This program plays the graphics API's part -- the reference allocates the image memory itself (src/render.rs:820-869) -- with
// the one exportable allocator this image has: HIP's virtual-memory API (hipMemCreate with a POSIX-fd handle type; a Vulkan
// application would export its VkDeviceMemory with VK_KHR_external_memory_fd instead).  The library imports the descriptor
// (ocean_bind_displacement_fd), frames land in the exporter's memory, and what the exporter reads through ITS OWN mapping must be
// the frame ocean_read_displacement returns from a library-owned map, bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <unistd.h>
#include "ocean_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define OK(x) do { int32_t s_ = (x); if (s_ != OCEAN_OK) { printf("%s -> %d: %s (line %d)\n", #x, (int)s_, ocean_last_error(ctx), __LINE__); return 3; } } while (0)

int main() {
    const int N = 512;
    const size_t map_bytes = (size_t)N * N * 16, offset = 1 << 16;      // the map does not start at the allocation's first byte
    CK(hipSetDevice(0));
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t size = (map_bytes + offset + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, size, &prop, 0));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, size, gran, nullptr, 0));
    CK(hipMemMap(va, size, 0, h, 0));
    hipMemAccessDesc acc;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, size, &acc, 1));
    CK(hipMemset(va, 0xFF, size));
    int fd = -1;
    CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));

    OceanContext* ctx = nullptr;
    if (ocean_context_create_ex(0, N, OCEAN_CTX_FUSED_ONLY, &ctx) != OCEAN_OK) { printf("create: %s\n", ocean_last_error(nullptr)); return 4; }
    std::vector<float> h0((size_t)N * N * 2), om((size_t)N * N);
    unsigned s = 12345u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f - 0.5f; };
    for (auto& v : h0) v = rnd() * 1e-2f;
    for (auto& v : om) v = 0.1f + 3.0f * (rnd() + 0.5f);
    OK(ocean_upload_spectrum(ctx, h0.data(), om.data()));
    if (ocean_bind_displacement_fd(ctx, -1, size, offset) != OCEAN_E_INVALID_ARG) return 5;
    if (ocean_bind_displacement_fd(ctx, fd, size, size) != OCEAN_E_INVALID_ARG) return 6;          // no room behind the offset
    OK(ocean_bind_displacement_fd(ctx, dup(fd), size, offset));
    OK(ocean_frame(ctx, 1.5f, nullptr));
    OK(ocean_sync(ctx));
    std::vector<float> theirs((size_t)N * N * 4), ours((size_t)N * N * 4), guard(4);
    CK(hipMemcpy(theirs.data(), (const char*)va + offset, map_bytes, hipMemcpyDeviceToHost));     // through the EXPORTER's mapping
    CK(hipMemcpy(guard.data(), (const char*)va + offset - 16, 16, hipMemcpyDeviceToHost));
    OK(ocean_bind_displacement(ctx, nullptr));                                                        // back to the library's own map: releases the import
    OK(ocean_frame(ctx, 1.5f, nullptr));
    OK(ocean_read_displacement(ctx, ours.data()));
    if (std::memcmp(theirs.data(), ours.data(), map_bytes) != 0) { printf("the imported map differs from the library's\n"); return 7; }
    unsigned char ff[16]; std::memset(ff, 0xFF, 16);
    if (std::memcmp(guard.data(), ff, 16) != 0) { printf("bytes in front of the map were written\n"); return 8; }
    double mx = 0; for (float v : ours) mx = std::fmax(mx, std::fabs((double)v));
    if (!(mx > 1e-3) || !std::isfinite(mx)) { printf("degenerate frame (max %g)\n", mx); return 9; }
    OK(ocean_bind_displacement_fd(ctx, dup(fd), size, 0));                                             // a second import, released by the destroy
    OK(ocean_frame(ctx, 2.0f, nullptr));
    ocean_context_destroy(ctx);
    close(fd);
    CK(hipMemUnmap(va, size));
    CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, size));
    printf("interop: frame written through an imported file descriptor == library-owned frame, bit for bit (max |texel| %.3f)\n", mx);
    return 0;
}
