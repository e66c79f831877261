// cumask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (For an experiment that runs pass 1
// of frame f + 1 and pass 2 of frame f side by side on disjoint sets of CUs.)  Launches many one-wave workgroups that record
// XCC_ID and HW_ID on a masked stream and prints, per mask, how many distinct CUs of every XCD were seen.
//   hipcc --offload-arch=gfx950 -O3 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_where(uint32_t* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay a little so that the launch spreads over every CU the stream may use
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(20);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void probe(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    const int G = 4096;
    uint32_t* d; CK(hipMalloc(&d, G * 8)); CK(hipMemset(d, 0xff, G * 8));
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(2 * G); CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;          // xcc -> {se, sh, cu}
    for (int i = 0; i < G; ++i) cus[h[2 * i + 1] & 0xf].insert((h[2 * i] >> 8) & 0xff);
    size_t total = 0;
    printf("%-28s", name);
    for (auto& kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total %zu CUs\n", total);
    if (total <= 40) {
        for (auto& kv : cus) { printf("    xcc%u:", kv.first); for (unsigned v : kv.second) printf(" se%u.sh%u.cu%u", (v >> 5) & 7, (v >> 4) & 1, v & 15); printf("\n"); }
    }
    CK(hipFree(d)); CK(hipStreamDestroy(s));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs\n", p.name, p.multiProcessorCount);
    auto bits = [](std::initializer_list<std::pair<int, int>> ranges) {
        std::vector<uint32_t> m(8, 0u);
        for (auto r : ranges) for (int b = r.first; b < r.second; ++b) m[b / 32] |= 1u << (b % 32);
        return m;
    };
    probe("all 256 bits", bits({{0, 256}}));
    probe("bits 0..127", bits({{0, 128}}));
    probe("bits 128..255", bits({{128, 256}}));
    probe("bits 0..31", bits({{0, 32}}));
    probe("bits 0..7", bits({{0, 8}}));
    probe("bits 8..15", bits({{8, 16}}));
    probe("bits 0..63", bits({{0, 64}}));
    probe("bits 0..191", bits({{0, 192}}));
    probe("bits 192..255", bits({{192, 256}}));
    {   // every fourth bit
        std::vector<uint32_t> m(8, 0x11111111u);
        probe("every 4th bit", m);
    }
    {   // three of every four bits
        std::vector<uint32_t> m(8, 0xEEEEEEEEu);
        probe("3 of every 4 bits", m);
    }
    return 0;
}
