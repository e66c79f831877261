// Native C++ caller of the C ABI through the host mirror (gfx_ocean_amd/csrc/host/ocean.hpp).
//
//   host_mirror_check                               CPU tier: compile + link + ABI version only
//   host_mirror_check spectrum.bin omega.bin crop.f32 [time]
//       GPU tier (tests/test_gpu_native.py): what a C++ gfx-ocean would do where `Renderer::new` /
//       `Renderer::render` sit (src/render.rs:223-225, 742-924, 1101-1310): decode the reference's bincode
//       inputs, upload, record the 8 staged dispatches (ocean_host::render) and the fused frame (Device::frame)
//       at N = 512, and compare the top-left 64 x 64 texels of each displacement map with the golden crop
//       (raw float32 [64][64][4], exported by the test from tests/golden/spirv_frame512_t1.npz = the
//       reference's shipped SPIR-V executed on the same inputs).  Exit code 0 = both within 1e-4.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "../gfx_ocean_amd/csrc/host/ocean.hpp"

namespace {

// bincode 1.x Vec<T>: u64-LE element count + little-endian payload (src/render.rs:769-771, 808-810)
template <class T> std::vector<T> read_bincode(const char* path, size_t expect) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    uint64_t count = 0;
    f.read(reinterpret_cast<char*>(&count), sizeof count);
    if (count != expect) throw std::runtime_error(std::string(path) + ": unexpected element count");
    std::vector<T> v((size_t)count);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
    if (!f) throw std::runtime_error(std::string(path) + ": short read");
    return v;
}

// SURVEY 8d metric on the crop: per channel max|a - b| / max|b|; alpha must be exactly 0 (correction.comp:34)
double crop_error(const std::vector<float>& img, int n, const std::vector<float>& gold, int crop) {
    double worst = 0.0;
    for (int c = 0; c < 3; ++c) {
        double num = 0.0, den = 0.0;
        for (int y = 0; y < crop; ++y)
            for (int x = 0; x < crop; ++x) {
                const double a = img[((size_t)y * n + x) * 4 + c], b = gold[((size_t)y * crop + x) * 4 + c];
                num = std::fmax(num, std::fabs(a - b));
                den = std::fmax(den, std::fabs(b));
            }
        worst = std::fmax(worst, num / den);
    }
    for (int y = 0; y < crop; ++y)
        for (int x = 0; x < crop; ++x)
            if (img[((size_t)y * n + x) * 4 + 3] != 0.0f) return 1.0;
    return worst;
}

}  // namespace

int main(int argc, char** argv) {
    if (ocean_abi_version() != OCEAN_ABI_VERSION) return 1;
    if (argc < 4) return 0;
    try {
        constexpr int N = ocean_host::RESOLUTION, CROP = 64;
        const float time = (argc > 4) ? (float)std::atof(argv[4]) : 1.0f;
        const auto h0 = read_bincode<std::complex<float>>(argv[1], (size_t)N * N);
        const auto omega = read_bincode<float>(argv[2], (size_t)N * N);
        std::vector<float> gold((size_t)CROP * CROP * 4);
        {
            std::ifstream g(argv[3], std::ios::binary);
            g.read(reinterpret_cast<char*>(gold.data()), (std::streamsize)(gold.size() * sizeof(float)));
            if (!g) throw std::runtime_error("golden crop: short read");
        }
        ocean_host::Device d(N);
        auto fft = ocean_host::Fft::init(d);                       // src/render.rs:223
        auto propagation = ocean_host::Propagation::init(d);       // :224
        auto correction = ocean_host::Correction::init(d);         // :225
        d.upload_spectrum(h0, omega);                              // :742-924
        d.frame(time);                                             // :1101-1310 in one call: the product path (2 fused launches)
        const double e_fused = crop_error(d.read_displacement(), N, gold, CROP);
        ocean_host::render(d, propagation, fft, correction, time); // the same frame dispatch by dispatch (8 staged calls)
        const double e_staged = crop_error(d.read_displacement(), N, gold, CROP);
        propagation.destroy(); fft.destroy(); correction.destroy();
        std::printf("native c++: staged %.3e fused %.3e (normalised max on the 64x64 crop, tolerance 1e-4)\n", e_staged, e_fused);
        return (e_staged <= 1e-4 && e_fused <= 1e-4) ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_mirror_check: %s\n", e.what());
        return 3;
    }
}
