// ocean.hpp -- C++ host-side mirror of the reference's `mod ocean` / `mod fft` (src/ocean.rs,
// src/fft.rs) and of the compute slice of `Renderer` (src/render.rs) over include/ocean_hip.h.
// Header-only; link with -locean_hip.  Same names, same init/destroy life cycle; errors that the
// reference returns as `Result<_, Box<dyn Error>>` (or `.unwrap()`s) become ocean::Error.
#pragma once
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ocean_hip.h"

namespace ocean_host {

struct Error : std::runtime_error {
    int32_t status;
    Error(int32_t st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

constexpr int RESOLUTION = 512;        // src/render.rs:44
constexpr float DOMAIN_SIZE = 1000.0f; // src/render.rs:46
enum Field : int32_t { DX = OCEAN_FIELD_DX, DY = OCEAN_FIELD_DY, DZ = OCEAN_FIELD_DZ, ALL = OCEAN_FIELD_ALL };

using PropagateLocals = OceanPropagateLocals;   // src/ocean.rs:8-13
using CorrectionLocals = OceanCorrectionLocals; // src/ocean.rs:179-182

class Device {
public:
    explicit Device(int resolution, int ordinal = 0) {
        const int32_t st = ocean_context_create(ordinal, resolution, &ctx_);
        if (st != OCEAN_OK) throw Error(st, ocean_last_error(nullptr));
    }
    // `tiles` independent tiles in one context (N <= 1024): upload_spectrum_tile each, frame_tiles = one frame of every tile per launch pair
    Device(int resolution, int ordinal, int tiles) {
        const int32_t st = ocean_context_create_tiles(ordinal, resolution, tiles, &ctx_);
        if (st != OCEAN_OK) throw Error(st, ocean_last_error(nullptr));
    }
    void upload_spectrum_tile(int tile, const std::vector<std::complex<float>>& h0, const std::vector<float>& omega) {
        check(ocean_upload_spectrum_tile(ctx_, tile, reinterpret_cast<const float*>(h0.data()), omega.data()));
    }
    // the upload's device-side half alone (copy_buffer, src/render.rs:896-915): from memory the GPU can read, asynchronous on `stream`
    void upload_spectrum_device(const void* h0_device, const void* omega_device, int tile = 0, void* stream = nullptr) {
        check(ocean_upload_spectrum_device(ctx_, tile, h0_device, omega_device, stream));
    }
    void frame_tiles(float time, void* stream = nullptr) { check(ocean_frame_tiles(ctx_, time, nullptr, 0, stream)); }
    ~Device() { ocean_context_destroy(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    OceanContext* raw() const { return ctx_; }
    void check(int32_t st) const { if (st != OCEAN_OK) throw Error(st, ocean_last_error(ctx_)); }
    int resolution() const { return ocean_resolution(ctx_); }
    // src/render.rs:742-924
    void upload_spectrum(const std::vector<std::complex<float>>& h0, const std::vector<float>& omega) {
        check(ocean_upload_spectrum(ctx_, reinterpret_cast<const float*>(h0.data()), omega.data()));
    }
    void frame(float time, void* stream = nullptr) { check(ocean_frame(ctx_, time, stream)); }
    // K time steps t0 + dt * i of this tile, one launch pair at N <= 1024 (library-owned maps; ocean_frame_batch)
    void frame_batch(float t0, float dt, int count, void* stream = nullptr) { check(ocean_frame_batch(ctx_, t0, dt, count, nullptr, 0, stream)); }
    std::vector<float> read_batch_displacement(int index) {
        std::vector<float> out((size_t)resolution() * resolution() * 4);
        check(ocean_read_batch_displacement(ctx_, index, out.data()));
        return out;
    }
    // the frame with its normal field (shader/ocean.frag:50-66) as one workload: channel 0..2 on, -1 off
    void set_frame_normals(int source_channel) { check(ocean_set_frame_normals(ctx_, source_channel)); }
    std::vector<float> read_normals() {
        std::vector<float> out((size_t)resolution() * resolution() * 4);
        check(ocean_read_normals(ctx_, out.data()));
        return out;
    }
    // measurement (HIP events on the context stream)
    float time_frames(int frames, float t0 = 0.0f, float dt = 1.0f / 60.0f) { float ms = 0; check(ocean_time_frames(ctx_, frames, t0, dt, &ms)); return ms; }
    std::vector<float> time_frame_batches(int batches, int frames_per_batch, float t0 = 0.0f, float dt = 1.0f / 60.0f) {
        std::vector<float> ms((size_t)(batches > 0 ? batches : 0));
        check(ocean_time_frame_batches(ctx_, batches, frames_per_batch, t0, dt, ms.data()));
        return ms;
    }
    // SURVEY 8a Q1/Q2 switches; OCEAN_QUIRKS_REFERENCE (default) = the shipped shaders
    void set_quirks(uint32_t quirks) { check(ocean_set_quirks(ctx_, quirks)); }
    uint32_t quirks() const { return ocean_quirks(ctx_); }
    void sync() { check(ocean_sync(ctx_)); }
    std::vector<float> read_displacement() {
        std::vector<float> out((size_t)resolution() * resolution() * 4);
        check(ocean_read_displacement(ctx_, out.data()));
        return out;
    }
private:
    OceanContext* ctx_ = nullptr;
};

// src/ocean.rs:15-177
class Propagation {
public:
    static Propagation init(Device& d) { Propagation p(d); d.check(ocean_propagation_init(d.raw(), &p.h_)); return p; }
    void dispatch(const PropagateLocals& l, void* stream = nullptr) { dev_->check(ocean_propagate(h_, &l, stream)); }
    void destroy() { ocean_propagation_destroy(h_); h_ = nullptr; }
private:
    explicit Propagation(Device& d) : dev_(&d) {}
    Device* dev_;
    OceanPropagation* h_ = nullptr;
};

// src/ocean.rs:184-328
class Correction {
public:
    static Correction init(Device& d) { Correction c(d); d.check(ocean_correction_init(d.raw(), &c.h_)); return c; }
    void dispatch(const CorrectionLocals& l, void* stream = nullptr) { dev_->check(ocean_correct(h_, &l, stream)); }
    void destroy() { ocean_correction_destroy(h_); h_ = nullptr; }
private:
    explicit Correction(Device& d) : dev_(&d) {}
    Device* dev_;
    OceanCorrection* h_ = nullptr;
};

// src/fft.rs:7-111
class Fft {
public:
    static Fft init(Device& d) { Fft f(d); d.check(ocean_fft_init(d.raw(), &f.h_)); return f; }
    void row_pass(Field f = ALL, void* stream = nullptr) { dev_->check(ocean_fft_rows(h_, f, stream)); }  // render.rs:1158-1179
    void col_pass(Field f = ALL, void* stream = nullptr) { dev_->check(ocean_fft_cols(h_, f, stream)); }  // render.rs:1210-1231
    void destroy() { ocean_fft_destroy(h_); h_ = nullptr; }
private:
    explicit Fft(Device& d) : dev_(&d) {}
    Device* dev_;
    OceanFft* h_ = nullptr;
};

// The recorder of src/render.rs:1122-1310 (8 dispatches; stream order replaces the 4 barriers).
inline void render(Device& d, Propagation& p, Fft& fft, Correction& c, float time, float domain = DOMAIN_SIZE,
                   void* stream = nullptr) {
    p.dispatch(PropagateLocals{time, d.resolution(), domain}, stream);
    fft.row_pass(ALL, stream);
    fft.col_pass(ALL, stream);
    c.dispatch(CorrectionLocals{(uint32_t)d.resolution()}, stream);
}

}  // namespace ocean_host
