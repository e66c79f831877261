// ocean_api.hip -- C ABI (include/ocean_hip.h) over the gfx950 kernels.
//
// Replaces, for the compute path only, what `Renderer::new` / `Renderer::render` do in the
// reference: buffer allocation (src/render.rs:607-670), staging upload (:742-924), descriptor
// wiring (:933-988) and the 8 dispatches + 4 barriers per frame (:1122-1310).  Barriers become
// stream order; descriptor sets become plain device pointers held by the context.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <type_traits>
#include <unordered_set>
#include <vector>

#include "../../include/ocean_hip.h"
#include "ocean_kernels.hpp"
#include "ocean_staged_kernels.hpp"
#include "ocean_aux_kernels.hpp"

using namespace ocean;

namespace {

thread_local std::string g_create_error;

constexpr uint32_t MAGIC_CTX = 0x4F43454E;   // 'OCEN'
constexpr uint32_t MAGIC_FFT = 0x4F464654;
constexpr uint32_t MAGIC_PRO = 0x4F50524F;
constexpr uint32_t MAGIC_COR = 0x4F434F52;

bool supported_n(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384; }

// Every handle the library has handed out and not yet destroyed.  A handle is checked against this set before it
// is dereferenced, so a stage object used after ocean_context_destroy, a double destroy or a stray pointer gets
// OCEAN_E_INVALID_ARG instead of a read of freed memory (the reference has the same life-cycle rule -- stages
// are destroyed before the device, src/render.rs:1383-1438 -- but enforces nothing).
std::mutex g_live_mu;
std::unordered_set<const void*> g_live;
void live_add(const void* p) { std::lock_guard<std::mutex> l(g_live_mu); g_live.insert(p); }
void live_remove(const void* p) { std::lock_guard<std::mutex> l(g_live_mu); g_live.erase(p); }
bool live(const void* p) { if (!p) return false; std::lock_guard<std::mutex> l(g_live_mu); return g_live.count(p) != 0; }
// A context's address can be reused by a later context; stage handles therefore remember the generation of the
// context they were made for, and a stale handle is rejected instead of silently driving the new context.
std::atomic<uint64_t> g_generation{1};

}  // namespace

struct OceanContext {
    uint32_t magic = MAGIC_CTX;
    uint64_t generation = 0;    // unique per ocean_context_create (stage handles compare it)
    int device = 0;
    int n = 0;
    uint32_t flags = 0;         // OCEAN_CTX_* of ocean_context_create_ex
    // ocean_context_create_tile_rank: h0T / omegaT are address ranges of the full size of which only the lines this rank's pass 1 reads
    // are backed by memory (HIP virtual-memory API: the kernels keep indexing absolute lines)
    bool bands = false;
    int32_t tile_rank = 0, tile_world = 1;
    std::vector<std::pair<int, int>> band_blocks;                 // [first, count) in blocks of 32 lines, sorted, disjoint
    struct Mapping { void* va; size_t bytes; hipMemGenericAllocationHandle_t handle; };
    std::vector<Mapping> band_maps;
    size_t h0T_reserved = 0, omegaT_reserved = 0;
    int32_t tiles = 1;          // ocean_context_create_tiles: this many independent tiles' static inputs, one frame of each per launch pair
    hipStream_t stream = nullptr;
    bool foreign_stream = false;  // some dispatch ran on a caller stream: readbacks then wait for the whole device
    hipEvent_t ev_a = nullptr, ev_b = nullptr;   // reused by ocean_time_frames (event creation is not free)
    hipEvent_t ev_order = nullptr;               // orders an upload on a caller stream against the context stream (no timing)
    // natural-layout buffers of the staged path (src/render.rs:608-670)
    c32* h0 = nullptr;          // initial_spec
    float* omega = nullptr;     // omega_buffer
    c32* field[3] = {nullptr, nullptr, nullptr};   // dx_spec, dy_spec, dz_spec
    // Staged path, N <= 4096: ocean_fft_rows hands the field to ocean_fft_cols / ocean_correct in the 4 x 4-chunk
    // layout (k_stage_rows / k_stage_cols, ocean_kernels.hpp).  Per field: which copy holds the current contents.
    c32* cfield[3] = {nullptr, nullptr, nullptr};  // chunked copies (layout `lay`), allocated with the context
    // Staged path, N >= 8192: the column pass runs in two steps, the second out of place (k_cols4_a / k_cols4_b); its destination
    // becomes the field and the old buffer the next destination.  Allocated on the first ocean_fft_cols of a field.
    c32* field_alt[3] = {nullptr, nullptr, nullptr};
    // ... and the second step is DEFERRED to the field's next consumer (settle_field): when that is ocean_correct behind the three
    // column passes -- the reference's order -- step B of the three fields and the correction are one kernel (k_cols4_b_correct).
    bool step_b_pending[3] = {false, false, false};
    bool nat_valid[3] = {true, true, true};
    bool chk_valid[3] = {false, false, false};
    bool stage_chunked = false;
    // fused path: transposed static inputs + chunked intermediate
    c32* h0T = nullptr;         // fp32 complex, or (h0_f16) packed half2 in the first N*N*4 bytes
    bool h0_f16 = false;        // BASELINE config 5: fp16 spectrum storage for the fused path
    int scale_log2 = 0;
    float* omegaT = nullptr;
    c32* inter = nullptr;
    InterLayout lay{0, 0, 0, 0};     // three complex fields, all N columns, B = 1   (the staged path's chunked hand-off)
    InterLayout lay_h{0, 0, 0, 0};   // three complex fields, columns 0..N/2-1       (half-spectrum path)
    c32* nyq = nullptr;           // scratch of the half-spectrum path: the Nyquist column's 3 spectra, 3 x N complex
    bool inter16 = false;         // ocean_set_intermediate(OCEAN_INTER_BFP16): int16 intermediate + block scales (N = 8192)
    float* inter_scale = nullptr; // [3][N/64][N/4] block scales of that mode (allocated on first use)
    c32* tw = nullptr;          // e^{+2 pi i k/N}
    float4* out_own = nullptr;  // displacement map (src/render.rs:820-869), linear RGBA32F
    float4* out = nullptr;      // = out_own or the caller's buffer (ocean_bind_displacement)
    hipExternalMemory_t ext_mem = nullptr;   // ocean_bind_displacement_fd: the imported allocation the map currently lives in
    float4* normals = nullptr;    // allocated on first ocean_normals call
    // The frame with the normal field (ocean_set_frame_normals; BASELINE config 3 "height + displacement + normal"):
    // pass 2 also stores the source channel as a dense fp32 plane, which k_normals_plane differentiates behind it.
    int frame_normals = -1;       // source channel 0..2, or -1: the frame is the map alone
    float* plane = nullptr;       // N x N floats, allocated by ocean_set_frame_normals
    // ocean_frame_batch (N <= 1024: K time steps in one launch pair): K intermediates + Nyquist scratches, and K maps when the
    // caller brings no buffer; grown on demand
    c32* batch_inter = nullptr;
    c32* batch_nyq = nullptr;
    int32_t batch_cap = 0;
    float4* batch_out = nullptr;
    int32_t batch_out_cap = 0;
    float* batch_plane = nullptr;       // ... and with the normal field switched on: K planes and K normal fields
    float4* batch_normals = nullptr;
    int32_t batch_normals_cap = 0;
    // what the LAST batch left behind in the library-owned buffers (the capacities above only grow): ocean_read_batch_* refuse an
    // index past it instead of handing out a frame of an earlier, larger batch
    int32_t last_batch_maps = 0;        // frames of the last batch that went into batch_out
    int32_t last_batch_normals = 0;     // normal fields of the last batch (0: the normal field was off)
    float4* positions = nullptr;  // ocean_positions: verts x verts float4, (re)allocated on demand
    int32_t position_verts = 0;
    unsigned long long* checksum_acc = nullptr;   // ocean_checksum_displacement
    bool uploaded = false;        // every tile's inputs are there (uploaded_tiles has `tiles` bits set)
    uint64_t uploaded_tiles = 0;
    float default_domain = 1000.0f;   // src/render.rs:46
    uint32_t quirks = OCEAN_QUIRKS_REFERENCE;   // ocean_set_quirks
    std::string err;
};
struct OceanFft { uint32_t magic = MAGIC_FFT; OceanContext* ctx = nullptr; uint64_t generation = 0; };
struct OceanPropagation { uint32_t magic = MAGIC_PRO; OceanContext* ctx = nullptr; uint64_t generation = 0; };
struct OceanCorrection { uint32_t magic = MAGIC_COR; OceanContext* ctx = nullptr; uint64_t generation = 0; };

namespace {

int32_t fail(OceanContext* ctx, int32_t code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}
int32_t hip_fail(OceanContext* ctx, hipError_t e, const char* what) {
    const int32_t code = (e == hipErrorOutOfMemory) ? OCEAN_E_OOM : OCEAN_E_HIP;
    return fail(ctx, code, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                              \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #expr);        \
    } while (0)

bool valid(const OceanContext* c) { return live(c) && c->magic == MAGIC_CTX; }
template <class H> bool valid_stage(const H* h, uint32_t magic) {
    return live(h) && h->magic == magic && valid(h->ctx) && h->ctx->generation == h->generation;
}

// What a context created with OCEAN_CTX_FUSED_ONLY / OCEAN_CTX_TILE_RANK has no buffers for (ocean_context_create_ex).
int32_t need_staged(OceanContext* c, const char* what) {
    if (c->flags & OCEAN_CTX_FUSED_ONLY)
        return fail(c, OCEAN_E_STATE, std::string(what) + ": this context was created with OCEAN_CTX_FUSED_ONLY and has no natural-layout "
                                      "buffers (the staged dispatches, ocean_read/write_field, ocean_read_spectrum and non-reference quirks need them)");
    return OCEAN_OK;
}
int32_t need_frame(OceanContext* c, const char* what) {
    if (c->flags & OCEAN_CTX_TILE_RANK)
        return fail(c, OCEAN_E_STATE, std::string(what) + ": this context was created with OCEAN_CTX_TILE_RANK and has neither an intermediate nor a "
                                      "map of its own (only ocean_upload_spectrum* and ocean_tile_pass1/2 work on it)");
    return OCEAN_OK;
}
#define NEED(expr) do { const int32_t st_ = (expr); if (st_ != OCEAN_OK) return st_; } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- per-N launchers ------------------------------------------------------------------------
// Launch, optionally with events bound to the dispatch itself (hipExtLaunchKernelGGL: begin/end timestamps of
// this kernel, what rocprofv3 reports) instead of events recorded around it on the stream (which add the
// barrier and signal packets, ~8 us, to a 100 us kernel).
struct Timing { hipEvent_t begin = nullptr, end = nullptr; };
template <typename K, typename... A>
void launch(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t s, Timing t, A... args) {
    if (t.begin) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, t.begin, t.end, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}

template <int N> struct Launch {
    using G = Geo<N>;
    // Staged column pass at N >= 8192: two steps with 1024-point sub-transforms of sixteen columns at a time (k_cols4_a / k_cols4_b)
    // instead of whole columns two at a time (k_fft_lines<COL>: 16-byte pieces, 1.26 TB/s).
    static constexpr bool COLS4 = N >= 8192;
    static constexpr int COLS4_S = COLS4 ? N / 512 : 1, COLS4_LPW = 16;   // sub-transforms of 512 points: two workgroups per CU (ocean_staged_kernels.hpp)
    static constexpr int COLS4_LDS = COLS4_LPW * LinePitch<COLS4 ? N / COLS4_S : 1024>::elems * (int)sizeof(c32);
    static hipError_t prepare() {
        hipError_t e;
        e = hipFuncSetAttribute((const void*)k_fft_lines<N, G::E, G::ROW_LPW, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::row_lds);
        if (e != hipSuccess) return e;
        if constexpr (COLS4) e = hipFuncSetAttribute((const void*)k_cols4_a<N, COLS4_S, G::E, COLS4_LPW>, hipFuncAttributeMaxDynamicSharedMemorySize, COLS4_LDS);
        else e = hipFuncSetAttribute((const void*)k_fft_lines<N, G::E, G::COL_LPW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G::col_lds);
        if (e != hipSuccess) return e;
        if constexpr (G::stage_chunked) {
            e = hipFuncSetAttribute((const void*)k_stage_rows<N, G::E>, hipFuncAttributeMaxDynamicSharedMemorySize, G::stage_lds);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)k_stage_cols<N, G::E>, hipFuncAttributeMaxDynamicSharedMemorySize, G::stage_lds);
            if (e != hipSuccess) return e;
        }
        return prepare_fused();
    }
    // Lines per pass-1 workgroup of the fused frame, per size (measured best): one column at 512 (the
    // latency path), two where two co-resident workgroups pay (256, 1024, 2048) and where four lines do not fit the LDS
    // (8192), four at 4096.  N >= 8192 runs the split kernels (every line as two interleaved N/2-point transforms); at
    // 16384 two 8192-point sub-lines fill the LDS: one column per workgroup (the size that needs several GPUs).
    static constexpr int PSEL = (N == 512 || N > 8192) ? 1 : ((N <= 2048 || N > 4096) ? 2 : 4);
    static constexpr bool SPLIT = N > 4096;
    static constexpr bool BATCHED = batched_launches<N>;
    using H = Geo<N, PSEL>;
    static_assert(!SPLIT || H::can_split, "split geometry");
    static constexpr bool I16_BUILT = SPLIT && H::P == 2;      // the opt-in 16-bit intermediate: N = 8192 (BASELINE config 5)
    // the shipped kernel instances of this size (fp32 / fp16-stored spectrum; split: + the opt-in 16-bit intermediate)
    template <bool H16> static constexpr auto pass1_kernel() { return k_half_pass1<N, H::E1, H::P, H16, H::dma, H::fpar>; }
    // Pass 1 of a BATCHED launch (ocean_frame_batch, count > 1): two columns per workgroup and no field-parallel groups where the
    // single frame has them (Geo<.., THROUGHPUT>), the arithmetic of the single-frame kernel (SHR as there: bit-identical frames).
    // Measured (r05_run4, A/B on one box, frames/s at K = 8 / 16 / 64): N = 512 255k / 311-319k / 371-373k with the single-frame
    // geometry, 296-297k / 383k / 485k with this one; N = 256 586-595k / 827-830k / 1.23M against 600-613k / 842-861k / 1.42M, but
    // 399-421k against 346-371k at K = 4: from 8 frames on at 256.  (N = 1024 has no field-parallel groups, and sits at 5.8 TB/s.)
    using HB = Geo<N, H::fpar ? 2 : PSEL, true>;
    static constexpr bool BATCH_GEO = BATCHED && H::fpar && !SPLIT;
    static constexpr int BATCH_GEO_MIN_COUNT = (N == 256) ? 8 : 2;
    template <bool H16> static constexpr auto pass1_batch_kernel() { return k_half_pass1<N, HB::E1, HB::P, H16, HB::dma, HB::fpar, !H::fpar>; }
    template <bool H16, bool I16> static constexpr auto pass1_split_kernel() { return k_half_pass1_split<N, H::E1S, H::P, H16, I16>; }
    template <bool SHARD, bool PLANE = false> static constexpr auto pass2_kernel() { return k_half_pass2<N, H::E2, CHUNK_W, H::R2h, H::p2_group, H::ppar, SHARD, PLANE>; }
    // Pass 2 at N >= 8192: real-output rows (three N/2-point transforms per row, half the LDS and half the threads of a row:
    // k_half_pass2_real).  Measured against two N-point transforms per row (r04_run23, one box, two repetitions): pass 2 at
    // 8192 440 -> 350 us, at 16384 2.60 -> 1.67-1.72 ms; at 4096 88-91 -> 97-98 us and at 2048 17.8 -> 22.6 us (a row of
    // 128 or 64 threads is a longer serial chain than it saves), so the sizes below keep k_half_pass2.
    static constexpr bool REAL2 = SPLIT;
    template <bool SHARD, bool I16, bool PLANE = false> static constexpr auto pass2_real_kernel() { return k_half_pass2_real<N, H::E, CHUNK_W, H::p2_group, SHARD, I16, 4, PLANE>; }
    static hipError_t prepare_fused() {
        hipError_t e = hipSuccess;
        auto lds = [&](auto kernel, int bytes) {
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        };
        if constexpr (SPLIT) {
            lds(pass1_split_kernel<false, false>(), H::split_lds1); lds(pass1_split_kernel<true, false>(), H::split_lds1);
            lds(pass2_real_kernel<false, false>(), H::real_lds2); lds(pass2_real_kernel<true, false>(), H::real_lds2);
            lds(pass2_real_kernel<false, false, true>(), H::real_lds2);
            if constexpr (I16_BUILT) {
                lds(pass1_split_kernel<false, true>(), H::split_lds1);  lds(pass1_split_kernel<true, true>(), H::split_lds1);
                lds(pass2_real_kernel<false, true>(), H::real_lds2); lds(pass2_real_kernel<false, true, true>(), H::real_lds2);
            }
        } else {
            lds(pass1_kernel<false>(), H::half_lds1); lds(pass1_kernel<true>(), H::half_lds1);
            if constexpr (BATCH_GEO) { lds(pass1_batch_kernel<false>(), HB::half_lds1); lds(pass1_batch_kernel<true>(), HB::half_lds1); }
            lds(pass2_kernel<false>(), H::half_lds2); lds(pass2_kernel<true>(), H::half_lds2);
            lds(pass2_kernel<false, true>(), H::half_lds2);
        }
        return e;
    }
    // Fused pass 1 on the column groups [x_group0, x_group0 + groups) of the half spectrum, written in layout `lay` to
    // `inter` (the context's intermediate for ocean_frame; the all-to-all send buffer of a sharded tile).
    // `count` > 1 (N <= 1024 only: batched_launches<N>): that many time steps time, time + batch.dt, ... in this one launch.
    static void pass1_on(OceanContext* c, float time, float domain, c32* inter, const InterLayout& lay, int groups, int x_group0,
                         hipStream_t s, Timing t, c32* nyq = nullptr, FrameBatch batch = FrameBatch(), int count = 1, bool allow_inter16 = true) {
        if (!nyq) nyq = c->nyq;
        const bool inter16 = c->inter16 && allow_inter16;          // (the sharded tile always ships the fp32 intermediate)
        const float descale = c->h0_f16 ? std::ldexp(1.0f, -c->scale_log2) : 1.0f;
        const void* h0T = c->h0T;
        const float* omT = c->omegaT;
        const c32* tw = c->tw;
        if constexpr (SPLIT) {
            const dim3 g(groups), b(H::split_threads1);
            float* scales = inter16 ? c->inter_scale : nullptr;      // opt-in precision mode (ocean_set_intermediate)
            if constexpr (I16_BUILT) {
                if (inter16) {
                    if (c->h0_f16) launch(pass1_split_kernel<true, true>(), g, b, H::split_lds1, s, t, h0T, descale, omT, inter, c->nyq, tw, lay, time, domain, x_group0, scales);
                    else launch(pass1_split_kernel<false, true>(), g, b, H::split_lds1, s, t, h0T, descale, omT, inter, c->nyq, tw, lay, time, domain, x_group0, scales);
                    return;
                }
            }
            if (c->h0_f16) launch(pass1_split_kernel<true, false>(), g, b, H::split_lds1, s, t, h0T, descale, omT, inter, c->nyq, tw, lay, time, domain, x_group0, scales);
            else launch(pass1_split_kernel<false, false>(), g, b, H::split_lds1, s, t, h0T, descale, omT, inter, c->nyq, tw, lay, time, domain, x_group0, scales);
        } else {
            if constexpr (BATCH_GEO) {
                if (count >= BATCH_GEO_MIN_COUNT) {
                    const dim3 g(HB::half_grid1, count), b(HB::half_threads1);
                    if (c->h0_f16) launch(pass1_batch_kernel<true>(), g, b, HB::half_lds1, s, t, h0T, descale, omT, inter, nyq, tw, lay, time, domain, x_group0, batch);
                    else launch(pass1_batch_kernel<false>(), g, b, HB::half_lds1, s, t, h0T, descale, omT, inter, nyq, tw, lay, time, domain, x_group0, batch);
                    return;
                }
            }
            const dim3 g(groups, count), b(H::half_threads1);
            if (c->h0_f16) launch(pass1_kernel<true>(), g, b, H::half_lds1, s, t, h0T, descale, omT, inter, nyq, tw, lay, time, domain, x_group0, batch);
            else launch(pass1_kernel<false>(), g, b, H::half_lds1, s, t, h0T, descale, omT, inter, nyq, tw, lay, time, domain, x_group0, batch);
        }
    }
    static void pass1(OceanContext* c, float time, float domain, hipStream_t s, Timing t = Timing()) {
        pass1_on(c, time, domain, c->inter, c->lay_h, H::half_grid1, 0, s, t);
    }
    // Pass 2 from `inter` into the map `out`; with the normal field switched on (ocean_set_frame_normals) the PLANE instances,
    // which also store the source channel as the dense plane k_normals_plane reads: `pl`, the caller's choice -- the frame's
    // c->plane, or the K planes of a batch (c->batch_plane; a batch may be ONE frame, so the count says nothing about it).
    // `count` > 1: a batch (pass1_on).
    static void pass2_on(OceanContext* c, const c32* inter, float4* out, float* pl, hipStream_t s, Timing t, FrameBatch batch = FrameBatch(), int count = 1) {
        const c32* tw = c->tw;
        const bool plane = c->frame_normals >= 0;
        const int ch = c->frame_normals;
        if constexpr (REAL2) {
            const dim3 g(N), b(H::real_threads2);
            if constexpr (I16_BUILT) {
                if (c->inter16) {
                    const float* sc = c->inter_scale;
                    if (plane) launch(pass2_real_kernel<false, true, true>(), g, b, H::real_lds2, s, t, inter, out, tw, c->lay_h, sc, pl, ch);
                    else launch(pass2_real_kernel<false, true>(), g, b, H::real_lds2, s, t, inter, out, tw, c->lay_h, sc, (float*)nullptr, 0);
                    return;
                }
            }
            if (plane) launch(pass2_real_kernel<false, false, true>(), g, b, H::real_lds2, s, t, inter, out, tw, c->lay_h, (const float*)nullptr, pl, ch);
            else launch(pass2_real_kernel<false, false>(), g, b, H::real_lds2, s, t, inter, out, tw, c->lay_h, (const float*)nullptr, (float*)nullptr, 0);
        } else {
            const dim3 g(H::half_grid2, count), b(H::half_threads2);
            if (plane) launch(pass2_kernel<false, true>(), g, b, H::half_lds2, s, t, inter, out, tw, c->lay_h, pl, ch, batch);
            else launch(pass2_kernel<false>(), g, b, H::half_lds2, s, t, inter, out, tw, c->lay_h, (float*)nullptr, 0, batch);
        }
    }
    static void pass2(OceanContext* c, hipStream_t s, Timing t = Timing()) { pass2_on(c, c->inter, c->out, c->plane, s, t); }
        // ---- one tile sharded over `world` GPUs (ocean_tile_pass1 / ocean_tile_pass2): the same kernels on this rank's
    // block of half-spectrum columns (pass 1, writing the all-to-all send buffer) and block of rows (pass 2, reading the
    // receive buffer).
    static bool tile_supported(int world, int parts) { return H::tile_supported(world, parts); }
    static void tile_pass1(OceanContext* c, float time, float domain, int rank, int world, int part, int parts, c32* send, hipStream_t s) {
        const int groups = (N / 2 / world / parts) / H::P;
        pass1_on(c, time, domain, send, H::tile_layout(world, parts), groups, (rank * parts + part) * groups, s, Timing(), nullptr, FrameBatch(), 1,
                 /*allow_inter16=*/false);                         // (the 16-bit intermediate is not combined with the sharded tile)
    }
    static void tile_pass2(OceanContext* c, int world, int parts, const c32* recv, float4* out_rows, hipStream_t s) {
        const InterLayout lay = H::tile_layout(world, parts);
        const int rows = N / world;
        const c32* tw = c->tw;
        if constexpr (REAL2)
            hipLaunchKernelGGL((pass2_real_kernel<true, false>()), dim3(rows), dim3(H::real_threads2), H::real_lds2, s, recv, out_rows, tw, lay, (const float*)nullptr, (float*)nullptr, 0);
        else
            hipLaunchKernelGGL((pass2_kernel<true>()), dim3(rows / H::R2h), dim3(H::half_threads2), H::half_lds2, s, recv, out_rows, tw, lay, (float*)nullptr, 0, FrameBatch());
    }
    static void stage_rows(OceanContext* c, int f, hipStream_t s) {
        if constexpr (G::stage_chunked)
            hipLaunchKernelGGL((k_stage_rows<N, G::E>), dim3(G::stage_grid), dim3(G::stage_threads), G::stage_lds, s,
                               (const c32*)c->field[f], c->cfield[f], (const c32*)c->tw, c->lay);
    }
    static void stage_cols(OceanContext* c, int f, hipStream_t s) {
        if constexpr (G::stage_chunked)
            hipLaunchKernelGGL((k_stage_cols<N, G::E>), dim3(G::stage_grid), dim3(G::stage_threads), G::stage_lds, s,
                               c->cfield[f], (const c32*)c->tw, c->lay);
    }
    static void rows(OceanContext* c, c32* data, hipStream_t s) {
        hipLaunchKernelGGL((k_fft_lines<N, G::E, G::ROW_LPW, false>), dim3(G::row_grid), dim3(G::row_threads),
                           G::row_lds, s, data, c->tw);
    }
    static void cols(OceanContext* c, c32* data, hipStream_t s) {
        if constexpr (!COLS4)
            hipLaunchKernelGGL((k_fft_lines<N, G::E, G::COL_LPW, true>), dim3(G::col_grid), dim3(G::col_threads),
                               G::col_lds, s, data, c->tw);
    }
    // step A in place on `data`; step B from `src` into `dst` (the caller swaps the two); step B of the three fields + correction
    static void cols4_a(OceanContext* c, c32* data, hipStream_t s) {
        if constexpr (COLS4)
            hipLaunchKernelGGL((k_cols4_a<N, COLS4_S, G::E, COLS4_LPW>), dim3((N / COLS4_LPW) * COLS4_S), dim3((N / COLS4_S / G::E) * COLS4_LPW), COLS4_LDS, s,
                               data, (const c32*)c->tw);
    }
    static void cols4_b(const c32* src, c32* dst, hipStream_t s) {
        if constexpr (COLS4)
            hipLaunchKernelGGL((k_cols4_b<N, COLS4_S>), dim3((N / ((COLS4_S >= 32) ? 1 : 2) / 256) * (N / COLS4_S)), dim3(256), 0, s, src, dst);
    }
    static void cols4_b_correct(OceanContext* c, hipStream_t s) {
        if constexpr (COLS4)
            hipLaunchKernelGGL((k_cols4_b_correct<N, COLS4_S>), dim3((N / 256) * (N / COLS4_S)), dim3(256), 0, s, (const c32*)c->field[OCEAN_FIELD_DY],
                               (const c32*)c->field[OCEAN_FIELD_DX], (const c32*)c->field[OCEAN_FIELD_DZ], c->out);
    }
};

#define OCEAN_DISPATCH(n, STMT)                           \
    switch (n) {                                          \
        case 256: { using L = Launch<256>; STMT; } break;   \
        case 512: { using L = Launch<512>; STMT; } break;   \
        case 1024: { using L = Launch<1024>; STMT; } break; \
        case 2048: { using L = Launch<2048>; STMT; } break; \
        case 4096: { using L = Launch<4096>; STMT; } break; \
        case 8192: { using L = Launch<8192>; STMT; } break; \
        case 16384: { using L = Launch<16384>; STMT; } break; \
        default: break;                                   \
    }

hipStream_t pick(OceanContext* c, void* stream) {
    if (stream && (hipStream_t)stream != c->stream) c->foreign_stream = true;
    return stream ? (hipStream_t)stream : c->stream;
}
// Readbacks wait for the context stream; once a dispatch has been put on a caller stream they wait for the device
// (the library cannot know whether that stream still exists, so it does not name it).
hipError_t sync_for_readback(OceanContext* c) { return c->foreign_stream ? hipDeviceSynchronize() : hipStreamSynchronize(c->stream); }

void launch_propagate(OceanContext* c, float time, float domain, hipStream_t s) {
    if (c->quirks == OCEAN_QUIRKS_REFERENCE) {     // the whole tile, reference arithmetic: each spectrum texel is read once
        const unsigned gridp = (unsigned)(((size_t)c->n * c->n / 4 + 255) / 256);
        hipLaunchKernelGGL(k_propagate_paired, dim3(gridp), dim3(256), 0, s, (const c32*)c->h0, (const float*)c->omega,
                           c->field[OCEAN_FIELD_DY], c->field[OCEAN_FIELD_DX], c->field[OCEAN_FIELD_DZ], c->n, time, domain);
        for (int f = 0; f < 3; ++f) { c->nat_valid[f] = true; c->chk_valid[f] = false; c->step_b_pending[f] = false; }
        return;
    }
    const unsigned grid = (unsigned)(((size_t)c->n * c->n / 2 + 255) / 256);
    hipLaunchKernelGGL(k_propagate, dim3(grid), dim3(256), 0, s, (const c32*)c->h0, (const c32*)c->h0, (const float*)c->omega,
                       c->field[OCEAN_FIELD_DY], c->field[OCEAN_FIELD_DX], c->field[OCEAN_FIELD_DZ], c->n, 0, c->n, time,
                       domain, c->quirks);
    for (int f = 0; f < 3; ++f) { c->nat_valid[f] = true; c->chk_valid[f] = false; c->step_b_pending[f] = false; }
}
// Make the natural copy of field f current (no-op when it already is): run the deferred second step of a two-step column pass
// (N >= 8192), or unchunk (N <= 4096).
void settle_field(OceanContext* c, int f, hipStream_t s) {
    if (c->step_b_pending[f]) {                     // field[f] holds step A's result; step B's destination becomes the field
        OCEAN_DISPATCH(c->n, L::cols4_b((const c32*)c->field[f], c->field_alt[f], s));
        std::swap(c->field[f], c->field_alt[f]);
        c->step_b_pending[f] = false;
        return;
    }
    if (c->nat_valid[f]) return;
    hipLaunchKernelGGL(k_unchunk, dim3((unsigned)c->n / 4), dim3(256), 0, s, (const c32*)c->cfield[f], c->field[f], c->n, c->lay);
    c->nat_valid[f] = true;
}
void launch_correct(OceanContext* c, hipStream_t s) {
    const int dy = OCEAN_FIELD_DY, dx = OCEAN_FIELD_DX, dz = OCEAN_FIELD_DZ;
    if (c->chk_valid[dy] && c->chk_valid[dx] && c->chk_valid[dz]) {         // straight from the chunked fields
        hipLaunchKernelGGL(k_correct_chunked, dim3((unsigned)c->n / 4), dim3(256), 0, s, (const c32*)c->cfield[dy],
                           (const c32*)c->cfield[dx], (const c32*)c->cfield[dz], c->out, c->n, c->lay);
        return;
    }
    if (c->step_b_pending[dy] && c->step_b_pending[dx] && c->step_b_pending[dz]) {   // N >= 8192, behind the three column passes
        OCEAN_DISPATCH(c->n, L::cols4_b_correct(c, s));                          // (the fields stay as step A left them)
        return;
    }
    for (int f = 0; f < 3; ++f) settle_field(c, f, s);
    const unsigned grid = (unsigned)(((size_t)c->n * c->n / 2 + 255) / 256);
    hipLaunchKernelGGL(k_correct, dim3(grid), dim3(256), 0, s, (const c32*)c->field[dy], (const c32*)c->field[dx],
                       (const c32*)c->field[dz], c->out, c->n, 0, c->n);
}
// Row pass of field f (shader/fft_row.comp:44-63).  N <= 4096: natural rows in, chunked field out; else in place.
void launch_rows(OceanContext* c, int f, hipStream_t s) {
    settle_field(c, f, s);                       // a second row pass on a chunked field starts from its natural copy
    if (c->stage_chunked) {
        OCEAN_DISPATCH(c->n, L::stage_rows(c, f, s));
        c->nat_valid[f] = false;
        c->chk_valid[f] = true;
        return;
    }
    OCEAN_DISPATCH(c->n, L::rows(c, c->field[f], s));
    c->chk_valid[f] = false;
}
// Column pass of field f (shader/fft_col.comp:44-63): in place on whichever copy is current (whole 128-byte chunks
// when it follows the row pass; the natural-layout kernel otherwise, e.g. after ocean_write_field).
void launch_cols(OceanContext* c, int f, hipStream_t s) {
    if (c->stage_chunked && c->chk_valid[f]) {
        OCEAN_DISPATCH(c->n, L::stage_cols(c, f, s));
        c->nat_valid[f] = false;
        return;
    }
    bool cols4 = false;
    OCEAN_DISPATCH(c->n, cols4 = L::COLS4);
    if (cols4) {                                    // N >= 8192: two steps; the second waits for the field's next consumer
        settle_field(c, f, s);                      // (a column pass behind a column pass: the first one's second step)
        OCEAN_DISPATCH(c->n, L::cols4_a(c, c->field[f], s));
        c->step_b_pending[f] = true;
    } else OCEAN_DISPATCH(c->n, L::cols(c, c->field[f], s));
    c->chk_valid[f] = false;
}
// The normal field of the current map (shader/ocean.frag:50-66 at texel centres) into c->normals.
// From the RGBA map (any path: 16 + 16 B/texel) ...
void launch_normals_rgba(OceanContext* c, int channel, hipStream_t s, Timing t = Timing()) {
    const int rows = normals_rows(c->n);
    const dim3 grid((unsigned)((c->n / 256) * (c->n / rows))), b(256);
    const float4* rgba = c->out;
    switch (rows) {
        case 1: launch(k_normals<1>, grid, b, 0, s, t, rgba, c->normals, c->n, channel); break;
        case 2: launch(k_normals<2>, grid, b, 0, s, t, rgba, c->normals, c->n, channel); break;
        case 4: launch(k_normals<4>, grid, b, 0, s, t, rgba, c->normals, c->n, channel); break;
        default: launch(k_normals<8>, grid, b, 0, s, t, rgba, c->normals, c->n, channel); break;
    }
}
// ... or from the source-channel plane the fused pass 2 has just stored (4 + 16 B/texel).
void launch_normals_plane(OceanContext* c, hipStream_t s, Timing t = Timing()) {
    const int rows = normals_plane_rows(c->n);
    const dim3 grid((unsigned)((c->n / 256) * (c->n / rows) / 4)), b(256);
    const float* plane = c->plane;
    if (c->n >= 8192) {                              // the stores are the bound: one contiguous span of whole rows per workgroup
        launch(k_normals_plane_bands<NORMALS_BAND_ROWS>, dim3((unsigned)(c->n / NORMALS_BAND_ROWS)), b, 0, s, t, plane, c->normals, c->n);
        return;
    }
    switch (rows) {
        case 2: launch(k_normals_plane<2>, grid, b, 0, s, t, plane, c->normals, c->n); break;
        case 4: launch(k_normals_plane<4>, grid, b, 0, s, t, plane, c->normals, c->n); break;
        case 16: launch(k_normals_plane<16>, grid, b, 0, s, t, plane, c->normals, c->n); break;
        default: launch(k_normals_plane<8>, grid, b, 0, s, t, plane, c->normals, c->n); break;
    }
}
// The second buffer of field f for the two-step column pass (N >= 8192), allocated on first use.
int32_t reserve_cols(OceanContext* c, int f) {
    bool cols4 = false;
    OCEAN_DISPATCH(c->n, cols4 = L::COLS4);
    if (cols4 && !c->field_alt[f]) HIP_TRY(c, hipMalloc((void**)&c->field_alt[f], (size_t)c->n * c->n * sizeof(c32)));
    return OCEAN_OK;
}
void launch_frame(OceanContext* c, float time, float domain, hipStream_t s) {
    if (c->quirks != OCEAN_QUIRKS_REFERENCE) {      // the fused kernels implement the reference's arithmetic only
        launch_propagate(c, time, domain, s);
        for (int f = 0; f < 3; ++f) launch_rows(c, f, s);
        for (int f = 0; f < 3; ++f) launch_cols(c, f, s);
        launch_correct(c, s);
        if (c->frame_normals >= 0) launch_normals_rgba(c, c->frame_normals, s);
        return;
    }
    OCEAN_DISPATCH(c->n, { L::pass1(c, time, domain, s); L::pass2(c, s); });
    if (c->frame_normals >= 0) launch_normals_plane(c, s);
}

int32_t check_launch(OceanContext* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(c, e, what);
    return OCEAN_OK;
}

void release_import(OceanContext* c) {
    if (c->ext_mem) { (void)hipDestroyExternalMemory(c->ext_mem); c->ext_mem = nullptr; }
}
void free_all(OceanContext* c) {
    release_import(c);
    if (c->bands) {                                // the two sparse address ranges: unmap and release what backs them, then the ranges themselves
        for (auto& m : c->band_maps) { (void)hipMemUnmap(m.va, m.bytes); (void)hipMemRelease(m.handle); }
        c->band_maps.clear();
        if (c->h0T) (void)hipMemAddressFree(c->h0T, c->h0T_reserved);
        if (c->omegaT) (void)hipMemAddressFree(c->omegaT, c->omegaT_reserved);
        c->h0T = nullptr;
        c->omegaT = nullptr;
    }
    auto f = [](void* p) { if (p) (void)hipFree(p); };
    f(c->h0); f(c->omega); f(c->field[0]); f(c->field[1]); f(c->field[2]);
    f(c->cfield[0]); f(c->cfield[1]); f(c->cfield[2]); f(c->field_alt[0]); f(c->field_alt[1]); f(c->field_alt[2]);
    f(c->h0T); f(c->omegaT); f(c->inter); f(c->nyq); f(c->tw); f(c->out_own); f(c->normals); f(c->plane); f(c->batch_inter); f(c->batch_nyq); f(c->batch_out); f(c->batch_plane); f(c->batch_normals); f(c->positions); f(c->checksum_acc); f(c->inter_scale);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

}  // namespace

extern "C" {

int32_t ocean_abi_version(void) { return OCEAN_ABI_VERSION; }

int32_t ocean_device_count(void) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) { (void)hip_fail(nullptr, e, "hipGetDeviceCount"); return (e == hipErrorNoDevice) ? 0 : OCEAN_E_HIP; }
    return count;
}
int32_t ocean_device_pci_bus_id(int32_t device, char* out, int32_t capacity) {
    if (!out || capacity < 16) return fail(nullptr, OCEAN_E_INVALID_ARG, "bus id buffer of at least 16 bytes");
    const hipError_t e = hipDeviceGetPCIBusId(out, capacity, device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipDeviceGetPCIBusId");
    return OCEAN_OK;
}

int32_t ocean_context_create(int32_t device, int32_t resolution, OceanContext** out_ctx) {
    return ocean_context_create_ex(device, resolution, 0u, out_ctx);
}
static int32_t context_create(int32_t device, int32_t resolution, uint32_t flags, int32_t tiles, OceanContext** out_ctx, int32_t band_rank = -1,
                              int32_t band_world = 1);
int32_t ocean_context_create_tile_rank(int32_t device, int32_t resolution, int32_t rank, int32_t world, OceanContext** out_ctx) {
    if (!out_ctx) return fail(nullptr, OCEAN_E_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (!supported_n(resolution)) return fail(nullptr, OCEAN_E_UNSUPPORTED_N, "resolution must be a power of two in [256, 16384]");
    if (world < 1 || (world & (world - 1)) || resolution / world < 32 || rank < 0 || rank >= world)
        return fail(nullptr, OCEAN_E_INVALID_ARG, "world a power of two with at least 32 rows per rank, 0 <= rank < world");
    return context_create(device, resolution, OCEAN_CTX_TILE_RANK | OCEAN_CTX_TILE_BANDS, 1, out_ctx, rank, world);
}
int32_t ocean_context_create_ex(int32_t device, int32_t resolution, uint32_t flags, OceanContext** out_ctx) {
    return context_create(device, resolution, flags, 1, out_ctx);
}
int32_t ocean_context_create_tiles(int32_t device, int32_t resolution, int32_t tiles, OceanContext** out_ctx) {
    if (!out_ctx) return fail(nullptr, OCEAN_E_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    bool batched = false;
    OCEAN_DISPATCH(resolution, batched = L::BATCHED);
    if (!batched) return fail(nullptr, OCEAN_E_UNSUPPORTED_N, "several tiles per launch pair exist at N <= 1024 (above, one tile fills the chip: one context per tile)");
    if (tiles < 1 || tiles > OCEAN_BATCH_MAX) return fail(nullptr, OCEAN_E_INVALID_ARG, "tiles must be in [1, OCEAN_BATCH_MAX]");
    return context_create(device, resolution, OCEAN_CTX_FUSED_ONLY, tiles, out_ctx);
}
static int32_t context_create(int32_t device, int32_t resolution, uint32_t flags, int32_t tiles, OceanContext** out_ctx, int32_t band_rank, int32_t band_world) {
    if (!out_ctx) return fail(nullptr, OCEAN_E_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (flags & ~(OCEAN_CTX_FUSED_ONLY | OCEAN_CTX_TILE_RANK | ((band_rank >= 0) ? OCEAN_CTX_TILE_BANDS : 0u)))
        return fail(nullptr, OCEAN_E_INVALID_ARG, "unknown context flags (OCEAN_CTX_TILE_BANDS comes from ocean_context_create_tile_rank)");
    if (flags & OCEAN_CTX_TILE_RANK) flags |= OCEAN_CTX_FUSED_ONLY;
    if (!supported_n(resolution))
        return fail(nullptr, OCEAN_E_UNSUPPORTED_N, "resolution must be a power of two in [256, 16384]");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= count) return fail(nullptr, OCEAN_E_INVALID_ARG, "no such HIP device");
    OceanContext* c = new (std::nothrow) OceanContext();
    if (!c) return fail(nullptr, OCEAN_E_OOM, "host allocation failed");
    c->device = device;
    c->n = resolution;
    c->flags = flags;
    c->tiles = tiles;
    c->bands = band_rank >= 0;
    c->tile_rank = c->bands ? band_rank : 0;
    c->tile_world = c->bands ? band_world : 1;
    c->generation = g_generation.fetch_add(1);
    DeviceGuard guard(device);
    const size_t n2 = (size_t)resolution * resolution;
    {
        // Chunks are 4 x 4 complex (128 B); chunk (X, Y) at X*sx + Y*sy, +32 elements
        // (256 B) of padding per slab so that the strided side of the hand-off does not revisit one channel.
        // Pass-2-contiguous: the chunks of one chunk row are adjacent.
        int bshift = 0;                            // blocks of 2^bshift chunk rows (Geo::inter_bshift: 0 = pass-2-contiguous)
        OCEAN_DISPATCH(resolution, bshift = L::H::inter_bshift);
        // chunk (X, Y) at (Y / B) * sy + X * sx + (Y % B) * 16, B = 2^bshift (ocean_kernels.hpp InterLayout); +32
        // elements (256 B) per block slab so that the strided side of the hand-off does not revisit one channel
        auto make = [&](size_t columns, int bs) {
            const size_t gx = columns / CHUNK_W, gy = (size_t)resolution / CHUNK_R;
            while (((size_t)1 << bs) > gy) --bs;
            const size_t B = (size_t)1 << bs;
            InterLayout l{0, 0, 0, bs};
            l.sx = B * 16;
            l.sy = gx * l.sx + 32;
            l.fs = l.sy * (gy / B);
            return l;
        };
        c->lay = make((size_t)resolution, 0);                   // the staged path's chunked hand-off
        c->lay_h = make((size_t)resolution / 2, bshift);  // the fused frame's half-spectrum intermediate
    }
    bool mapping_bands = false;                    // inside the virtual-memory calls of a band-limited rank context
    auto bail = [&](hipError_t err, const char* what) {
        int32_t code = hip_fail(nullptr, err, what);
        // a runtime / device that cannot reserve and map address ranges (no virtual-memory management): its own status, so that a
        // caller can fall back to a full OCEAN_CTX_TILE_RANK context for THIS reason only (gfx_ocean_amd/sharded.py)
        if (mapping_bands && code != OCEAN_E_OOM) code = fail(nullptr, OCEAN_E_UNSUPPORTED, g_create_error + " [ocean_context_create_tile_rank: the sparse mapping of the input lines failed]");
        free_all(c);
        delete c;
        return code;
    };
#define CTX_TRY(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) return bail(e2_, #expr); } while (0)
    CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTX_TRY(hipEventCreate(&c->ev_a));
    CTX_TRY(hipEventCreate(&c->ev_b));
    CTX_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    // One allocation per buffer the path in use needs (the reference sizes one allocation for exactly what it binds,
    // src/render.rs:607-670): the natural-layout copies, fields and chunked hand-off of the staged path are 60 (36 at N >= 8192)
    // of a full context's 100 (76) bytes per texel -- not allocated with OCEAN_CTX_FUSED_ONLY; a rank of a sharded tile
    // (OCEAN_CTX_TILE_RANK) keeps the static inputs only: its intermediate and rows live in the caller's exchange buffers.
    const bool staged = !(flags & OCEAN_CTX_FUSED_ONLY), framed = !(flags & OCEAN_CTX_TILE_RANK);
    OCEAN_DISPATCH(resolution, c->stage_chunked = L::G::stage_chunked);
    if (staged) {
        CTX_TRY(hipMalloc((void**)&c->h0, n2 * sizeof(c32)));
        CTX_TRY(hipMalloc((void**)&c->omega, n2 * sizeof(float)));
        for (int f = 0; f < 3; ++f) CTX_TRY(hipMalloc((void**)&c->field[f], n2 * sizeof(c32)));
        if (c->stage_chunked)
            for (int f = 0; f < 3; ++f) CTX_TRY(hipMalloc((void**)&c->cfield[f], c->lay.fs * sizeof(c32)));
    }
    if (c->bands) {
        // Reserve the full ranges, back the bands: own-type lines a-1 .. b-1 and mirror-type lines N-b .. N-a (mod N) of the rank's
        // half-spectrum columns [a, b), on rank 0 also the Nyquist column's N/2-1 and N/2 (gfx_ocean_amd/sharded.py tile_rank_lines;
        // tests/test_sharded.py poisons every other line and gets the same bits), rounded to the 32-line tiles of the upload.
        const int nb = resolution / 32, a = c->tile_rank * (resolution / 2 / c->tile_world), b = a + resolution / 2 / c->tile_world;
        std::vector<char> need((size_t)nb, 0);
        auto mark = [&](int line) { need[(size_t)(((line % resolution) + resolution) % resolution) / 32] = 1; };
        for (int x = a - 1; x < b; ++x) mark(x);
        for (int x = resolution - b; x <= resolution - a; ++x) mark(x);
        if (c->tile_rank == 0) { mark(resolution / 2 - 1); mark(resolution / 2); }
        for (int i = 0; i < nb;) {
            if (!need[(size_t)i]) { ++i; continue; }
            int j = i;
            while (j < nb && need[(size_t)j]) ++j;
            c->band_blocks.push_back({i, j - i});
            i = j;
        }
        mapping_bands = true;
        hipMemAllocationProp prop;
        std::memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        // Mappings in units of 2 MiB (the runtime reports a granularity of 4 KiB, but hipMemSetAccess refuses some ranges that are
        // not aligned to the 2 MiB fragments it maps with -- measured): every band's byte range rounded outwards, overlaps merged,
        // per array.
        size_t gran = 0;
        CTX_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0) return bail(hipErrorNotSupported, "no allocation granularity");
        if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
        hipMemAccessDesc acc;
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        c->h0T_reserved = (n2 * sizeof(c32) + gran - 1) / gran * gran;
        c->omegaT_reserved = (n2 * sizeof(float) + gran - 1) / gran * gran;
        CTX_TRY(hipMemAddressReserve((void**)&c->h0T, c->h0T_reserved, gran, nullptr, 0));
        CTX_TRY(hipMemAddressReserve((void**)&c->omegaT, c->omegaT_reserved, gran, nullptr, 0));
        for (int which = 0; which < 2; ++which) {
            const size_t line_bytes = (size_t)resolution * (which ? sizeof(float) : sizeof(c32));
            char* base = which ? (char*)c->omegaT : (char*)c->h0T;
            size_t cur_lo = 0, cur_hi = 0;                          // the merged range being built
            auto flush = [&]() -> hipError_t {                      // one allocation per 2 MiB piece (a larger mapping that is aligned to
                for (size_t at = cur_lo; at < cur_hi; at += gran) { //  2 MiB only is refused by hipMemSetAccess as well -- measured)
                    OceanContext::Mapping m;
                    m.bytes = gran;
                    m.va = base + at;
                    hipError_t me = hipMemCreate(&m.handle, m.bytes, &prop, 0);
                    if (me != hipSuccess) return me;
                    me = hipMemMap(m.va, m.bytes, 0, m.handle, 0);
                    if (me != hipSuccess) { (void)hipMemRelease(m.handle); return me; }
                    c->band_maps.push_back(m);
                    me = hipMemSetAccess(m.va, m.bytes, &acc, 1);
                    if (me != hipSuccess) return me;
                }
                return hipSuccess;
            };
            for (const auto& br : c->band_blocks) {
                const size_t lo = (size_t)br.first * 32 * line_bytes / gran * gran;
                const size_t hi = ((size_t)(br.first + br.second) * 32 * line_bytes + gran - 1) / gran * gran;
                if (lo <= cur_hi && cur_hi != cur_lo) { cur_hi = hi > cur_hi ? hi : cur_hi; continue; }
                CTX_TRY(flush());
                cur_lo = lo;
                cur_hi = hi;
            }
            CTX_TRY(flush());
        }
        mapping_bands = false;
    } else {
        CTX_TRY(hipMalloc((void**)&c->h0T, (size_t)tiles * n2 * sizeof(c32)));      // (tile k's inputs at k * N * N elements)
        CTX_TRY(hipMalloc((void**)&c->omegaT, (size_t)tiles * n2 * sizeof(float)));
    }
    CTX_TRY(hipMalloc((void**)&c->nyq, 3 * (size_t)resolution * sizeof(c32)));
    if (framed) {
        CTX_TRY(hipMalloc((void**)&c->inter, 3 * c->lay_h.fs * sizeof(c32)));
        CTX_TRY(hipMalloc((void**)&c->out_own, n2 * sizeof(float4)));
    }
    CTX_TRY(hipMalloc((void**)&c->tw, (size_t)resolution * sizeof(c32)));
    c->out = c->out_own;
    {
        std::vector<c32> tw((size_t)resolution);
        for (int i = 0; i < resolution; ++i) {
            const double a = 2.0 * M_PI * (double)i / (double)resolution;
            tw[(size_t)i] = mk((float)std::cos(a), (float)std::sin(a));
        }
        CTX_TRY(hipMemcpy(c->tw, tw.data(), tw.size() * sizeof(c32), hipMemcpyHostToDevice));
    }
    {
        hipError_t pe = hipSuccess;
        OCEAN_DISPATCH(resolution, pe = L::prepare());
        if (pe != hipSuccess) return bail(pe, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
#undef CTX_TRY
    live_add(c);
    *out_ctx = c;
    return OCEAN_OK;
}

void ocean_context_destroy(OceanContext* ctx) {
    if (!valid(ctx)) return;
    live_remove(ctx);
    DeviceGuard guard(ctx->device);
    (void)sync_for_readback(ctx);       // frames put on a caller stream may still be using the buffers freed below
    free_all(ctx);
    ctx->magic = 0;
    delete ctx;
}

const char* ocean_last_error(const OceanContext* ctx) {
    if (valid(ctx)) return ctx->err.c_str();
    return g_create_error.c_str();
}

int32_t ocean_resolution(const OceanContext* ctx) { return valid(ctx) ? ctx->n : OCEAN_E_INVALID_ARG; }
uint32_t ocean_context_flags(const OceanContext* ctx) { return valid(ctx) ? ctx->flags : 0u; }

namespace {

// The one-time re-layout of a natural-layout spectrum in device memory for the fused path: h0T[x][y] = h0[y][x], omegaT likewise
// (k_transpose; fp16 storage: quantise, pack, and write the dequantised values back into the natural source).  `w` says which
// window of the natural arrays `h0_nat` / `om_nat` hold (the whole arrays, a slab of rows, a slab of rows of one band of columns);
// the destination lines [32 xb0, 32 (xb0 + xblocks)) are written for the window's `rows` rows.
void relayout_window(OceanContext* ctx, c32* h0_nat, const float* om_nat, NaturalWindow w, int rows, int xb0, int xblocks, bool f16, int scale_log2,
                     int32_t tile, hipStream_t s) {
    const size_t n = (size_t)ctx->n, n2 = n * n;
    const unsigned grid = (unsigned)(xblocks * (rows / 32));
    if (f16)
        hipLaunchKernelGGL(k_quantise_f16_transpose, dim3(grid), dim3(256), 0, s, reinterpret_cast<float2*>(h0_nat), w,
                           reinterpret_cast<uint32_t*>(ctx->h0T), (int)n, xb0, xblocks, std::ldexp(1.0f, scale_log2), std::ldexp(1.0f, -scale_log2));
    else
        hipLaunchKernelGGL(k_transpose<float2>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float2*>(h0_nat), w,
                           reinterpret_cast<float2*>(ctx->h0T) + (size_t)tile * n2, (int)n, xb0, xblocks);
    hipLaunchKernelGGL(k_transpose<float>, dim3(grid), dim3(256), 0, s, om_nat, w, ctx->omegaT + (size_t)tile * n2, (int)n, xb0, xblocks);
}
// ... of whole natural arrays in device memory: every line, or -- band-limited rank context -- the blocks of 32 lines that are
// backed by memory.
void relayout_spectrum(OceanContext* ctx, c32* h0_nat, const float* om_nat, bool f16, int scale_log2, int32_t tile, hipStream_t s) {
    const int nb = ctx->n / 32;
    const std::vector<std::pair<int, int>> blocks = ctx->bands ? ctx->band_blocks : std::vector<std::pair<int, int>>{{0, nb}};
    for (const auto& br : blocks) relayout_window(ctx, h0_nat, om_nat, NaturalWindow{ctx->n, 0, 0}, ctx->n, br.first, br.second, f16, scale_log2, tile, s);
}
void mark_uploaded(OceanContext* ctx, bool f16, int scale_log2, int32_t tile) {
    ctx->h0_f16 = f16;
    ctx->scale_log2 = scale_log2;
    ctx->uploaded_tiles |= (uint64_t)1 << tile;
    ctx->uploaded = ctx->uploaded_tiles == ((ctx->tiles >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << ctx->tiles) - 1));
}

// What a fused-only context stages of a host upload at a time (it has no natural-layout copies): slabs of whole rows of at most
// this many bytes, so that the upload's transient footprint does not depend on N (round 5 staged the whole tile: 3 GiB at 16384,
// on a rank context that otherwise holds 0.4).
constexpr size_t UPLOAD_STAGING_BYTES = (size_t)64 << 20;

int32_t upload_common(OceanContext* ctx, const float* h0_re_im, const float* omega, bool f16, int32_t tile = 0) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!h0_re_im || !omega) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL input");
    if (tile < 0 || tile >= ctx->tiles) return fail(ctx, OCEAN_E_INVALID_ARG, "no such tile in this context");
    if (f16 && ctx->tiles > 1) return fail(ctx, OCEAN_E_INVALID_ARG, "a context of several tiles stores fp32 spectra (one scale per context, not per tile)");
    if (f16 && ctx->bands) return fail(ctx, OCEAN_E_INVALID_ARG, "a band-limited rank context stores the fp32 spectrum (its bands are mapped for 8-byte texels)");
    DeviceGuard guard(ctx->device);
    const size_t n = (size_t)ctx->n, n2 = n * n;
    HIP_TRY(ctx, sync_for_readback(ctx));                         // frames in flight still read the old inputs
    int scale_log2 = 0;
    if (f16) {                                                     // max |component| * 2^s lands in [2^14, 2^15)
        float mx = 0.0f;
        for (size_t i = 0; i < 2 * n2; ++i) { const float a = std::fabs(h0_re_im[i]); if (a > mx) mx = a; }
        if (!(mx > 0.0f) || !std::isfinite(mx)) scale_log2 = 0;
        else scale_log2 = 14 - (int)std::floor(std::log2(mx));
    }
    hipStream_t s = ctx->stream;
    if (!(ctx->flags & OCEAN_CTX_FUSED_ONLY)) {
        // natural layout (the staged path; = the reference's initial_spec / omega_buffer), then the one-time re-layout for the
        // fused path on the device: h0T[x][y] = h0[y][x], omegaT likewise (k_transpose; round 2 did this on one host core:
        // 0.9 s at N = 8192, 2.2 s with the fp16 packing)
        HIP_TRY(ctx, hipMemcpy(ctx->h0, h0_re_im, n2 * sizeof(c32), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->omega, omega, n2 * sizeof(float), hipMemcpyHostToDevice));
        relayout_spectrum(ctx, ctx->h0, ctx->omega, f16, scale_log2, tile, s);
        { const int32_t st = check_launch(ctx, "upload re-layout launch"); if (st != OCEAN_OK) return st; }
        HIP_TRY(ctx, hipStreamSynchronize(s));
        mark_uploaded(ctx, f16, scale_log2, tile);
        return OCEAN_OK;
    }
    // A fused-only context has no natural-layout copies: the upload goes through a staging buffer that lives for this call, a
    // slab of whole rows at a time -- and, on a band-limited rank context, only the columns of those rows that become lines the
    // rank reads (hipMemcpy2D per band: 1 / world of the tile crosses PCIe and nothing else is staged).
    struct Staging {
        void* p = nullptr;
        ~Staging() { if (p) (void)hipFree(p); }
    } stage;
    const size_t row_bytes = n * (sizeof(c32) + sizeof(float));
    size_t slab_rows = UPLOAD_STAGING_BYTES / row_bytes / 32 * 32;
    if (slab_rows < 32) slab_rows = 32;
    if (slab_rows > n) slab_rows = n;
    HIP_TRY(ctx, hipMalloc(&stage.p, slab_rows * row_bytes));
    c32* h0_st = reinterpret_cast<c32*>(stage.p);
    float* om_st = reinterpret_cast<float*>(h0_st + slab_rows * n);
    const c32* h0_host = reinterpret_cast<const c32*>(h0_re_im);
    const std::vector<std::pair<int, int>> blocks = ctx->bands ? ctx->band_blocks : std::vector<std::pair<int, int>>{{0, (int)(n / 32)}};
    for (size_t y0 = 0; y0 < n; y0 += slab_rows) {
        const size_t rows = (n - y0 < slab_rows) ? n - y0 : slab_rows;
        for (const auto& br : blocks) {
            const size_t bx0 = (size_t)br.first * 32, bw = (size_t)br.second * 32;
            if (bw == n) {                                         // whole rows: one contiguous piece per array
                HIP_TRY(ctx, hipMemcpy(h0_st, h0_host + y0 * n, rows * n * sizeof(c32), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(om_st, omega + y0 * n, rows * n * sizeof(float), hipMemcpyHostToDevice));
            } else {
                HIP_TRY(ctx, hipMemcpy2D(h0_st, bw * sizeof(c32), h0_host + y0 * n + bx0, n * sizeof(c32), bw * sizeof(c32), rows, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy2D(om_st, bw * sizeof(float), omega + y0 * n + bx0, n * sizeof(float), bw * sizeof(float), rows, hipMemcpyHostToDevice));
            }
            relayout_window(ctx, h0_st, om_st, NaturalWindow{(int)bw, (int)bx0, (int)y0}, (int)rows, br.first, br.second, f16, scale_log2, tile, s);
            { const int32_t st = check_launch(ctx, "upload re-layout launch"); if (st != OCEAN_OK) return st; }
            HIP_TRY(ctx, hipStreamSynchronize(s));                 // the staging buffer is reused by the next piece
        }
    }
    mark_uploaded(ctx, f16, scale_log2, tile);
    return OCEAN_OK;
}

// The address a kernel on `device` reads `p` through: device memory of that device, managed memory, or host memory registered
// with the runtime and mapped for the device (a mapped staging buffer, as the reference's CPU_VISIBLE one: src/render.rs:749-761).
// NULL: pageable host memory, another device's memory, or registered memory without a device mapping.
const void* device_view(const void* p, int device) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (attr.type == hipMemoryTypeDevice) return (attr.device == device) ? p : nullptr;
    if (attr.type == hipMemoryTypeManaged) return p;
    if (attr.type == hipMemoryTypeHost) return attr.devicePointer;      // (NULL when the registration carries no device mapping)
    return nullptr;
}

}  // namespace

int32_t ocean_upload_spectrum_device(OceanContext* ctx, int32_t tile, const void* h0_device, const void* omega_device, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!h0_device || !omega_device) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL input");
    if (tile < 0 || tile >= ctx->tiles) return fail(ctx, OCEAN_E_INVALID_ARG, "no such tile in this context");
    DeviceGuard guard(ctx->device);
    const void* h0_src = device_view(h0_device, ctx->device);
    const void* om_src = device_view(omega_device, ctx->device);
    if (!h0_src || !om_src)
        return fail(ctx, OCEAN_E_INVALID_ARG, "ocean_upload_spectrum_device reads memory of the context's device, managed memory or registered host memory that is "
                                              "mapped for the device (pageable host memory: ocean_upload_spectrum)");
    const size_t n2 = (size_t)ctx->n * ctx->n;
    hipStream_t s = pick(ctx, stream);
    // The re-layout overwrites the inputs of frames that may still be queued on the context stream, and frames launched there
    // afterwards must find the new spectrum: when the upload runs on a caller stream it is ordered against the context stream
    // on both sides (events; nothing waits on the host).  Other caller streams are the caller's to order (include/ocean_hip.h).
    if (s != ctx->stream) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_order, 0));
    }
    if (!(ctx->flags & OCEAN_CTX_FUSED_ONLY)) {   // the staged path's natural-layout copies (= copy_buffer into initial_spec / omega_buffer)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h0, h0_src, n2 * sizeof(c32), hipMemcpyDefault, s));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->omega, om_src, n2 * sizeof(float), hipMemcpyDefault, s));
    }
    // (k_transpose only reads its source; the fp16 storage needs the maximum first: ocean_upload_spectrum_f16)
    relayout_spectrum(ctx, const_cast<c32*>(static_cast<const c32*>(h0_src)), static_cast<const float*>(om_src), false, 0, tile, s);
    { const int32_t st = check_launch(ctx, "upload re-layout launch"); if (st != OCEAN_OK) return st; }
    if (s != ctx->stream) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_order, s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_order, 0));
    }
    mark_uploaded(ctx, false, 0, tile);
    return OCEAN_OK;
}

int32_t ocean_upload_spectrum(OceanContext* ctx, const float* h0_re_im, const float* omega) {
    return upload_common(ctx, h0_re_im, omega, false);
}
int32_t ocean_upload_spectrum_tile(OceanContext* ctx, int32_t tile, const float* h0_re_im, const float* omega) {
    return upload_common(ctx, h0_re_im, omega, false, tile);
}
int32_t ocean_context_tiles(const OceanContext* ctx) { return valid(ctx) ? ctx->tiles : OCEAN_E_INVALID_ARG; }
int32_t ocean_upload_spectrum_f16(OceanContext* ctx, const float* h0_re_im, const float* omega) {
    return upload_common(ctx, h0_re_im, omega, true);
}
int32_t ocean_spectrum_scale_log2(const OceanContext* ctx) { return valid(ctx) ? ctx->scale_log2 : OCEAN_E_INVALID_ARG; }
int32_t ocean_read_spectrum(OceanContext* ctx, float* host_re_im) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_staged(ctx, "ocean_read_spectrum"));
    if (!host_re_im) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, hipMemcpy(host_re_im, ctx->h0, (size_t)ctx->n * ctx->n * sizeof(c32), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

// ---- stage objects ----------------------------------------------------------------------------
int32_t ocean_fft_init(OceanContext* ctx, OceanFft** out) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!out) return fail(ctx, OCEAN_E_INVALID_ARG, "out is NULL");
    OceanFft* f = new (std::nothrow) OceanFft();
    if (!f) return fail(ctx, OCEAN_E_OOM, "host allocation failed");
    f->ctx = ctx;
    f->generation = ctx->generation;
    live_add(f);
    *out = f;
    return OCEAN_OK;
}
void ocean_fft_destroy(OceanFft* fft) { if (live(fft) && fft->magic == MAGIC_FFT) { live_remove(fft); fft->magic = 0; delete fft; } }

int32_t ocean_propagation_init(OceanContext* ctx, OceanPropagation** out) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!out) return fail(ctx, OCEAN_E_INVALID_ARG, "out is NULL");
    OceanPropagation* p = new (std::nothrow) OceanPropagation();
    if (!p) return fail(ctx, OCEAN_E_OOM, "host allocation failed");
    p->ctx = ctx;
    p->generation = ctx->generation;
    live_add(p);
    *out = p;
    return OCEAN_OK;
}
void ocean_propagation_destroy(OceanPropagation* p) { if (live(p) && p->magic == MAGIC_PRO) { live_remove(p); p->magic = 0; delete p; } }

int32_t ocean_correction_init(OceanContext* ctx, OceanCorrection** out) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!out) return fail(ctx, OCEAN_E_INVALID_ARG, "out is NULL");
    OceanCorrection* c = new (std::nothrow) OceanCorrection();
    if (!c) return fail(ctx, OCEAN_E_OOM, "host allocation failed");
    c->ctx = ctx;
    c->generation = ctx->generation;
    live_add(c);
    *out = c;
    return OCEAN_OK;
}
void ocean_correction_destroy(OceanCorrection* c) { if (live(c) && c->magic == MAGIC_COR) { live_remove(c); c->magic = 0; delete c; } }

// ---- dispatches ---------------------------------------------------------------------------------
int32_t ocean_propagate(OceanPropagation* p, const OceanPropagateLocals* locals, void* stream) {
    if (!valid_stage(p, MAGIC_PRO)) return OCEAN_E_INVALID_ARG;
    OceanContext* c = p->ctx;
    NEED(need_staged(c, "ocean_propagate"));
    if (!locals) return fail(c, OCEAN_E_INVALID_ARG, "locals is NULL");
    if (locals->resolution != c->n) return fail(c, OCEAN_E_INVALID_ARG, "PropagateLocals.resolution != context resolution");
    if (!(locals->domain_size > 0.0f)) return fail(c, OCEAN_E_INVALID_ARG, "domain_size must be > 0");
    if (!c->uploaded) return fail(c, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    DeviceGuard guard(c->device);
    launch_propagate(c, locals->time, locals->domain_size, pick(c, stream));
    return check_launch(c, "k_propagate launch");
}

static int32_t fft_pass_common(OceanFft* fft, int32_t field, void* stream, bool cols) {
    if (!valid_stage(fft, MAGIC_FFT)) return OCEAN_E_INVALID_ARG;
    OceanContext* c = fft->ctx;
    NEED(need_staged(c, cols ? "ocean_fft_cols" : "ocean_fft_rows"));
    if (field != OCEAN_FIELD_ALL && (field < 0 || field > 2)) return fail(c, OCEAN_E_INVALID_ARG, "bad field selector");
    DeviceGuard guard(c->device);
    hipStream_t s = pick(c, stream);
    // reference order of the three descriptor sets: dx, dy, dz (src/render.rs:1158-1179)
    for (int f = 0; f < 3; ++f)
        if (field == OCEAN_FIELD_ALL || field == f) {
            if (cols) { NEED(reserve_cols(c, f)); launch_cols(c, f, s); }
            else launch_rows(c, f, s);
        }
    return check_launch(c, cols ? "k_fft_lines<COL> launch" : "k_fft_lines<ROW> launch");
}
int32_t ocean_fft_rows(OceanFft* fft, int32_t field, void* stream) { return fft_pass_common(fft, field, stream, false); }
int32_t ocean_fft_cols(OceanFft* fft, int32_t field, void* stream) { return fft_pass_common(fft, field, stream, true); }

int32_t ocean_correct(OceanCorrection* cor, const OceanCorrectionLocals* locals, void* stream) {
    if (!valid_stage(cor, MAGIC_COR)) return OCEAN_E_INVALID_ARG;
    OceanContext* c = cor->ctx;
    NEED(need_staged(c, "ocean_correct"));
    if (!locals) return fail(c, OCEAN_E_INVALID_ARG, "locals is NULL");
    if (locals->resolution != (uint32_t)c->n) return fail(c, OCEAN_E_INVALID_ARG, "CorrectionLocals.resolution != context resolution");
    DeviceGuard guard(c->device);
    launch_correct(c, pick(c, stream));
    return check_launch(c, "k_correct launch");
}

int32_t ocean_frame_ex(OceanContext* ctx, const OceanPropagateLocals* locals, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_frame"));
    if (!locals) return fail(ctx, OCEAN_E_INVALID_ARG, "locals is NULL");
    if (locals->resolution != ctx->n) return fail(ctx, OCEAN_E_INVALID_ARG, "PropagateLocals.resolution != context resolution");
    if (!(locals->domain_size > 0.0f)) return fail(ctx, OCEAN_E_INVALID_ARG, "domain_size must be > 0");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    DeviceGuard guard(ctx->device);
    launch_frame(ctx, locals->time, locals->domain_size, pick(ctx, stream));
    return check_launch(ctx, "ocean_frame launch");
}
int32_t ocean_set_quirks(OceanContext* ctx, uint32_t quirks) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (quirks & ~OCEAN_QUIRKS_REFERENCE) return fail(ctx, OCEAN_E_INVALID_ARG, "unknown quirk bits");
    if (quirks != OCEAN_QUIRKS_REFERENCE) {        // ocean_frame then runs the staged dispatches: their buffers must exist before a frame is launched
        NEED(need_staged(ctx, "ocean_set_quirks (the fused kernels implement the reference quirks only)"));
        DeviceGuard guard(ctx->device);
        for (int f = 0; f < 3; ++f) NEED(reserve_cols(ctx, f));
    }
    ctx->quirks = quirks;
    return OCEAN_OK;
}
uint32_t ocean_quirks(const OceanContext* ctx) { return valid(ctx) ? ctx->quirks : 0u; }

int32_t ocean_set_intermediate(OceanContext* ctx, int32_t mode) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (mode != OCEAN_INTER_F32 && mode != OCEAN_INTER_BFP16) return fail(ctx, OCEAN_E_INVALID_ARG, "unknown intermediate mode");
    if (mode == OCEAN_INTER_BFP16) {
        bool ok = false;
        OCEAN_DISPATCH(ctx->n, ok = L::I16_BUILT);
        if (!ok) return fail(ctx, OCEAN_E_UNSUPPORTED_N, "the 16-bit intermediate exists for the split-line kernels only (N = 8192, BASELINE config 5)");
        if (!ctx->inter_scale) {
            DeviceGuard guard(ctx->device);
            HIP_TRY(ctx, hipMalloc((void**)&ctx->inter_scale, (size_t)3 * (ctx->n / 64) * (ctx->n / 4) * sizeof(float)));
        }
    }
    ctx->inter16 = (mode == OCEAN_INTER_BFP16);
    return OCEAN_OK;
}
int32_t ocean_intermediate(const OceanContext* ctx) { return valid(ctx) ? (ctx->inter16 ? OCEAN_INTER_BFP16 : OCEAN_INTER_F32) : OCEAN_E_INVALID_ARG; }

int32_t ocean_frame(OceanContext* ctx, float time, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    OceanPropagateLocals l{time, ctx->n, ctx->default_domain};
    return ocean_frame_ex(ctx, &l, stream);
}

int32_t ocean_normals(OceanContext* ctx, int32_t source_channel, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_normals"));
    if (source_channel < 0 || source_channel > 2) return fail(ctx, OCEAN_E_INVALID_ARG, "source_channel must be 0, 1 or 2");
    DeviceGuard guard(ctx->device);
    const size_t n2 = (size_t)ctx->n * ctx->n;
    if (!ctx->normals) HIP_TRY(ctx, hipMalloc((void**)&ctx->normals, n2 * sizeof(float4)));
    launch_normals_rgba(ctx, source_channel, pick(ctx, stream));
    return check_launch(ctx, "k_normals launch");
}
int32_t ocean_set_frame_normals(OceanContext* ctx, int32_t source_channel) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_set_frame_normals"));
    if (source_channel < -1 || source_channel > 2) return fail(ctx, OCEAN_E_INVALID_ARG, "source_channel must be -1 (off), 0, 1 or 2");
    if (source_channel >= 0) {
        DeviceGuard guard(ctx->device);
        const size_t n2 = (size_t)ctx->n * ctx->n;
        if (!ctx->normals) HIP_TRY(ctx, hipMalloc((void**)&ctx->normals, n2 * sizeof(float4)));
        if (!ctx->plane) HIP_TRY(ctx, hipMalloc((void**)&ctx->plane, n2 * sizeof(float)));
    }
    ctx->frame_normals = source_channel;
    return OCEAN_OK;
}
int32_t ocean_frame_normals(const OceanContext* ctx) { return valid(ctx) ? ctx->frame_normals : OCEAN_E_INVALID_ARG; }
void* ocean_normals_device_ptr(OceanContext* ctx) { return valid(ctx) ? (void*)ctx->normals : nullptr; }
int32_t ocean_read_normals(OceanContext* ctx, float* host_xyz0) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!host_xyz0) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->normals) return fail(ctx, OCEAN_E_STATE, "ocean_normals has not been called");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    HIP_TRY(ctx, hipMemcpy(host_xyz0, ctx->normals, (size_t)ctx->n * ctx->n * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int32_t ocean_positions(OceanContext* ctx, int32_t verts, float offset_x, float offset_z, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_positions"));
    if (verts < 2 || verts > 16384) return fail(ctx, OCEAN_E_INVALID_ARG, "verts must be in [2, 16384]");
    DeviceGuard guard(ctx->device);
    const size_t nv = (size_t)verts * verts;
    if (ctx->position_verts != verts) {
        HIP_TRY(ctx, sync_for_readback(ctx));                     // an earlier ocean_positions (possibly on a caller stream) may still write the old buffer
        if (ctx->positions) (void)hipFree(ctx->positions);
        ctx->positions = nullptr;
        ctx->position_verts = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->positions, nv * sizeof(float4)));
        ctx->position_verts = verts;
    }
    hipLaunchKernelGGL(k_positions, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, pick(ctx, stream),
                       (const float4*)ctx->out, ctx->positions, ctx->n, verts, offset_x, offset_z);
    return check_launch(ctx, "k_positions launch");
}
int32_t ocean_read_positions(OceanContext* ctx, float* host_xyz1) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!host_xyz1) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->positions) return fail(ctx, OCEAN_E_STATE, "ocean_positions has not been called");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    HIP_TRY(ctx, hipMemcpy(host_xyz1, ctx->positions, (size_t)ctx->position_verts * ctx->position_verts * sizeof(float4),
                           hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int32_t ocean_sync(OceanContext* ctx) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));   // the context stream; the whole device once a dispatch ran on a caller stream
    return OCEAN_OK;
}

// ---- device-side consumers of the map: checksum (reproducibility tests), packed copies (final gather) ----------
int32_t ocean_checksum_displacement(OceanContext* ctx, void* stream, uint64_t* out_sum) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_checksum_displacement"));
    if (!out_sum) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    DeviceGuard guard(ctx->device);
    hipStream_t s = pick(ctx, stream);
    if (!ctx->checksum_acc) HIP_TRY(ctx, hipMalloc((void**)&ctx->checksum_acc, sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->checksum_acc, 0, sizeof(unsigned long long), s));
    const size_t vecs = (size_t)ctx->n * ctx->n;                     // one uint4 per RGBA32F texel
    const unsigned grid = (unsigned)((vecs / 256 < 2048) ? (vecs / 256) : 2048);
    hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(256), 0, s, (const uint4*)ctx->out, vecs, ctx->checksum_acc);
    { const int32_t st = check_launch(ctx, "k_checksum launch"); if (st != OCEAN_OK) return st; }
    unsigned long long host = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&host, ctx->checksum_acc, sizeof host, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *out_sum = (uint64_t)host;
    return OCEAN_OK;
}

int64_t ocean_packed_bytes(const OceanContext* ctx, int32_t format) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    const int64_t n2 = (int64_t)ctx->n * ctx->n;
    switch (format) {
        case OCEAN_PACK_RGBA32F: return n2 * 16;
        case OCEAN_PACK_RGB32F: return n2 * 12;
        case OCEAN_PACK_HEIGHT32F: return n2 * 4;
        default: return OCEAN_E_INVALID_ARG;
    }
}
int32_t ocean_pack_displacement(OceanContext* ctx, int32_t format, void* device_out, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_pack_displacement"));
    if (!device_out || (reinterpret_cast<uintptr_t>(device_out) & 15u))
        return fail(ctx, OCEAN_E_INVALID_ARG, "packed output must be a 16-byte aligned device pointer");
    if (format != OCEAN_PACK_RGBA32F && format != OCEAN_PACK_RGB32F && format != OCEAN_PACK_HEIGHT32F)
        return fail(ctx, OCEAN_E_INVALID_ARG, "unknown pack format");
    DeviceGuard guard(ctx->device);
    hipStream_t s = pick(ctx, stream);
    const size_t quads = (size_t)ctx->n * ctx->n / 4;
    const unsigned grid = (unsigned)((quads + 255) / 256);
    if (format == OCEAN_PACK_RGBA32F)
        HIP_TRY(ctx, hipMemcpyAsync(device_out, ctx->out, quads * 64, hipMemcpyDeviceToDevice, s));
    else if (format == OCEAN_PACK_RGB32F)
        hipLaunchKernelGGL(k_pack_rgb32f, dim3(grid), dim3(256), 0, s, (const float4*)ctx->out, (float4*)device_out, quads);
    else
        hipLaunchKernelGGL(k_pack_height32f, dim3(grid), dim3(256), 0, s, (const float4*)ctx->out, (float4*)device_out, quads);
    return check_launch(ctx, "k_pack launch");
}

// ---- readback / injection ---------------------------------------------------------------------
int32_t ocean_read_displacement(OceanContext* ctx, float* host_rgba) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_read_displacement"));
    if (!host_rgba) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    HIP_TRY(ctx, hipMemcpy(host_rgba, ctx->out, (size_t)ctx->n * ctx->n * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
int32_t ocean_read_field(OceanContext* ctx, int32_t field, float* host_re_im) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_staged(ctx, "ocean_read_field"));
    if (!host_re_im || field < 0 || field > 2) return fail(ctx, OCEAN_E_INVALID_ARG, "bad field or NULL output");
    DeviceGuard guard(ctx->device);
    // The field may live in the chunked hand-off layout: natural copy first.  The un-chunk runs on the context stream,
    // so whatever produced the chunked copy (possibly on a caller stream) has to be complete before it starts.
    if (ctx->foreign_stream) HIP_TRY(ctx, hipDeviceSynchronize());
    settle_field(ctx, field, ctx->stream);
    { const int32_t st = check_launch(ctx, "k_unchunk launch"); if (st != OCEAN_OK) return st; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(host_re_im, ctx->field[field], (size_t)ctx->n * ctx->n * sizeof(c32), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
int32_t ocean_write_field(OceanContext* ctx, int32_t field, const float* host_re_im) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_staged(ctx, "ocean_write_field"));
    if (!host_re_im || field < 0 || field > 2) return fail(ctx, OCEAN_E_INVALID_ARG, "bad field or NULL input");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    HIP_TRY(ctx, hipMemcpy(ctx->field[field], host_re_im, (size_t)ctx->n * ctx->n * sizeof(c32), hipMemcpyHostToDevice));
    ctx->nat_valid[field] = true;
    ctx->chk_valid[field] = false;
    ctx->step_b_pending[field] = false;
    return OCEAN_OK;
}

void* ocean_displacement_device_ptr(OceanContext* ctx) { return valid(ctx) ? (void*)ctx->out : nullptr; }
int32_t ocean_bind_displacement(OceanContext* ctx, void* device_rgba) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_bind_displacement"));
    if (device_rgba && (reinterpret_cast<uintptr_t>(device_rgba) & 15u))
        return fail(ctx, OCEAN_E_INVALID_ARG, "displacement buffer must be 16-byte aligned");
    if (ctx->ext_mem) {                                            // frames in flight may still write the imported allocation
        DeviceGuard guard(ctx->device);
        HIP_TRY(ctx, sync_for_readback(ctx));
        release_import(ctx);
    }
    ctx->out = device_rgba ? (float4*)device_rgba : ctx->out_own;
    return OCEAN_OK;
}
int32_t ocean_bind_displacement_fd(OceanContext* ctx, int32_t fd, uint64_t allocation_bytes, uint64_t offset_bytes) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_bind_displacement_fd"));
    const uint64_t map_bytes = (uint64_t)ctx->n * ctx->n * sizeof(float4);
    if (fd < 0 || (offset_bytes & 15u) || offset_bytes > allocation_bytes || allocation_bytes - offset_bytes < map_bytes)
        return fail(ctx, OCEAN_E_INVALID_ARG, "import: a valid descriptor, a 16-byte aligned offset, and N*N*16 bytes behind it inside the allocation");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    hipExternalMemoryHandleDesc d;
    std::memset(&d, 0, sizeof d);
    d.type = hipExternalMemoryHandleTypeOpaqueFd;                  // VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT / a HIP-exported POSIX descriptor
    d.handle.fd = fd;
    d.size = allocation_bytes;
    hipExternalMemory_t ext = nullptr;
    HIP_TRY(ctx, hipImportExternalMemory(&ext, &d));               // (on success the runtime owns the descriptor)
    hipExternalMemoryBufferDesc b;
    std::memset(&b, 0, sizeof b);
    b.offset = offset_bytes;
    b.size = map_bytes;
    void* mapped = nullptr;
    hipError_t e = hipExternalMemoryGetMappedBuffer(&mapped, ext, &b);
    if (e != hipSuccess || !mapped || (reinterpret_cast<uintptr_t>(mapped) & 15u)) {
        (void)hipDestroyExternalMemory(ext);
        return (e != hipSuccess) ? hip_fail(ctx, e, "hipExternalMemoryGetMappedBuffer") : fail(ctx, OCEAN_E_HIP, "the imported mapping is not 16-byte aligned");
    }
    release_import(ctx);
    ctx->ext_mem = ext;
    ctx->out = (float4*)mapped;
    return OCEAN_OK;
}
void* ocean_stream(OceanContext* ctx) { return valid(ctx) ? (void*)ctx->stream : nullptr; }

// ---- one tile over several GPUs, second generation (half-spectrum, fused; include/ocean_hip.h) ------------------------
static int32_t tile_check(OceanContext* ctx, int32_t rank, int32_t world, int32_t part, int32_t parts) {
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    if (ctx->quirks != OCEAN_QUIRKS_REFERENCE) return fail(ctx, OCEAN_E_STATE, "the fused kernels implement the reference quirks only (ocean_set_quirks)");
    bool ok = false;
    OCEAN_DISPATCH(ctx->n, ok = L::tile_supported(world, parts));
    if (ctx->bands && (rank != ctx->tile_rank || world != ctx->tile_world))
        return fail(ctx, OCEAN_E_INVALID_ARG, "this context holds the input lines of ONE rank of ONE world size (ocean_context_create_tile_rank): other ranks' lines are not mapped");
    if (!ok || rank < 0 || rank >= world || part < 0 || part >= parts)
        return fail(ctx, OCEAN_E_INVALID_ARG, "sharded tile: world and parts must be powers of two with at least 32 rows per rank and one "
                                              "column group per part, 0 <= rank < world, 0 <= part < parts");
    return OCEAN_OK;
}
int64_t ocean_tile_exchange_bytes(const OceanContext* ctx, int32_t world) {
    if (!valid(ctx) || world < 1 || (world & (world - 1)) || ctx->n / world < 32) return OCEAN_E_INVALID_ARG;
    return (int64_t)3 * (ctx->n / 2) * (int64_t)(ctx->n / world) * 8;          // per rank: send buffer = receive buffer
}
int32_t ocean_tile_pass1(OceanContext* ctx, const OceanPropagateLocals* locals, int32_t rank, int32_t world, int32_t part,
                         int32_t parts, void* send_device, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!locals || !send_device) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL argument");
    if (locals->resolution != ctx->n) return fail(ctx, OCEAN_E_INVALID_ARG, "PropagateLocals.resolution != context resolution");
    if (!(locals->domain_size > 0.0f)) return fail(ctx, OCEAN_E_INVALID_ARG, "domain_size must be > 0");
    if (reinterpret_cast<uintptr_t>(send_device) & 15u) return fail(ctx, OCEAN_E_INVALID_ARG, "send buffer must be 16-byte aligned");
    { const int32_t st = tile_check(ctx, rank, world, part, parts); if (st != OCEAN_OK) return st; }
    DeviceGuard guard(ctx->device);
    OCEAN_DISPATCH(ctx->n, L::tile_pass1(ctx, locals->time, locals->domain_size, rank, world, part, parts, (c32*)send_device, pick(ctx, stream)));
    return check_launch(ctx, "ocean_tile_pass1 launch");
}
int32_t ocean_tile_pass2(OceanContext* ctx, int32_t rank, int32_t world, int32_t parts, const void* recv_device, void* out_rows_device,
                         void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!recv_device || !out_rows_device) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL argument");
    if ((reinterpret_cast<uintptr_t>(recv_device) | reinterpret_cast<uintptr_t>(out_rows_device)) & 15u)
        return fail(ctx, OCEAN_E_INVALID_ARG, "buffers must be 16-byte aligned");
    { const int32_t st = tile_check(ctx, rank, world, 0, parts); if (st != OCEAN_OK) return st; }
    DeviceGuard guard(ctx->device);
    OCEAN_DISPATCH(ctx->n, L::tile_pass2(ctx, world, parts, (const c32*)recv_device, (float4*)out_rows_device, pick(ctx, stream)));
    return check_launch(ctx, "ocean_tile_pass2 launch");
}

// ---- K time steps of one tile per launch pair (the latency-bound sizes) -------------------------------------------------
namespace {
int32_t batch_check(OceanContext* ctx, int32_t count) {
    NEED(need_frame(ctx, "ocean_frame_batch"));
    if (count < 1 || count > OCEAN_BATCH_MAX) return fail(ctx, OCEAN_E_INVALID_ARG, "count must be in [1, OCEAN_BATCH_MAX]");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    if (ctx->quirks != OCEAN_QUIRKS_REFERENCE) return fail(ctx, OCEAN_E_STATE, "the fused kernels implement the reference quirks only (ocean_set_quirks)");
    if (ctx->frame_normals >= 0) {                                  // the batch carries the normal field at the sizes whose batches are one launch pair
        bool batched = false;
        OCEAN_DISPATCH(ctx->n, batched = L::BATCHED);
        if (!batched) return fail(ctx, OCEAN_E_STATE, "above N = 1024 a batch is K ordinary frames: call ocean_frame per frame for the normal field (or ocean_set_frame_normals(ctx, -1))");
    }
    return OCEAN_OK;
}
// the K intermediates / Nyquist scratches (and, without a caller buffer, the K maps) of a batch
int32_t batch_reserve(OceanContext* ctx, int32_t count, bool own_out) {
    bool batched = false;
    OCEAN_DISPATCH(ctx->n, batched = L::BATCHED);
    const size_t n2 = (size_t)ctx->n * ctx->n;
    if (batched && ctx->batch_cap < count) {
        HIP_TRY(ctx, sync_for_readback(ctx));
        if (ctx->batch_inter) (void)hipFree(ctx->batch_inter);
        if (ctx->batch_nyq) (void)hipFree(ctx->batch_nyq);
        ctx->batch_inter = nullptr; ctx->batch_nyq = nullptr; ctx->batch_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->batch_inter, (size_t)count * 3 * ctx->lay_h.fs * sizeof(c32)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->batch_nyq, (size_t)count * 3 * ctx->n * sizeof(c32)));
        ctx->batch_cap = count;
    }
    if (batched && ctx->frame_normals >= 0 && ctx->batch_normals_cap < count) {
        HIP_TRY(ctx, sync_for_readback(ctx));
        if (ctx->batch_plane) (void)hipFree(ctx->batch_plane);
        if (ctx->batch_normals) (void)hipFree(ctx->batch_normals);
        ctx->batch_plane = nullptr; ctx->batch_normals = nullptr; ctx->batch_normals_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->batch_plane, (size_t)count * n2 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->batch_normals, (size_t)count * n2 * sizeof(float4)));
        ctx->batch_normals_cap = count;
    }
    if (own_out && ctx->batch_out_cap < count) {
        HIP_TRY(ctx, sync_for_readback(ctx));
        if (ctx->batch_out) (void)hipFree(ctx->batch_out);
        ctx->batch_out = nullptr; ctx->batch_out_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->batch_out, (size_t)count * n2 * sizeof(float4)));
        ctx->batch_out_cap = count;
    }
    return OCEAN_OK;
}
// `tiles`: the K frames are K different tiles at the same time (ocean_frame_tiles) instead of K time steps of tile 0.
void launch_batch(OceanContext* c, float t0, float dt, int32_t count, float4* out, size_t out_stride_texels, hipStream_t s, bool tiles = false) {
    bool batched = false;
    OCEAN_DISPATCH(c->n, batched = L::BATCHED);
    if (batched) {
        FrameBatch b;
        b.dt = dt;
        b.inter_stride = (uint32_t)(3 * c->lay_h.fs);
        b.out_stride = out_stride_texels;
        if (tiles) {
            b.spec_stride_bytes = (size_t)c->n * c->n * sizeof(c32);
            b.omega_stride = (uint32_t)((size_t)c->n * c->n);
        }
        OCEAN_DISPATCH(c->n, {
            L::pass1_on(c, t0, c->default_domain, c->batch_inter, c->lay_h, L::H::half_grid1, 0, s, Timing(), c->batch_nyq, b, count);
            L::pass2_on(c, c->batch_inter, out, c->batch_plane, s, Timing(), b, count);
        });
        if (c->frame_normals >= 0) {                                // K normal fields from the K planes, one launch (blockIdx.y = frame)
            const int rows = normals_plane_rows(c->n);
            const dim3 grid((unsigned)((c->n / 256) * (c->n / rows) / 4), (unsigned)count), blk(256);
            const float* pl = c->batch_plane;
            switch (rows) {
                case 2: hipLaunchKernelGGL(k_normals_plane<2>, grid, blk, 0, s, pl, c->batch_normals, c->n); break;
                default: hipLaunchKernelGGL(k_normals_plane<4>, grid, blk, 0, s, pl, c->batch_normals, c->n); break;   // (N = 1024; batches exist up to there)
            }
        }
        return;
    }
    for (int i = 0; i < count; ++i)                                 // N > 1024: one frame fills the chip; the same frames, one launch pair each
        OCEAN_DISPATCH(c->n, { L::pass1(c, t0 + dt * (float)i, c->default_domain, s); L::pass2_on(c, c->inter, out + (size_t)i * out_stride_texels, c->plane, s, Timing()); });
}
// what ocean_read_batch_* may hand out afterwards
void batch_launched(OceanContext* c, int32_t count, bool own_out) {
    bool batched = false;
    OCEAN_DISPATCH(c->n, batched = L::BATCHED);
    if (own_out) c->last_batch_maps = count;                         // (a batch into caller memory leaves the library-owned maps as they were)
    c->last_batch_normals = (batched && c->frame_normals >= 0) ? count : 0;
}
}  // namespace

int32_t ocean_frame_batch(OceanContext* ctx, float t0, float dt, int32_t count, void* out_base_device, int64_t out_stride_bytes, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    { const int32_t st = batch_check(ctx, count); if (st != OCEAN_OK) return st; }
    const int64_t map_bytes = (int64_t)ctx->n * ctx->n * 16;
    if (out_base_device) {
        if ((reinterpret_cast<uintptr_t>(out_base_device) & 15u) || (out_stride_bytes & 15) || out_stride_bytes < map_bytes)
            return fail(ctx, OCEAN_E_INVALID_ARG, "batch output: 16-byte aligned base, stride a multiple of 16 and >= N*N*16");
    } else out_stride_bytes = map_bytes;
    DeviceGuard guard(ctx->device);
    { const int32_t st = batch_reserve(ctx, count, out_base_device == nullptr); if (st != OCEAN_OK) return st; }
    float4* out = out_base_device ? (float4*)out_base_device : ctx->batch_out;
    launch_batch(ctx, t0, dt, count, out, (size_t)(out_stride_bytes / 16), pick(ctx, stream));
    batch_launched(ctx, count, out_base_device == nullptr);
    return check_launch(ctx, "ocean_frame_batch launch");
}
int32_t ocean_frame_tiles(OceanContext* ctx, float time, void* out_base_device, int64_t out_stride_bytes, void* stream) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    { const int32_t st = batch_check(ctx, ctx->tiles); if (st != OCEAN_OK) return st; }
    const int64_t map_bytes = (int64_t)ctx->n * ctx->n * 16;
    if (out_base_device) {
        if ((reinterpret_cast<uintptr_t>(out_base_device) & 15u) || (out_stride_bytes & 15) || out_stride_bytes < map_bytes)
            return fail(ctx, OCEAN_E_INVALID_ARG, "tile maps: 16-byte aligned base, stride a multiple of 16 and >= N*N*16");
    } else out_stride_bytes = map_bytes;
    DeviceGuard guard(ctx->device);
    { const int32_t st = batch_reserve(ctx, ctx->tiles, out_base_device == nullptr); if (st != OCEAN_OK) return st; }
    float4* out = out_base_device ? (float4*)out_base_device : ctx->batch_out;
    launch_batch(ctx, time, 0.0f, ctx->tiles, out, (size_t)(out_stride_bytes / 16), pick(ctx, stream), true);
    batch_launched(ctx, ctx->tiles, out_base_device == nullptr);
    return check_launch(ctx, "ocean_frame_tiles launch");
}
void* ocean_batch_device_ptr(OceanContext* ctx) { return valid(ctx) ? (void*)ctx->batch_out : nullptr; }
void* ocean_batch_normals_device_ptr(OceanContext* ctx) { return valid(ctx) ? (void*)ctx->batch_normals : nullptr; }
int32_t ocean_read_batch_normals(OceanContext* ctx, int32_t index, float* host_xyz0) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!host_xyz0) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->batch_normals || index < 0 || index >= ctx->last_batch_normals)
        return fail(ctx, OCEAN_E_STATE, "the last batch left no normal field with this index (its frame count; a batch launched with ocean_set_frame_normals on)");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    const size_t n2 = (size_t)ctx->n * ctx->n;
    HIP_TRY(ctx, hipMemcpy(host_xyz0, ctx->batch_normals + (size_t)index * n2, n2 * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
int32_t ocean_read_batch_displacement(OceanContext* ctx, int32_t index, float* host_rgba) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (!host_rgba) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->batch_out || index < 0 || index >= ctx->last_batch_maps)
        return fail(ctx, OCEAN_E_STATE, "the last batch left no library-owned map with this index (its frame count; ocean_frame_batch / ocean_frame_tiles with out_base_device = NULL)");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, sync_for_readback(ctx));
    const size_t n2 = (size_t)ctx->n * ctx->n;
    HIP_TRY(ctx, hipMemcpy(host_rgba, ctx->batch_out + (size_t)index * n2, n2 * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
// `launches` batches of `count` frames back to back on the context stream between two events (library-owned maps): *out_ms.
int32_t ocean_time_frame_batch(OceanContext* ctx, int32_t launches, int32_t count, float t0, float dt, float* out_ms) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    if (launches < 1 || launches > 65536 || !out_ms) return fail(ctx, OCEAN_E_INVALID_ARG, "launches in [1, 65536], out_ms non-NULL");
    { const int32_t st = batch_check(ctx, count); if (st != OCEAN_OK) return st; }
    if (ctx->tiles > 1 && count != ctx->tiles) return fail(ctx, OCEAN_E_INVALID_ARG, "a context of K tiles is timed with count = K");
    DeviceGuard guard(ctx->device);
    { const int32_t st = batch_reserve(ctx, count, true); if (st != OCEAN_OK) return st; }
    const size_t n2 = (size_t)ctx->n * ctx->n;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    const bool tiles = ctx->tiles > 1;                             // a context of several tiles: every launch pair = one frame of each tile
    for (int i = 0; i < launches; ++i)
        launch_batch(ctx, tiles ? t0 + dt * (float)i : t0 + dt * (float)((int64_t)i * count), tiles ? 0.0f : dt, count, ctx->batch_out, n2, ctx->stream, tiles);
    batch_launched(ctx, count, true);
    HIP_TRY(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_b));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev_a, ctx->ev_b));
    return check_launch(ctx, "ocean_time_frame_batch");
}

// ---- measurement ----------------------------------------------------------------------------------
int32_t ocean_time_frames(OceanContext* ctx, int32_t frames, float t0, float dt, float* out_ms) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_time_frames"));
    if (frames <= 0 || !out_ms) return fail(ctx, OCEAN_E_INVALID_ARG, "frames must be > 0 and out_ms non-NULL");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    DeviceGuard guard(ctx->device);
    hipEvent_t a = ctx->ev_a, b = ctx->ev_b;
    HIP_TRY(ctx, hipEventRecord(a, ctx->stream));
    for (int i = 0; i < frames; ++i) launch_frame(ctx, t0 + dt * (float)i, ctx->default_domain, ctx->stream);
    HIP_TRY(ctx, hipEventRecord(b, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(b));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, a, b));
    return check_launch(ctx, "ocean_time_frames");
}

static int32_t profile_common(OceanContext* ctx, float time, int32_t cap, const char** names, float* ms,
                              int32_t* out_n, bool staged) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_profile_*"));
    if (staged) {
        NEED(need_staged(ctx, "ocean_profile_staged"));
        DeviceGuard guard0(ctx->device);
        for (int f = 0; f < 3; ++f) NEED(reserve_cols(ctx, f));
    }
    if (!names || !ms || !out_n) return fail(ctx, OCEAN_E_INVALID_ARG, "NULL output");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    static const char* kNatural[8] = {"k_propagate", "k_fft_lines<ROW> dx", "k_fft_lines<ROW> dy", "k_fft_lines<ROW> dz",
                                      "k_fft_lines<COL> dx", "k_fft_lines<COL> dy", "k_fft_lines<COL> dz", "k_correct"};
    static const char* kChunked[8] = {"k_propagate", "k_stage_rows(fft) dx", "k_stage_rows(fft) dy", "k_stage_rows(fft) dz",
                                      "k_stage_cols(fft) dx", "k_stage_cols(fft) dy", "k_stage_cols(fft) dz", "k_correct_chunked"};
    // N >= 8192: the column pass of a field is two steps (k_cols4_a: sub-transforms of sixteen columns, in place; k_cols4_b: the
    // S-point step over consecutive rows into the field's second buffer) -- 0.43-0.47 ms per field at 8192 where whole columns two
    // at a time (k_fft_lines<COL>, 16-byte pieces) took 0.85; staged frame 4.2 -> 2.8 ms (r05_run12/13).  The second step is deferred
    // to the field's consumer: behind the three column passes that is the correction, and k_cols4_b_correct does both.  The staged
    // calls remain the 1:1 compatibility path, ocean_frame (0.75 ms) the product.
    static const char* kTwoStep[8] = {"k_propagate", "k_fft_lines<ROW> dx", "k_fft_lines<ROW> dy", "k_fft_lines<ROW> dz",
                                      "k_cols4_a dx [two-step column pass, step 1]", "k_cols4_a dy", "k_cols4_a dz",
                                      "k_cols4_b_correct [step 2 of the three column passes + correction]"};
    const char* const* kStaged = ctx->stage_chunked ? kChunked : (ctx->n >= 8192 ? kTwoStep : kNatural);
    const char* kFused[2] = {"k_half_pass1", "k_half_pass2"};
    const int count = staged ? 8 : 2;
    if (cap < count) return fail(ctx, OCEAN_E_INVALID_ARG, "capacity too small");
    if (!staged && ctx->quirks != OCEAN_QUIRKS_REFERENCE)
        return fail(ctx, OCEAN_E_STATE, "the fused kernels implement the reference quirks only (ocean_set_quirks)");
    DeviceGuard guard(ctx->device);
    hipStream_t s = ctx->stream;
    struct Events {                                // destroyed on every exit path (HIP_TRY returns early)
        hipEvent_t e[13] = {};
        int n = 0;
        hipError_t add(int count) {
            for (int i = 0; i < count; ++i) { hipError_t r = hipEventCreate(&e[n]); if (r != hipSuccess) return r; ++n; }
            return hipSuccess;
        }
        ~Events() { for (int i = 0; i < n; ++i) (void)hipEventDestroy(e[i]); }
    } bag;
    HIP_TRY(ctx, bag.add(count + 1 + (staged ? 0 : 4)));
    hipEvent_t* ev = bag.e;                        // count + 1 stream events
    hipEvent_t* kev = bag.e + count + 1;           // fused: begin/end of the two dispatches
    HIP_TRY(ctx, hipEventRecord(ev[0], s));
    if (staged) {
        launch_propagate(ctx, time, ctx->default_domain, s);
        HIP_TRY(ctx, hipEventRecord(ev[1], s));
        for (int f = 0; f < 3; ++f) { launch_rows(ctx, f, s); HIP_TRY(ctx, hipEventRecord(ev[2 + f], s)); }
        for (int f = 0; f < 3; ++f) { launch_cols(ctx, f, s); HIP_TRY(ctx, hipEventRecord(ev[5 + f], s)); }
        launch_correct(ctx, s);
        HIP_TRY(ctx, hipEventRecord(ev[8], s));
    } else {
        // the fused kernels are timed by events bound to their own dispatches (see launch())
        // behind two untimed frames, so that the timed one runs in the steady state of a frame loop
        for (int w = 0; w < 2; ++w) launch_frame(ctx, time, ctx->default_domain, s);
        OCEAN_DISPATCH(ctx->n, L::pass1(ctx, time, ctx->default_domain, s, Timing{kev[0], kev[1]}));
        OCEAN_DISPATCH(ctx->n, L::pass2(ctx, s, Timing{kev[2], kev[3]}));
        HIP_TRY(ctx, hipEventRecord(ev[2], s));
    }
    HIP_TRY(ctx, hipEventSynchronize(ev[count]));
    for (int i = 0; i < count; ++i) {
        names[i] = staged ? kStaged[i] : kFused[i];
        if (staged) HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
        else HIP_TRY(ctx, hipEventElapsedTime(&ms[i], kev[2 * i], kev[2 * i + 1]));
    }
    *out_n = count;
    return check_launch(ctx, "profile");
}
// `batches` x `frames_per_batch` plain frames back to back on the context stream, one stream event between batches and a
// single sync at the end: batch_ms[b] = the b-th batch.  What a frame costs in an undisturbed loop, as a distribution
// (an event per FRAME adds ~5 % of gaps: measured r04_run3, 193 against 185.5 us at N = 4096; one per 10 frames does not).
int32_t ocean_time_frame_batches(OceanContext* ctx, int32_t batches, int32_t frames_per_batch, float t0, float dt, float* batch_ms) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_time_frame_batches"));
    if (batches <= 0 || batches > 4096 || frames_per_batch <= 0 || frames_per_batch > 4096 || !batch_ms)
        return fail(ctx, OCEAN_E_INVALID_ARG, "batches and frames_per_batch in [1, 4096], batch_ms non-NULL");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    DeviceGuard guard(ctx->device);
    struct Events {
        std::vector<hipEvent_t> e;
        ~Events() { for (hipEvent_t x : e) (void)hipEventDestroy(x); }
    } bag;
    for (int i = 0; i <= batches; ++i) { hipEvent_t x; HIP_TRY(ctx, hipEventCreate(&x)); bag.e.push_back(x); }
    HIP_TRY(ctx, hipEventRecord(bag.e[0], ctx->stream));
    for (int b = 0; b < batches; ++b) {
        for (int i = 0; i < frames_per_batch; ++i) launch_frame(ctx, t0 + dt * (float)((int64_t)b * frames_per_batch + i), ctx->default_domain, ctx->stream);
        HIP_TRY(ctx, hipEventRecord(bag.e[b + 1], ctx->stream));
    }
    HIP_TRY(ctx, hipEventSynchronize(bag.e[batches]));
    for (int b = 0; b < batches; ++b) HIP_TRY(ctx, hipEventElapsedTime(&batch_ms[b], bag.e[b], bag.e[b + 1]));
    return check_launch(ctx, "ocean_time_frame_batches");
}
// Per-frame times of a back-to-back frame loop ("median + p10/p90"): every dispatch carries its own begin/end
// events (hipExtLaunchKernelGGL: the kernel's timestamps, no extra packets on the stream), read after one sync.
//   pass1_ms[i], pass2_ms[i]: the two kernels of frame i;   period_ms[i] = begin(pass 1 of frame i + 1) - begin(pass 1 of
//   frame i), the last entry begin(pass 1) -> end(pass 2) -- in THIS loop, whose event-carrying launches leave larger gaps
//   than plain ones (ocean_time_frame_batches is the undisturbed loop).
int32_t ocean_frame_times_ex(OceanContext* ctx, int32_t frames, float t0, float dt, float* pass1_ms, float* pass2_ms, float* normals_ms,
                             float* period_ms) {
    if (!valid(ctx)) return OCEAN_E_INVALID_ARG;
    NEED(need_frame(ctx, "ocean_frame_times"));
    if (frames <= 0 || frames > 4096) return fail(ctx, OCEAN_E_INVALID_ARG, "frames must be in [1, 4096]");
    if (!ctx->uploaded) return fail(ctx, OCEAN_E_STATE, "ocean_upload_spectrum has not been called");
    if (ctx->quirks != OCEAN_QUIRKS_REFERENCE)
        return fail(ctx, OCEAN_E_STATE, "the fused kernels implement the reference quirks only (ocean_set_quirks)");
    const bool nrm = ctx->frame_normals >= 0;
    if (normals_ms && !nrm) return fail(ctx, OCEAN_E_STATE, "normals_ms asked for, but the frame carries no normal field (ocean_set_frame_normals)");
    DeviceGuard guard(ctx->device);
    struct Events {
        std::vector<hipEvent_t> e;
        hipError_t add(size_t count) {
            for (size_t i = 0; i < count; ++i) { hipEvent_t x; hipError_t r = hipEventCreate(&x); if (r != hipSuccess) return r; e.push_back(x); }
            return hipSuccess;
        }
        ~Events() { for (hipEvent_t x : e) (void)hipEventDestroy(x); }
    } bag;
    const size_t per = nrm ? 6 : 4;                                 // begin / end of every dispatch of a frame
    HIP_TRY(ctx, bag.add((size_t)frames * per));
    hipStream_t s = ctx->stream;
    for (int i = 0; i < frames; ++i) {
        hipEvent_t* k = bag.e.data() + (size_t)i * per;
        const float time = t0 + dt * (float)i;
        OCEAN_DISPATCH(ctx->n, L::pass1(ctx, time, ctx->default_domain, s, Timing{k[0], k[1]}));
        OCEAN_DISPATCH(ctx->n, L::pass2(ctx, s, Timing{k[2], k[3]}));
        if (nrm) launch_normals_plane(ctx, s, Timing{k[4], k[5]});
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < frames; ++i) {
        hipEvent_t* k = bag.e.data() + (size_t)i * per;
        if (pass1_ms) HIP_TRY(ctx, hipEventElapsedTime(&pass1_ms[i], k[0], k[1]));
        if (pass2_ms) HIP_TRY(ctx, hipEventElapsedTime(&pass2_ms[i], k[2], k[3]));
        if (normals_ms) HIP_TRY(ctx, hipEventElapsedTime(&normals_ms[i], k[4], k[5]));
        if (period_ms) HIP_TRY(ctx, hipEventElapsedTime(&period_ms[i], k[0], (i + 1 < frames) ? k[per] : k[per - 1]));
    }
    return check_launch(ctx, "ocean_frame_times");
}
int32_t ocean_frame_times(OceanContext* ctx, int32_t frames, float t0, float dt, float* pass1_ms, float* pass2_ms, float* period_ms) {
    return ocean_frame_times_ex(ctx, frames, t0, dt, pass1_ms, pass2_ms, nullptr, period_ms);
}
int32_t ocean_profile_frame(OceanContext* ctx, float time, int32_t cap, const char** names, float* ms, int32_t* out_n) {
    return profile_common(ctx, time, cap, names, ms, out_n, false);
}
int32_t ocean_profile_staged(OceanContext* ctx, float time, int32_t cap, const char** names, float* ms, int32_t* out_n) {
    return profile_common(ctx, time, cap, names, ms, out_n, true);
}

}  // extern "C"

#include "ocean_shard.hip"   // the sharded-tile entry points (same translation unit, same kernels)
