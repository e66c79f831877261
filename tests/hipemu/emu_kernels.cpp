// Runs the product's kernel templates on the CPU (one std::thread per GPU thread, std::barrier for
// __syncthreads) so that the non-GPU test tier can check them against the oracle.
// Build: g++ -std=c++20 -O1 -pthread -shared -fPIC -I tests/hipemu -I gfx_ocean_amd/csrc ...
#include <barrier>
#include <functional>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>

thread_local emu_dim3 threadIdx;
thread_local emu_dim3 blockIdx;
emu_dim3 blockDim;
emu_dim3 gridDim;
static std::barrier<>* g_barrier = nullptr;
void __syncthreads() { g_barrier->arrive_and_wait(); }

namespace ocean { alignas(16) unsigned char smem[160 * 1024]; float g_emu_wave_scratch[16][64]; }

#include "ocean_kernels.hpp"
#include "ocean_staged_kernels.hpp"

template <class F>
static void emu_launch(int grid, int threads, F&& body, int grid_y = 1) {
    std::barrier<> bar(threads);
    g_barrier = &bar;
    gridDim.x = (unsigned)grid;
    gridDim.y = (unsigned)grid_y;
    blockDim.x = (unsigned)threads;
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t] {
            threadIdx.x = (unsigned)t;
            for (int b = 0; b < grid * grid_y; ++b) {
                blockIdx.x = (unsigned)(b % grid);
                blockIdx.y = (unsigned)(b / grid);
                body();
                bar.arrive_and_wait();   // the next block reuses the LDS buffer
            }
        });
    }
    for (auto& th : pool) th.join();
    g_barrier = nullptr;
}

using namespace ocean;

template <int N> static int run_fft_lines(int col, c32* data, const c32* tw) {
    using G = Geo<N>;
    if (col) emu_launch(G::col_grid, G::col_threads, [&] { k_fft_lines<N, G::E, G::COL_LPW, true>(data, tw); });
    else emu_launch(G::row_grid, G::row_threads, [&] { k_fft_lines<N, G::E, G::ROW_LPW, false>(data, tw); });
    return 0;
}
// (pass 2 always runs as its PLANE instance here: the map must not depend on it, and emu_plane checks the plane)
static float* plane = nullptr;
static int plane_channel = 0;
// ... and as a batch of `batch_count` time steps in one launch pair when emu_set_batch says so (ocean_frame_batch, N <= 1024)
static int batch_count = 1;
static FrameBatch batch{};
template <int N, int PSEL> static int run_half_p(const void* h0T, int f16, float descale, const float* omT, c32* inter, c32* nyq,
                                                 float4* out, const c32* tw, InterLayout lay, float time, float L) {
    using G = Geo<N, PSEL>;
    if (batch_count > 1 && !batched_launches<N>) return -8;
    if constexpr (G::fpar) {
        if (batch_count > 1) {                          // as Launch<N>::pass1_batch_kernel of csrc/ocean_api.hip: the throughput geometry, unshared normalisation
            using GB = Geo<N, 2, true>;
            if (f16) emu_launch(GB::half_grid1, GB::half_threads1,
                                [&] { k_half_pass1<N, GB::E1, GB::P, true, GB::dma, GB::fpar, false>(h0T, descale, omT, inter, nyq, tw, lay, time, L, 0, batch); }, batch_count);
            else emu_launch(GB::half_grid1, GB::half_threads1,
                            [&] { k_half_pass1<N, GB::E1, GB::P, false, GB::dma, GB::fpar, false>(h0T, 1.0f, omT, inter, nyq, tw, lay, time, L, 0, batch); }, batch_count);
            emu_launch(G::half_grid2, G::half_threads2,
                       [&] { k_half_pass2<N, G::E2, CHUNK_W, G::R2h, G::p2_group, G::ppar, false, true>(inter, out, tw, lay, plane, plane_channel, batch); }, batch_count);
            return 0;
        }
    }
    if (f16) emu_launch(G::half_grid1, G::half_threads1,
                        [&] { k_half_pass1<N, G::E1, G::P, true, G::dma, G::fpar>(h0T, descale, omT, inter, nyq, tw, lay, time, L, 0, batch); }, batch_count);
    else emu_launch(G::half_grid1, G::half_threads1,
                    [&] { k_half_pass1<N, G::E1, G::P, false, G::dma, G::fpar>(h0T, 1.0f, omT, inter, nyq, tw, lay, time, L, 0, batch); }, batch_count);
    emu_launch(G::half_grid2, G::half_threads2,
               [&] { k_half_pass2<N, G::E2, CHUNK_W, G::R2h, G::p2_group, G::ppar, false, true>(inter, out, tw, lay, plane, plane_channel, batch); }, batch_count);
    return 0;
}
template <int N, bool I16, int PS = 2> static int run_half_split(const void* h0T, int f16, float descale, const float* omT, c32* inter, c32* nyq,
                                                     float4* out, const c32* tw, InterLayout lay, float time, float L, float* scales) {
    using G = Geo<N, PS>;
    static_assert(G::can_split, "split geometry");
    if (f16) emu_launch(G::half_grid1, G::split_threads1,
                        [&] { k_half_pass1_split<N, G::E1S, G::P, true, I16>(h0T, descale, omT, inter, nyq, tw, lay, time, L, 0, scales); });
    else emu_launch(G::half_grid1, G::split_threads1,
                    [&] { k_half_pass1_split<N, G::E1S, G::P, false, I16>(h0T, 1.0f, omT, inter, nyq, tw, lay, time, L, 0, scales); });
    // the split geometry's pass 2 is the real-output kernel, which reads its column-major chunks (N >= 8192 in the product)
    if constexpr (G::real_threads2 % 16 == 0)
        emu_launch(N, G::real_threads2, [&] { k_half_pass2_real<N, G::E, CHUNK_W, G::p2_group, false, I16, 4, true>(inter, out, tw, lay, scales, plane, plane_channel); });
    else return -7;
    return 0;
}
template <int N> static int run_half(int psel, const void* h0T, int f16, float descale, const float* omT, c32* inter, c32* nyq,
                                     float4* out, const c32* tw, InterLayout lay, float time, float L, float* scales) {
    if (psel == 21) {                                   // P = 1 with the split geometry (the N = 16384 kernels: one column per workgroup)
        if constexpr (N >= 1024) return run_half_split<N, false, 1>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L, nullptr);   // (whole waves)
        else return -3;
    }
    if (psel == 22 || psel == 23) {                     // P = 2 with the split geometry; 23: + the 16-bit intermediate
        if constexpr (N >= 512) {
            if (psel == 23) return scales ? run_half_split<N, true>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L, scales) : -6;
            return run_half_split<N, false>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L, nullptr);
        } else return -3;
    }
    if (psel == 2) return run_half_p<N, 2>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L);
    if (psel == 1) return run_half_p<N, 1>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L);
    if constexpr (CHUNK_W % Geo<N, 0>::P != 0) return -4;
    else return run_half_p<N, 0>(h0T, f16, descale, omT, inter, nyq, out, tw, lay, time, L);
}

// one tile sharded over `world` ranks, second generation (ocean_tile_pass1 / ocean_tile_pass2 of csrc/ocean_api.hip: same
// kernels, same geometry, same layouts)
template <int N, int PSEL> static int run_tile_pass1(int rank, int world, int part, int parts, const void* h0T, int f16, float descale, const float* omT,
                                                     c32* send, c32* nyq, const c32* tw, float time, float L) {
    using G = Geo<N, PSEL>;
    if (!G::tile_supported(world, parts)) return -5;
    const InterLayout lay = G::tile_layout(world, parts);
    const int groups = (N / 2 / world / parts) / G::P;
    const int x_group0 = (rank * parts + part) * groups;
    if (f16) emu_launch(groups, G::half_threads1,
                        [&] { k_half_pass1<N, G::E1, G::P, true, G::dma, G::fpar>(h0T, descale, omT, send, nyq, tw, lay, time, L, x_group0, FrameBatch{}); });
    else emu_launch(groups, G::half_threads1,
                    [&] { k_half_pass1<N, G::E1, G::P, false, G::dma, G::fpar>(h0T, 1.0f, omT, send, nyq, tw, lay, time, L, x_group0, FrameBatch{}); });
    return 0;
}
template <int N, int PSEL> static int run_tile_pass2(int world, int parts, const c32* recv, float4* out, const c32* tw) {
    using G = Geo<N, PSEL>;
    if (!G::tile_supported(world, parts)) return -5;
    const InterLayout lay = G::tile_layout(world, parts);
    emu_launch((N / world) / G::R2h, G::half_threads2,
               [&] { k_half_pass2<N, G::E2, CHUNK_W, G::R2h, G::p2_group, G::ppar, true>(recv, out, tw, lay, nullptr, 0, FrameBatch{}); });
    return 0;
}
// ... with the split geometry (what ocean_tile_pass1 / ocean_tile_pass2 launch at N >= 8192: k_half_pass1_split, k_half_pass2_real<SHARD>)
template <int N, int PS> static int run_tile_split(int what, int rank, int world, int part, int parts, const void* h0T, int f16, float descale, const float* omT,
                                                   c32* buf, c32* nyq, float4* out, const c32* tw, float time, float L) {
    using G = Geo<N, PS>;
    if constexpr (!G::can_split || G::real_threads2 % 16 != 0) return -7;
    else {
        if (!G::tile_supported(world, parts)) return -5;
        const InterLayout lay = G::tile_layout(world, parts);
        if (what == 1) {
            const int groups = (N / 2 / world / parts) / G::P;
            const int x_group0 = (rank * parts + part) * groups;
            if (f16) emu_launch(groups, G::split_threads1, [&] { k_half_pass1_split<N, G::E1S, G::P, true, false>(h0T, descale, omT, buf, nyq, tw, lay, time, L, x_group0, nullptr); });
            else emu_launch(groups, G::split_threads1, [&] { k_half_pass1_split<N, G::E1S, G::P, false, false>(h0T, 1.0f, omT, buf, nyq, tw, lay, time, L, x_group0, nullptr); });
        } else {
            emu_launch(N / world, G::real_threads2, [&] { k_half_pass2_real<N, G::E, CHUNK_W, G::p2_group, true>(buf, out, tw, lay, nullptr, nullptr, 0); });
        }
        return 0;
    }
}
template <int N> static int run_tile(int what, int psel, int rank, int world, int part, int parts, const void* h0T, int f16, float descale, const float* omT,
                                     c32* buf, c32* nyq, float4* out, const c32* tw, float time, float L) {
    if (psel == 22) return run_tile_split<N, 2>(what, rank, world, part, parts, h0T, f16, descale, omT, buf, nyq, out, tw, time, L);
    if (psel == 21) {
        if constexpr (N >= 1024) return run_tile_split<N, 1>(what, rank, world, part, parts, h0T, f16, descale, omT, buf, nyq, out, tw, time, L);
        else return -3;
    }
    if (psel == 1) return what == 1 ? run_tile_pass1<N, 1>(rank, world, part, parts, h0T, f16, descale, omT, buf, nyq, tw, time, L) : run_tile_pass2<N, 1>(world, parts, buf, out, tw);
    if (psel == 2) return what == 1 ? run_tile_pass1<N, 2>(rank, world, part, parts, h0T, f16, descale, omT, buf, nyq, tw, time, L) : run_tile_pass2<N, 2>(world, parts, buf, out, tw);
    if constexpr (CHUNK_W % Geo<N, 0>::P != 0) return -4;
    else return what == 1 ? run_tile_pass1<N, 0>(rank, world, part, parts, h0T, f16, descale, omT, buf, nyq, tw, time, L) : run_tile_pass2<N, 0>(world, parts, buf, out, tw);
}

// staged path with the chunked hand-off: rows (natural -> chunked), cols (in place, chunked), correction / un-chunk
template <int N> static int run_stage(int what, c32* nat, c32* chk, const c32* tw, InterLayout lay) {
    using G = Geo<N>;
    if constexpr (!G::stage_chunked) return -4;
    else {
        if (what == 0) emu_launch(G::stage_grid, G::stage_threads, [&] { k_stage_rows<N, G::E>(nat, chk, tw, lay); });
        else emu_launch(G::stage_grid, G::stage_threads, [&] { k_stage_cols<N, G::E>(chk, tw, lay); });
        return 0;
    }
}

// one tile sharded by row blocks (ocean_shard_rows / ocean_shard_cols of csrc/ocean_shard.hip, same kernels, same order)
template <int N> static int run_shard_rows(int rank, int world, const c32* h0_own, const c32* h0_partner, const float* om,
                                           c32* fld, c32* send, const c32* tw, float time, float L) {
    constexpr int E = 16, T = N / E, LPW = (256 / T) > 1 ? (256 / T) : 1;
    const int rows = N / world;
    const size_t block = (size_t)rows * N;
    int cols_log2 = 0;
    while ((1 << cols_log2) < rows) ++cols_log2;
    emu_launch((int)((block / 2 + 255) / 256), 256, [&] {
        k_propagate(h0_own, h0_partner, om, fld + 1 * block, fld + 0 * block, fld + 2 * block, N, rank * rows, rows, time, L, 3u);
    });
    for (int f = 0; f < 3; ++f)
        emu_launch(rows / LPW, T * LPW, [&] { k_shard_rows<N, E, LPW>(fld + f * block, send, tw, f, rows, cols_log2); });
    return 0;
}
template <int N> static int run_shard_cols(int rank, int world, const c32* recv, c32* fld, float4* out, const c32* tw) {
    constexpr int E = 16, T = N / E, LPW = (256 / T) > 1 ? (256 / T) : 1;
    const int cols = N / world;
    const size_t block = (size_t)cols * N;
    for (int f = 0; f < 3; ++f) {
        emu_launch((N / 32) * (cols / 32), 256, [&] { k_shard_transpose(recv, fld + f * block, N, f, cols, cols); });
        emu_launch(cols / LPW, T * LPW, [&] { k_fft_lines<N, E, LPW, false>(fld + f * block, tw); });
    }
    emu_launch((int)((block / 2 + 255) / 256), 256, [&] {
        k_correct(fld + 1 * block, fld + 0 * block, fld + 2 * block, out, N, rank * cols, cols);
    });
    return 0;
}

#define DISPATCH(n, CALL)                 \
    switch (n) {                          \
        case 256: return CALL(256);       \
        case 512: return CALL(512);       \
        case 1024: return CALL(1024);     \
        case 2048: return CALL(2048);     \
        case 4096: return CALL(4096);     \
        default: return -2;               \
    }

extern "C" {
int emu_chunk_w() { return CHUNK_W; }
int emu_chunk_r() { return CHUNK_R; }
int emu_inter_bshift(int n) {
#define C_(N) Geo<N>::inter_bshift
    DISPATCH(n, C_)
#undef C_
}
// 1 when fused pass 1 of this size takes its inputs through the LDS-DMA ring (psel as emu_frame_half: 22 / 23 = split kernels, always)
int emu_uses_dma(int n, int psel) {
    if (psel == 21 || psel == 22 || psel == 23) return 1;
#define C_(N) ((psel == 2) ? (int)Geo<N, 2>::dma : (psel == 1) ? (int)Geo<N, 1>::dma : (int)Geo<N, 0>::dma)
    DISPATCH(n, C_)
#undef C_
}
int emu_frame_p(int n) {
#define C_(N) Geo<N>::P
    DISPATCH(n, C_)
#undef C_
}
int emu_fft_lines(int n, int col, float* data, const float* tw) {
#define C_(N) run_fft_lines<N>(col, (c32*)data, (const c32*)tw)
    DISPATCH(n, C_)
#undef C_
}
int emu_frame_half(int n, int psel, const void* h0T, int f16, float descale, const float* omT, float* inter, c32* nyq, float* out,
                   const float* tw, size_t sx, size_t sy, size_t fs, int bshift, float time, float L, float* scales) {
#define C_(N) run_half<N>(psel, h0T, f16, descale, omT, (c32*)inter, (c32*)nyq, (float4*)out, (const c32*)tw, InterLayout{sx, sy, fs, bshift}, time, L, scales)
    DISPATCH(n, C_)
#undef C_
}
int emu_tile(int n, int what, int psel, int rank, int world, int part, int parts, const void* h0T, int f16, float descale, const float* omT, float* buf,
             float* nyq, float* out, const float* tw, float time, float L) {
#define C_(N) run_tile<N>(what, psel, rank, world, part, parts, h0T, f16, descale, omT, (c32*)buf, (c32*)nyq, (float4*)out, (const c32*)tw, time, L)
    DISPATCH(n, C_)
#undef C_
}
int emu_stage(int n, int what, float* nat, float* chk, const float* tw, size_t sx, size_t sy, size_t fs) {
#define C_(N) run_stage<N>(what, (c32*)nat, (c32*)chk, (const c32*)tw, InterLayout{sx, sy, fs})
    DISPATCH(n, C_)
#undef C_
}
int emu_unchunk(int n, const float* chk, float* nat, size_t sx, size_t sy, size_t fs) {
    emu_launch(n / 4, 256, [&] { k_unchunk((const c32*)chk, (c32*)nat, n, InterLayout{sx, sy, fs}); });
    return 0;
}
int emu_correct_chunked(int n, const float* h, const float* dx, const float* dz, float* out, size_t sx, size_t sy, size_t fs) {
    emu_launch(n / 4, 256, [&] { k_correct_chunked((const c32*)h, (const c32*)dx, (const c32*)dz, (float4*)out, n, InterLayout{sx, sy, fs}); });
    return 0;
}
int emu_shard_rows(int n, int rank, int world, const float* h0_own, const float* h0_partner, const float* om, float* fld,
                   float* send, const float* tw, float time, float L) {
#define C_(N) run_shard_rows<N>(rank, world, (const c32*)h0_own, (const c32*)h0_partner, om, (c32*)fld, (c32*)send, (const c32*)tw, time, L)
    DISPATCH(n, C_)
#undef C_
}
int emu_shard_cols(int n, int rank, int world, const float* recv, float* fld, float* out, const float* tw) {
#define C_(N) run_shard_cols<N>(rank, world, (const c32*)recv, (c32*)fld, (float4*)out, (const c32*)tw)
    DISPATCH(n, C_)
#undef C_
}
int emu_positions(int n, const float* rgba, float* positions, int verts, float ox, float oz) {
    const int grid = (verts * verts + 255) / 256;
    emu_launch(grid, 256, [&] { k_positions((const float4*)rgba, (float4*)positions, n, verts, ox, oz); });
    return 0;
}
int emu_normals(int n, const float* rgba, float* normals, int channel) {
    const int rows = normals_rows(n);                   // as ocean_normals of csrc/ocean_api.hip
    const int grid = (n / 256) * (n / rows);
    if (rows == 1) emu_launch(grid, 256, [&] { k_normals<1>((const float4*)rgba, (float4*)normals, n, channel); });
    else if (rows == 2) emu_launch(grid, 256, [&] { k_normals<2>((const float4*)rgba, (float4*)normals, n, channel); });
    else if (rows == 4) emu_launch(grid, 256, [&] { k_normals<4>((const float4*)rgba, (float4*)normals, n, channel); });
    else emu_launch(grid, 256, [&] { k_normals<8>((const float4*)rgba, (float4*)normals, n, channel); });
    return 0;
}
// where the fused pass 2 of the following emu_frame_half calls stores its source-channel plane (n * n floats; must be set)
int emu_set_plane(float* p, int channel) { plane = p; plane_channel = channel; return 0; }
// the following emu_frame_half calls (plain geometry, N <= 1024) run `count` time steps t, t + dt, ... as ONE launch pair; frame y
// uses inter + y * inter_stride (elements), nyq + y * 3 n, out + y * out_stride (texels)
int emu_set_batch(int count, float dt, unsigned inter_stride, size_t out_stride) {
    batch_count = count;
    batch = FrameBatch{dt, inter_stride, out_stride};
    return 0;
}
// ... and K TILES instead of K time steps (ocean_frame_tiles): frame y reads its own transposed inputs, these strides apart
int emu_set_batch_tiles(size_t spec_stride_bytes, unsigned omega_stride) {
    batch.spec_stride_bytes = spec_stride_bytes;
    batch.omega_stride = omega_stride;
    return 0;
}
int emu_normals_plane_bands(int n, const float* src_plane, float* normals) {   // what launch_normals_plane runs at N >= 8192
    emu_launch(n / NORMALS_BAND_ROWS, 256, [&] { k_normals_plane_bands<NORMALS_BAND_ROWS>(src_plane, (float4*)normals, n); });
    return 0;
}
int emu_normals_plane(int n, const float* src_plane, float* normals) {
    const int rows = normals_plane_rows(n);             // as launch_normals_plane of csrc/ocean_api.hip
    const int grid = (n / 256) * (n / rows) / 4;
    if (rows == 2) emu_launch(grid, 256, [&] { k_normals_plane<2>(src_plane, (float4*)normals, n); });
    else if (rows == 4) emu_launch(grid, 256, [&] { k_normals_plane<4>(src_plane, (float4*)normals, n); });
    else if (rows == 16) emu_launch(grid, 256, [&] { k_normals_plane<16>(src_plane, (float4*)normals, n); });
    else emu_launch(grid, 256, [&] { k_normals_plane<8>(src_plane, (float4*)normals, n); });
    return 0;
}
// the staged column pass of N >= 8192 (two steps, k_cols4_a / k_cols4_b) at sizes the emulation can run: nf = s * m
int emu_cols4(int nf, int s, float* data, float* dst, const float* tw) {
    auto run = [&](auto NF, auto S) {
        constexpr int nf_ = decltype(NF)::value, s_ = decltype(S)::value, m_ = nf_ / s_;
        emu_launch((nf_ / 16) * s_, (m_ / 16) * 16, [&] { k_cols4_a<nf_, s_, 16, 16>((c32*)data, (const c32*)tw); });
        emu_launch((nf_ / ((s_ >= 32) ? 1 : 2) / 256) * m_, 256, [&] { k_cols4_b<nf_, s_>((const c32*)data, (c32*)dst); });
        return 0;
    };
    if (nf == 1024 && s == 4) return run(std::integral_constant<int, 1024>{}, std::integral_constant<int, 4>{});
    if (nf == 2048 && s == 8) return run(std::integral_constant<int, 2048>{}, std::integral_constant<int, 8>{});
    if (nf == 4096 && s == 16) return run(std::integral_constant<int, 4096>{}, std::integral_constant<int, 16>{});
    return -2;
}
// ... and its second step fused with the correction (k_cols4_b_correct): step A of the three fields in place, then the map
int emu_cols4_correct(int nf, int s, float* h, float* dx, float* dz, float* out, const float* tw) {
    auto run = [&](auto NF, auto S) {
        constexpr int nf_ = decltype(NF)::value, s_ = decltype(S)::value, m_ = nf_ / s_;
        for (float* f : {h, dx, dz})
            emu_launch((nf_ / 16) * s_, (m_ / 16) * 16, [&] { k_cols4_a<nf_, s_, 16, 16>((c32*)f, (const c32*)tw); });
        emu_launch((nf_ / 256) * m_, 256, [&] { k_cols4_b_correct<nf_, s_>((const c32*)h, (const c32*)dx, (const c32*)dz, (float4*)out); });
        return 0;
    };
    if (nf == 1024 && s == 4) return run(std::integral_constant<int, 1024>{}, std::integral_constant<int, 4>{});
    if (nf == 2048 && s == 8) return run(std::integral_constant<int, 2048>{}, std::integral_constant<int, 8>{});
    return -2;
}
int emu_propagate(int n, const float* h0, const float* omega, float* h, float* dx, float* dz, float time, float L, unsigned quirks) {
    if (quirks == 3u) {                                 // as launch_propagate of csrc/ocean_api.hip: the paired kernel for the reference quirks
        const int gridp = (n * n / 4 + 255) / 256;
        emu_launch(gridp, 256, [&] { k_propagate_paired((const c32*)h0, omega, (c32*)h, (c32*)dx, (c32*)dz, n, time, L); });
        return 0;
    }
    const int grid = (n * n / 2 + 255) / 256;
    emu_launch(grid, 256, [&] { k_propagate((const c32*)h0, (const c32*)h0, omega, (c32*)h, (c32*)dx, (c32*)dz, n, 0, n, time, L, quirks); });
    return 0;
}
int emu_correct(int n, const float* h, const float* dx, const float* dz, float* out) {
    const int grid = (n * n / 2 + 255) / 256;
    emu_launch(grid, 256, [&] { k_correct((const c32*)h, (const c32*)dx, (const c32*)dz, (float4*)out, n, 0, n); });
    return 0;
}
}
