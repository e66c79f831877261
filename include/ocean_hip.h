/*
 * ocean_hip.h -- C ABI of the MI355X-native gfx-ocean compute path.
 *
 * Drop-in boundary for the reference's private `ocean` / `fft` modules
 * (src/lib.rs:38-39) and for the 190-line dispatch block of `Renderer::render`
 * (src/render.rs:1101-1310).  Plain pointers and sizes only: a Rust shim binds
 * these with `extern "C"` (INTEGRATION.md shows the stub), Python binds them
 * with ctypes (gfx_ocean_amd/_lib.py), C++ through host/ocean.hpp.
 *
 * Conventions
 *   - Every call returns an int32_t status (OCEAN_OK = 0, negative = error) and
 *     never throws or aborts; `ocean_last_error` gives the message.  The Rust
 *     shim maps a non-zero status to `Err(Box<dyn Error>)` where the reference
 *     returns `Result` (src/fft.rs:19, src/ocean.rs:25,194) and would `.unwrap()`.
 *   - Complex fields are interleaved (re, im) fp32, row-major, index = x + N*y
 *     (shader/propagate.comp:43).  The displacement map is linear RGBA32F,
 *     out[(y*N + x)*4 + c] = (disp_x, height, disp_z, 0)  (shader/correction.comp:31-34),
 *     replacing the reference's Rgba32Sfloat storage image (src/render.rs:820-869).
 *   - `stream` is a `hipStream_t` passed as `void*`; NULL = the context's own
 *     stream.  Stream order replaces the reference's pipeline barriers
 *     (src/render.rs:1132-1156,1181-1208,1233-1278,1289-1310).  The staged calls keep
 *     per-field layout state on the host: issue the calls of one field on ONE stream.
 *     The synchronous readbacks (ocean_read_*) wait for the context stream, and for the
 *     whole device once any dispatch of this context has been put on a caller stream.
 *     The launching calls (ocean_frame*, the staged dispatches, ocean_normals, ocean_positions, ocean_pack_displacement,
 *     ocean_tile_pass1/2) enqueue kernels on `stream` and nothing else -- no allocation once their buffers exist, no copy, no
 *     host wait -- so a caller stream in capture mode records them into a hipGraph of the caller's (kernel nodes only;
 *     tests/test_gpu_parity.py::test_frames_are_capturable_in_a_hip_graph).  Bind outputs / switch the normal field on BEFORE
 *     the capture: the pointers and the time are baked into the recorded launches.
 *   - A context is bound to one GPU and is not thread-safe (the reference is
 *     single-threaded: winit loop, src/lib.rs:100-170).  One context per GPU for
 *     tile-parallel runs.
 */
#ifndef OCEAN_HIP_H
#define OCEAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (additive throughout: no entry point has changed its signature or meaning)
 * 2: + ocean_checksum_displacement, ocean_pack_displacement, ocean_packed_bytes, the ocean_shard_* family;
 *    + ocean_set_intermediate, ocean_intermediate, ocean_tile_exchange_bytes, ocean_tile_pass1, ocean_tile_pass2 (shipped under
 *    the same number in round 3); readbacks wait for the whole device once a dispatch has been put on a caller stream
 * 3: + ocean_frame_times, ocean_time_frame_batches; ocean_sync and ocean_context_destroy honour caller streams like the readbacks
 * 4: + ocean_set_frame_normals, ocean_frame_normals, ocean_normals_device_ptr, ocean_frame_times_ex (the frame with the normal
 *    field as one workload); + ocean_context_create_tiles, ocean_context_tiles, ocean_upload_spectrum_tile, ocean_frame_tiles (K tiles per launch pair);
 *    + ocean_frame_batch, ocean_batch_device_ptr, ocean_batch_normals_device_ptr, ocean_read_batch_normals, ocean_read_batch_displacement, ocean_time_frame_batch
 *    (K time steps per launch pair); + ocean_context_create_ex, ocean_context_create_tile_rank, ocean_context_flags (contexts with only the buffers -- and, for a rank of a
 *    sharded tile, only the input lines -- their path uses); + ocean_device_count,
 *    ocean_device_pci_bus_id; + ocean_bind_displacement_fd (the map in memory imported from another API's file descriptor);
 *    ocean_time_frame_batches also bounds frames_per_batch (<= 4096); + ocean_upload_spectrum_device (the upload's device-side half,
 *    asynchronous, from memory the GPU can read)
 *    (round 6, still 4: additive) + status OCEAN_E_UNSUPPORTED from ocean_context_create_tile_rank; ocean_read_batch_* bound the index by the
 *    LAST batch's frame count; ocean_upload_spectrum_device orders itself against the context stream */
#define OCEAN_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------------------------- */
#define OCEAN_OK 0
#define OCEAN_E_INVALID_ARG (-1)
#define OCEAN_E_UNSUPPORTED_N (-2) /* resolution must be a power of two in [256, 16384] */
#define OCEAN_E_HIP (-3)
#define OCEAN_E_OOM (-4)
#define OCEAN_E_STATE (-5) /* e.g. frame requested before ocean_upload_spectrum */
#define OCEAN_E_UNSUPPORTED (-6) /* ocean_context_create_tile_rank: the runtime cannot reserve / map address ranges sparsely (use OCEAN_CTX_TILE_RANK) */

/* ---- field selectors: Fft::desc_sets[0,1,2] -> dx_spec, dy_spec, dz_spec (src/render.rs:971-988) */
#define OCEAN_FIELD_DX 0 /* disp_x spectrum  (propagate binding 4, src/render.rs:951-952) */
#define OCEAN_FIELD_DY 1 /* height spectrum  (propagate binding 3, src/render.rs:949-950) */
#define OCEAN_FIELD_DZ 2 /* disp_z spectrum  (propagate binding 5, src/render.rs:953-954) */
#define OCEAN_FIELD_ALL (-1)

/* ---- uniform blocks ----------------------------------------------------------------------- */
/* src/ocean.rs:8-13 `PropagateLocals` / shader/propagate.comp:16-20 (std140 offsets 0/4/8).
 * The reference struct has no #[repr(C)] (quirk Q6); this is the explicit layout. */
typedef struct OceanPropagateLocals {
    float time;
    int32_t resolution;
    float domain_size;
} OceanPropagateLocals;

/* src/ocean.rs:179-182 `CorrectionLocals` / shader/correction.comp:6-8.  The reference shader
 * ignores it and hard-codes 512 (quirk Q4); here it must equal the context resolution. */
typedef struct OceanCorrectionLocals {
    uint32_t resolution;
} OceanCorrectionLocals;

/* ---- opaque handles ----------------------------------------------------------------------- */
typedef struct OceanContext OceanContext;         /* device + buffers: the slice of `Renderer` (src/render.rs:72-101) the path needs */
typedef struct OceanFft OceanFft;                 /* src/fft.rs:7-16   `Fft<B>` */
typedef struct OceanPropagation OceanPropagation; /* src/ocean.rs:15-22 `Propagation<B>` */
typedef struct OceanCorrection OceanCorrection;   /* src/ocean.rs:184-191 `Correction<B>` */

/* ---- context: device open + buffer allocation (src/render.rs:118-172, 607-729, 820-869) ---- */
int32_t ocean_abi_version(void);
/* Visible HIP devices (0 if none; < 0: error status), and the PCI bus id "dddd:bb:dd.f" of one of them -- what a launcher needs to
 * fail fast when asked for more ranks than there are GPUs and to place rank r's host thread on the NUMA node of GPU r
 * (/sys/bus/pci/devices/<id>/numa_node, local_cpulist; bench.py).  No context needed. */
int32_t ocean_device_count(void);
int32_t ocean_device_pci_bus_id(int32_t device_ordinal, char* out, int32_t capacity /* >= 16 */);
int32_t ocean_context_create(int32_t device_ordinal, int32_t resolution, OceanContext** out_ctx);
/* ... with exactly the buffers the caller's path uses (the reference sizes one allocation for what it binds, src/render.rs:607-670).
 * A full context holds both paths' buffers: 100 B/texel at N <= 4096 (1.6 GiB at 4096), 76 above (20 GiB at 16384; + 8 per field
 * on the first ocean_fft_cols of that field, whose two-step column pass at N >= 8192 works out of place).
 *   OCEAN_CTX_FUSED_ONLY  no natural-layout copies, fields or chunked hand-off (the staged path's 60 / 36 B/texel): ocean_frame*,
 *                         the consumers, the batch, the measurement loops and ocean_tile_pass1/2 work; ocean_propagate,
 *                         ocean_fft_rows/cols, ocean_correct, ocean_read/write_field, ocean_read_spectrum, ocean_profile_staged and
 *                         non-reference quirks return OCEAN_E_STATE with a message.  40 B/texel (10 GiB at 16384); the upload goes
 *                         through a staging buffer of at most 64 MiB that lives for the call (slabs of whole rows).
 *   OCEAN_CTX_TILE_RANK   one rank of a tile sharded over several GPUs (implies FUSED_ONLY): the static inputs only, 12 B/texel --
 *                         the intermediate and the rows live in the caller's exchange buffers; everything but the upload and
 *                         ocean_tile_pass1/2 returns OCEAN_E_STATE.
 *   OCEAN_CTX_TILE_BANDS  (reported by ocean_context_flags, set by ocean_context_create_tile_rank only) a TILE_RANK context of ONE
 *                         rank of ONE world size that backs only the lines of the transposed inputs that rank's pass 1 reads -- two
 *                         bands of N/(2 world) + 1 lines each (+ the Nyquist column's two on rank 0) out of N: 12/world B/texel
 *                         instead of 12 (0.4 instead of 3 GiB per rank at N = 16384, world 8).  The ranges keep their full-size
 *                         addresses (HIP virtual-memory API), so the kernels index absolute lines as always; the upload stages and
 *                         copies only the columns that become those lines (1 / world of the tile crosses PCIe); fp32 spectrum only;
 *                         ocean_tile_pass1 with another rank or world: OCEAN_E_INVALID_ARG. */
#define OCEAN_CTX_FUSED_ONLY 1u
#define OCEAN_CTX_TILE_RANK 2u
#define OCEAN_CTX_TILE_BANDS 4u
int32_t ocean_context_create_ex(int32_t device_ordinal, int32_t resolution, uint32_t flags, OceanContext** out_ctx);
int32_t ocean_context_create_tile_rank(int32_t device_ordinal, int32_t resolution, int32_t rank, int32_t world, OceanContext** out_ctx);
uint32_t ocean_context_flags(const OceanContext* ctx);
void ocean_context_destroy(OceanContext* ctx);            /* NULL-safe; src/render.rs:1383-1438 */
const char* ocean_last_error(const OceanContext* ctx);    /* ctx may be NULL: last error of a failed create */
int32_t ocean_resolution(const OceanContext* ctx);

/* Staging upload of the initial spectrum h0 (N*N complex) and dispersion omega (N*N real):
 * src/render.rs:742-818 (decode + staging) and :872-924 (copy_buffer, submit, wait).  Synchronous. */
int32_t ocean_upload_spectrum(OceanContext* ctx, const float* h0_re_im, const float* omega);

/* The second half of that upload alone -- cmd_buffer.copy_buffer(staging -> initial_spec / omega_buffer), src/render.rs:896-915 --
 * for a spectrum that already lives where the GPU can read it: device or managed memory (generated there, e.g. when the wind
 * changes), or host memory registered with the runtime (a mapped staging buffer of the caller's, as the reference's CPU_VISIBLE
 * one, src/render.rs:749-761).  Natural layout, c32[N*N] and f32[N*N].  ASYNCHRONOUS on `stream` (NULL = the context's): nothing
 * waits on the host; frames launched afterwards on that stream OR on the context's own see the new spectrum, frames launched before on
 * either keep the old one (an upload on a caller stream is ordered against the context stream with events on both sides), and the
 * source buffers may be reused once the stream has passed the call.  Frames on OTHER caller streams are the caller's to order against
 * the upload, as the reference orders its copy_buffer against the dispatches with a barrier (src/render.rs:896-915).  fp32 storage
 * (the fp16 storage needs the spectrum's maximum first: ocean_upload_spectrum_f16).  `tile`: 0, or the tile of a context of several.
 * Pageable host memory, memory of another device and registered host memory without a mapping for the context's device are rejected
 * with OCEAN_E_INVALID_ARG. */
int32_t ocean_upload_spectrum_device(OceanContext* ctx, int32_t tile, const void* h0_device, const void* omega_device, void* stream);

/* BASELINE config 5 ("fp16 spectrum / fp32 accumulate"): same upload, but the fused path keeps the
 * initial spectrum in HBM as two fp16 per texel, h0 * 2^scale_log2 rounded to nearest even, the
 * scale chosen so that max|component| lands in [2^14, 2^15) (small amplitudes would otherwise fall
 * below the fp16 normal range).  Every kernel computes in fp32 on the dequantised values; the staged
 * path and ocean_read_spectrum use exactly those values, so parity is judged against the oracle
 * fed the same quantised inputs.  omega stays fp32. */
int32_t ocean_upload_spectrum_f16(OceanContext* ctx, const float* h0_re_im, const float* omega);
int32_t ocean_spectrum_scale_log2(const OceanContext* ctx);          /* 0 for an fp32 upload */
/* The initial spectrum the kernels actually use (dequantised if uploaded as fp16), N*N complex. */
int32_t ocean_read_spectrum(OceanContext* ctx, float* host_re_im);

/* ---- stage objects: init/destroy mirror the reference 1:1 -------------------------------- */
int32_t ocean_fft_init(OceanContext* ctx, OceanFft** out);                  /* src/fft.rs:19-100 */
void ocean_fft_destroy(OceanFft* fft);                                      /* src/fft.rs:102-110 */
int32_t ocean_propagation_init(OceanContext* ctx, OceanPropagation** out);  /* src/ocean.rs:25-168 */
void ocean_propagation_destroy(OceanPropagation* p);                        /* src/ocean.rs:170-176 */
int32_t ocean_correction_init(OceanContext* ctx, OceanCorrection** out);    /* src/ocean.rs:194-319 */
void ocean_correction_destroy(OceanCorrection* c);                          /* src/ocean.rs:321-327 */

/* ---- dispatches: one call per reference `dispatch` ------------------------------------------ */
/* bind propagate pipeline + dispatch [N/16, N/16, 1]: src/render.rs:1101-1130, shader/propagate.comp:42-72 */
int32_t ocean_propagate(OceanPropagation* p, const OceanPropagateLocals* locals, void* stream);
/* bind row_pass; dispatch [1, N, 1] per set: src/render.rs:1158-1179, shader/fft_row.comp:44-63 */
int32_t ocean_fft_rows(OceanFft* fft, int32_t field, void* stream);
/* bind col_pass; dispatch [1, N, 1] per set: src/render.rs:1210-1231, shader/fft_col.comp:44-63.
 * (N >= 8192: the pass has two steps and the second runs with the field's next consumer on THAT call's stream -- inside
 * ocean_correct when it follows the three column passes, before ocean_read_field / a further pass otherwise.  Results are those
 * of the in-place pass in every order of calls; callers that spread the staged calls over several streams order them as the
 * reference's barriers do, src/render.rs:1181-1208, 1233-1278.) */
int32_t ocean_fft_cols(OceanFft* fft, int32_t field, void* stream);
/* bind correction; dispatch [N/16, N/16, 1]: src/render.rs:1280-1287, shader/correction.comp:24-35 */
int32_t ocean_correct(OceanCorrection* c, const OceanCorrectionLocals* locals, void* stream);

/* Whole hot path of one frame (src/render.rs:1101-1310) with the default domain size 1000
 * (src/render.rs:46) or the one given; fused kernels, same results as the four staged calls
 * within fp32 re-association.  The staged field buffers are NOT updated by this call.
 * Range of `time`: the reference feeds wall-clock seconds (src/lib.rs:139-141) into d = omega * t, one fp32 multiply
 * (shader/propagate.comp:55), and so does this library; cos / sin of that fp32 phase are then taken after a two-constant
 * reduction to revolutions (fused path) or by ocml's sincosf (staged path).  Measured on the reference's data at N = 512:
 * t = 2e4, 2e5, 2e6 s (|omega t| up to 9.5e6 rad, 23 days of run time) -> normalised max error against the oracle's
 * correctly rounded cos / sin of the same fp32 phase 1.4e-6, 1.2e-6, 1.3e-6 (fused), 1.7e-6, 1.5e-6, 1.6e-6 (staged);
 * tolerance 1e-4 (tests/test_gpu_parity.py::test_phase_range_of_the_fused_propagate).  The reduction is exact to
 * < 1e-7 revolutions up to |omega t| ~ 1e8 rad; past ~1e7 rad the fp32 PHASE itself (1 ulp = 1 rad) is the error,
 * for the reference as for this library. */
int32_t ocean_frame(OceanContext* ctx, float time, void* stream);
int32_t ocean_frame_ex(OceanContext* ctx, const OceanPropagateLocals* locals, void* stream);

/* Quirk switches.  Default = OCEAN_QUIRKS_REFERENCE = the shipped shaders' arithmetic.
 *   Q1 (shader/propagate.comp:45-46,50-53): the wave index 2g - N - 1 is evaluated in uint and wraps for g <= N/2;
 *      off = signed.
 *   Q2 (shader/propagate.comp:48,59-62): the "-k" partner is texel N-1-g and is NOT conjugated; off = the partner
 *      is (N+1-g) % N on both axes (k(g) = -k(N+1-g); g = 0 and 1 pair with each other) and enters conjugated.
 * Every entry point honours the setting; with a non-reference setting ocean_frame runs the eight staged
 * dispatches (the fused kernels implement the reference only) and therefore updates the field buffers. */
#define OCEAN_QUIRK_Q1_UINT_WAVE_INDEX 1u
#define OCEAN_QUIRK_Q2_MIRROR_NO_CONJ 2u
#define OCEAN_QUIRKS_REFERENCE 3u
int32_t ocean_set_quirks(OceanContext* ctx, uint32_t quirks);
uint32_t ocean_quirks(const OceanContext* ctx);

/* Precision of the intermediate between the two fused launches.  Default OCEAN_INTER_F32 (complex fp32, what every
 * parity figure of this library refers to).  OCEAN_INTER_BFP16 is SURVEY 8d's "B_frame16" for BASELINE config 5: int16
 * (re, im) mantissas with one power-of-two scale per block of 64 rows x 2 columns in a side array -- 12 instead of
 * 24 B/texel through the intermediate; 2.7-3.1e-5 normalised max against the fp32 intermediate at N = 8192 (tolerance
 * 1e-4; tools/inter16_numerics.py).  Opt-in, never the default; N = 8192 only (OCEAN_E_UNSUPPORTED_N otherwise).
 * Halves the intermediate's footprint (403 instead of 805 MB); since the real-output row pass (ABI unchanged) it is no
 * longer faster than the default: 1327 against 1390-1425 frames/s with the fp16-stored spectrum. */
#define OCEAN_INTER_F32 0
#define OCEAN_INTER_BFP16 1
int32_t ocean_set_intermediate(OceanContext* ctx, int32_t mode);
int32_t ocean_intermediate(const OceanContext* ctx);

/* K consecutive time steps of this tile -- frame i at time t0 + dt * (float)i (fp32, as ocean_time_frames counts) -- each into
 * its own map: out_base_device + i * out_stride_bytes (caller device memory: 16-byte aligned, stride >= N*N*16), or with
 * out_base_device = NULL into library-owned maps N*N*16 bytes apart (ocean_batch_device_ptr, ocean_read_batch_displacement).
 * At N <= 1024, where one frame's two launches fill an eighth of the chip and the host needs 5-7 us to submit them, the K
 * frames are ONE launch pair (blockIdx.y = frame; K intermediates, allocated on demand) -- the reference keeps 3 frames in
 * flight by command-buffer rotation (src/lib.rs:86,150) and draws 4 instances of one map (src/render.rs:540-551,1360); at
 * N >= 2048 a frame fills the chip and the call is K ordinary launch pairs.  Every map is bit-identical to ocean_frame at the
 * same time.  Reference quirks only.  ocean_frame's own map is untouched. */
#define OCEAN_BATCH_MAX 64
/* ... and K independent TILES per launch pair (N <= 1024, fp32 spectra): ocean_context_create_tiles makes a fused-only context
 * that holds K tiles' static inputs (ocean_upload_spectrum_tile, k = 0 .. K-1; ocean_upload_spectrum = tile 0), and
 * ocean_frame_tiles(ctx, time, ...) computes the frame of every tile at `time` in ONE launch pair (blockIdx.y = tile), maps as in
 * ocean_frame_batch.  Tile k's map is bit-identical to ocean_frame on a context that holds that tile alone.  (ocean_frame and
 * ocean_frame_batch on such a context use tile 0.) */
int32_t ocean_context_create_tiles(int32_t device_ordinal, int32_t resolution, int32_t tiles, OceanContext** out_ctx);
int32_t ocean_context_tiles(const OceanContext* ctx);
int32_t ocean_upload_spectrum_tile(OceanContext* ctx, int32_t tile, const float* h0_re_im, const float* omega);
int32_t ocean_frame_tiles(OceanContext* ctx, float time, void* out_base_device, int64_t out_stride_bytes, void* stream);
int32_t ocean_frame_batch(OceanContext* ctx, float t0, float dt, int32_t count, void* out_base_device, int64_t out_stride_bytes,
                          void* stream);
void* ocean_batch_device_ptr(OceanContext* ctx);                    /* library-owned maps of the last NULL-buffer batch */
/* index < the frame count of that batch; beyond it (or before any such batch): OCEAN_E_STATE, never a frame of an earlier, larger batch */
int32_t ocean_read_batch_displacement(OceanContext* ctx, int32_t index, float* host_rgba /* N*N*4 */);
/* With the normal field switched on (ocean_set_frame_normals) a batch of N <= 1024 -- time steps or tiles -- carries it: K planes, and
 * ONE more launch for the K fields (library-owned, N*N*16 bytes apart; bit-identical to the single frame's).  Above 1024:
 * OCEAN_E_STATE (a batch there is K ordinary frames: call ocean_frame). */
void* ocean_batch_normals_device_ptr(OceanContext* ctx);
/* index < the frame count of the LAST batch, which must have carried the normal field (OCEAN_E_STATE otherwise: a batch launched with
 * the field switched off leaves none, whatever an earlier batch left in the buffer) */
int32_t ocean_read_batch_normals(OceanContext* ctx, int32_t index, float* host_xyz0 /* N*N*4 */);

/* the reference's normal field (shader/ocean.frag:50-66: finite differences of the
 * displacement map with Tile wrap, height_scale 180) as a compute pass over the current
 * displacement map.  source_channel 0 = disp_x (what the reference differentiates, quirk Q5),
 * 1 = height.  Result: float4[N*N] = (n.x, n.y, n.z, 0), read with ocean_read_normals. */
int32_t ocean_normals(OceanContext* ctx, int32_t source_channel, void* stream);
int32_t ocean_read_normals(OceanContext* ctx, float* host_xyz0 /* N*N*4 */);
/* The frame WITH its normal field as one workload (BASELINE config 3: "height + displacement + normal"; north_star lists the
 * normal field among the path's outputs).  source_channel 0..2 switches it on, -1 (the default) off.  While on, every frame
 * this context launches -- ocean_frame, ocean_frame_ex, the ocean_time_* / ocean_frame_times* loops -- is followed on the
 * same stream by the normal-field kernel: the fused pass 2 additionally stores the source channel as a dense fp32 plane
 * (4 B/texel, the very floats it writes into the map), and the kernel differentiates that plane instead of the RGBA texels
 * (4 + 16 instead of 16 + 16 B/texel); with non-reference quirks (staged dispatches) it reads the map like ocean_normals.
 * Same arithmetic, bit-identical normals either way.  Allocates N*N*20 bytes on first use. */
int32_t ocean_set_frame_normals(OceanContext* ctx, int32_t source_channel);
int32_t ocean_frame_normals(const OceanContext* ctx);               /* -1..2; < -1: error status */
void* ocean_normals_device_ptr(OceanContext* ctx);                  /* float4[N*N] in HBM, NULL before the first use */

/* the vertex stage's use of the map (shader/ocean.vert:21-25) as a compute pass: a verts x verts
 * patch grid (src/render.rs:494-508; the reference's HALF_RESOLUTION is 128) with a_Pos = (x, 0, z) and
 * a_Uv = (x, z) / (verts - 1); the displacement map sampled bilinearly with Tile wrap (src/render.rs:398),
 * y / 3.0, xz / 3.5, plus the patch offset (src/render.rs:540-551).  Result: float4[verts*verts] = (pos, 1),
 * index z * verts + x, read with ocean_read_positions. */
int32_t ocean_positions(OceanContext* ctx, int32_t verts, float offset_x, float offset_z, void* stream);
int32_t ocean_read_positions(OceanContext* ctx, float* host_xyz1 /* verts*verts*4 */);

/* Wait for everything this context has launched: its own stream, or -- once any dispatch was put on a caller stream -- the
 * whole device, like the readbacks and ocean_context_destroy (the reference never waits: src/render.rs:1068-1075). */
int32_t ocean_sync(OceanContext* ctx);

/* ---- readback / injection (the reference has none; needed for parity checks) -------------- */
int32_t ocean_read_displacement(OceanContext* ctx, float* host_rgba /* N*N*4 */);
int32_t ocean_read_field(OceanContext* ctx, int32_t field, float* host_re_im /* N*N*2 */);
int32_t ocean_write_field(OceanContext* ctx, int32_t field, const float* host_re_im);

/* ---- device-side consumers of the displacement map ----------------------------------------------- */
/* Order-independent 64-bit checksum of the current map (sum of (word + c) * (2 index + 1) over its 32-bit words,
 * mod 2^64), computed on the device behind whatever `stream` holds: equal maps <=> equal sums whatever the launch
 * geometry.  Used to assert that repeated frames are bit-identical without reading N*N*16 bytes back (SURVEY 5:
 * the race discipline the reference gets from its barrier chain, shader/fft_row.comp:48-59).  Synchronous. */
int32_t ocean_checksum_displacement(OceanContext* ctx, void* stream, uint64_t* out_sum);
/* Packed copies of the map for the final gather of a multi-GPU run (N*N*16 B per tile and frame
 * RGBA32F; 12 without the always-zero alpha of shader/correction.comp:31-34; 4 height only). */
#define OCEAN_PACK_RGBA32F 0  /* plain copy, 16 B/texel */
#define OCEAN_PACK_RGB32F 1   /* (disp_x, height, disp_z), 12 B/texel */
#define OCEAN_PACK_HEIGHT32F 2 /* height, 4 B/texel */
int64_t ocean_packed_bytes(const OceanContext* ctx, int32_t format);   /* < 0: error status */
int32_t ocean_pack_displacement(OceanContext* ctx, int32_t format, void* device_out /* 16-byte aligned */, void* stream);

/* ---- zero-copy hooks for device-side consumers ----------------------------------------------- */
void* ocean_displacement_device_ptr(OceanContext* ctx);           /* float4[N*N] in HBM */
int32_t ocean_bind_displacement(OceanContext* ctx, void* device_rgba); /* write frames into caller memory (NULL = own) */
/* The same for memory another API owns and has exported as a POSIX file descriptor: the VkDeviceMemory behind the reference's
 * `displacement_map` image (src/render.rs:820-869: Rgba32Sfloat, STORAGE | SAMPLED; allocate it linear and export it with
 * VK_KHR_external_memory_fd, handle type OPAQUE_FD), or a HIP allocation exported with hipMemExportToShareableHandle.  The library
 * imports the allocation (hipImportExternalMemory; on success the descriptor belongs to the runtime, as with Vulkan's own import),
 * maps the N*N*16 bytes at `offset_bytes` and writes every following frame straight into them: the sampler of ocean.vert:21 /
 * ocean.frag:56-59 reads what pass 2 stored, no copy.  ocean_bind_displacement(ctx, NULL or another pointer), another import or
 * ocean_context_destroy release the import (after waiting for the frames in flight). */
int32_t ocean_bind_displacement_fd(OceanContext* ctx, int32_t fd, uint64_t allocation_bytes, uint64_t offset_bytes);
void* ocean_stream(OceanContext* ctx);                            /* the context's hipStream_t */

/* ---- measurement (HIP events on the stream the kernels run on) -------------------------------- */
/* Runs `frames` frames (time = t0 + i*dt) on the context stream between two events; *out_ms = total. */
int32_t ocean_time_frames(OceanContext* ctx, int32_t frames, float t0, float dt, float* out_ms);
/* `launches` (<= 65536) calls of ocean_frame_batch(count frames, library-owned maps) back to back between two events: *out_ms. */
int32_t ocean_time_frame_batch(OceanContext* ctx, int32_t launches, int32_t count, float t0, float dt, float* out_ms);
/* `batches` (<= 4096) x `frames_per_batch` (<= 4096) fused frames back to back with one stream event between batches and one sync at
 * the end: batch_ms[b] = duration of batch b.  The distribution SURVEY 8d asks for (median, p10 / p90 of a frame in an
 * undisturbed loop); the reference's only timing is an EMA of the vsync-bound frame delta (src/lib.rs:146-148). */
int32_t ocean_time_frame_batches(OceanContext* ctx, int32_t batches, int32_t frames_per_batch, float t0, float dt,
                                 float* batch_ms);
/* Per-frame times of a back-to-back loop of `frames` (<= 4096) fused frames, from events bound to the dispatches themselves
 * (the kernels' own begin/end timestamps): pass1_ms[i] / pass2_ms[i] = the two kernels of frame i, period_ms[i] = begin of
 * frame i -> begin of frame i + 1 (the last entry: begin of pass 1 -> end of pass 2).  Any array may be NULL.  Per-KERNEL
 * distributions; the event-carrying launches leave larger gaps between kernels than plain ones (+5 % on the period at
 * N = 4096), so the frame's own distribution comes from ocean_time_frame_batches. */
int32_t ocean_frame_times(OceanContext* ctx, int32_t frames, float t0, float dt, float* pass1_ms, float* pass2_ms,
                          float* period_ms);
/* ... and normals_ms[i] = the normal-field kernel of frame i when the frame carries one (ocean_set_frame_normals; otherwise
 * normals_ms must be NULL: OCEAN_E_STATE); the period's last entry then ends with that kernel. */
int32_t ocean_frame_times_ex(OceanContext* ctx, int32_t frames, float t0, float dt, float* pass1_ms, float* pass2_ms,
                             float* normals_ms, float* period_ms);
/* Per-kernel durations of ONE frame (begin/end timestamps of each dispatch, as rocprofv3 reports them; the
 * frame runs behind two untimed ones): names/ms arrays of capacity `cap`; returns count via *out_n. */
int32_t ocean_profile_frame(OceanContext* ctx, float time, int32_t cap, const char** names, float* ms,
                            int32_t* out_n);
/* Same for the staged 8-dispatch path (propagate, 3 rows, 3 cols, correct). */
int32_t ocean_profile_staged(OceanContext* ctx, float time, int32_t cap, const char** names, float* ms,
                             int32_t* out_n);

/* ---- one N x N tile sharded over the GPUs of a node ------------------------------------------------
 * The reference runs one 512 x 512 transform on one GPU; the only ordering it imposes is "all row passes, barrier,
 * all column passes" (src/render.rs:1158-1231).  For tiles too large or too slow for one GPU (N up to 16384) the same
 * chain is sharded by ROW BLOCKS: rank r of `world` owns rows [r N/world, (r+1) N/world) of the three spectra,
 *     ocean_shard_rows:  propagate (shader/propagate.comp:42-72) + row pass (shader/fft_row.comp:44-63) on its rows,
 *                        written as the send buffer of ONE all-to-all:  send[dest][field][row][column of dest];
 *     (caller)           all-to-all over xGMI (RCCL: torch.distributed.all_to_all_single, gfx_ocean_amd/sharded.py),
 *                        3 N^2 8 / world bytes per rank and frame;
 *     ocean_shard_cols:  recv[src][field][row of src][own column] -> column pass (shader/fft_col.comp:44-63) and
 *                        correction (shader/correction.comp:24-35) on its N/world columns.
 * The result is the rank's COLUMN block, transposed: out[(x - rank N/world) * N + y] = (disp_x, height, disp_z, 0).
 * Fields are ordered dx, dy, dz (OCEAN_FIELD_*), complex fp32.  Reference quirks only.  One OceanShard per GPU. */
typedef struct OceanShard OceanShard;
int32_t ocean_shard_create(int32_t device_ordinal, int32_t resolution /* 512 .. 16384 */, int32_t rank, int32_t world,
                           OceanShard** out);
void ocean_shard_destroy(OceanShard* shard);
const char* ocean_shard_last_error(const OceanShard* shard);      /* shard may be NULL: last error of a failed create */
/* The rank's static inputs (host pointers, rows of N texels): its own rows of h0 and omega, and the rows
 * [N - (rank+1) N/world, N - rank N/world) of h0 in which the "-k" partners of its texels live (propagate.comp:48). */
int32_t ocean_shard_upload(OceanShard* shard, const float* h0_own_rows, const float* h0_partner_rows,
                           const float* omega_own_rows);
int32_t ocean_shard_rows(OceanShard* shard, const OceanPropagateLocals* locals, void* send_device, void* stream);
int32_t ocean_shard_cols(OceanShard* shard, const void* recv_device, void* out_rgba_T_device, void* stream);
int32_t ocean_shard_sync(OceanShard* shard);
void* ocean_shard_stream(OceanShard* shard);

/* ---- one tile over several GPUs, second generation: the fused half-spectrum frame, sharded ------------------------------
 * The fused frame transforms columns first (ocean_frame: pass 1 = propagate + column transform of the N/2 distinct
 * columns of the symmetrised spectra, pass 2 = row transform + correction), so the sharded frame is
 *     ocean_tile_pass1:  rank r transforms the half-spectrum columns [r N/(2 world), (r+1) N/(2 world)) and writes them as
 *                        the send buffer of ONE all-to-all,  send[dest][...] = the rows of rank `dest`;
 *     (caller)           all-to-all over xGMI: 3 * N/2 * N/world * 8 bytes per rank and frame (ocean_tile_exchange_bytes)
 *                        -- half of what the row-block scheme above ships: the Hermitian half is enough;
 *     ocean_tile_pass2:  rank r rebuilds its rows [r N/world, (r+1) N/world) from recv[src][...], transforms them and
 *                        writes them in the NATURAL orientation: out_rows[(y - r N/world) * N + x] = (disp_x, height, disp_z, 0).
 * Both run on an ordinary context that holds the whole tile's static inputs (ocean_upload_spectrum on every rank: the
 * inputs are static and 12 bytes per texel): no state per rank, so all ranks of a tile can be driven from one context
 * on one GPU (tests) or one context per GPU (gfx_ocean_amd/sharded.py).  N = 256 .. 16384 (16384: every line as two
 * interleaved 8192-point transforms, one column per pass-1 workgroup; also what ocean_frame runs there), world a power of
 * two with at least 32 rows per rank.  Same barrier as the reference's between its row and column dispatches (src/render.rs:1181-1208). */
int64_t ocean_tile_exchange_bytes(const OceanContext* ctx, int32_t world);   /* bytes of a rank's send (= receive) buffer; < 0: error */
/* `parts` (a power of two, 1 = no pipelining) cuts the rank's column block -- and the exchange -- into that many pieces:
 * pass 1 of piece `part` fills send_part_device (ocean_tile_exchange_bytes / parts bytes, [dest][...]), the caller ships
 * it with its own all-to-all into piece `part` of the receive buffer (recv[part][src][...]) while pass 1 of the next piece
 * runs, and pass 2 consumes the whole receive buffer. */
int32_t ocean_tile_pass1(OceanContext* ctx, const OceanPropagateLocals* locals, int32_t rank, int32_t world, int32_t part,
                         int32_t parts, void* send_part_device, void* stream);
int32_t ocean_tile_pass2(OceanContext* ctx, int32_t rank, int32_t world, int32_t parts, const void* recv_device,
                         void* out_rows_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_HIP_H */
