//! Raw bindings of include/ocean_hip.h (one line per exported symbol).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct OceanContext { _p: [u8; 0] }
#[repr(C)]
pub struct OceanFft { _p: [u8; 0] }
#[repr(C)]
pub struct OceanPropagation { _p: [u8; 0] }
#[repr(C)]
pub struct OceanCorrection { _p: [u8; 0] }
#[repr(C)]
pub struct OceanShard { _p: [u8; 0] }

/// src/ocean.rs:8-13 -- here with the explicit layout the reference relies on by accident (Q6).
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct OceanPropagateLocals { pub time: f32, pub resolution: i32, pub domain_size: f32 }
/// src/ocean.rs:179-182
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct OceanCorrectionLocals { pub resolution: u32 }

extern "C" {
    pub fn ocean_abi_version() -> i32;
    pub fn ocean_device_count() -> i32;
    pub fn ocean_device_pci_bus_id(device: i32, out: *mut c_char, capacity: i32) -> i32;
    pub fn ocean_context_create(device: i32, resolution: i32, out: *mut *mut OceanContext) -> i32;
    pub fn ocean_context_create_ex(device: i32, resolution: i32, flags: u32, out: *mut *mut OceanContext) -> i32;
    pub fn ocean_context_create_tile_rank(device: i32, resolution: i32, rank: i32, world: i32, out: *mut *mut OceanContext) -> i32;
    pub fn ocean_context_flags(ctx: *const OceanContext) -> u32;
    pub fn ocean_context_destroy(ctx: *mut OceanContext);
    pub fn ocean_last_error(ctx: *const OceanContext) -> *const c_char;
    pub fn ocean_resolution(ctx: *const OceanContext) -> i32;
    pub fn ocean_upload_spectrum(ctx: *mut OceanContext, h0_re_im: *const f32, omega: *const f32) -> i32;
    pub fn ocean_upload_spectrum_f16(ctx: *mut OceanContext, h0_re_im: *const f32, omega: *const f32) -> i32;
    pub fn ocean_spectrum_scale_log2(ctx: *const OceanContext) -> i32;
    pub fn ocean_read_spectrum(ctx: *mut OceanContext, host_re_im: *mut f32) -> i32;
    pub fn ocean_fft_init(ctx: *mut OceanContext, out: *mut *mut OceanFft) -> i32;
    pub fn ocean_fft_destroy(fft: *mut OceanFft);
    pub fn ocean_propagation_init(ctx: *mut OceanContext, out: *mut *mut OceanPropagation) -> i32;
    pub fn ocean_propagation_destroy(p: *mut OceanPropagation);
    pub fn ocean_correction_init(ctx: *mut OceanContext, out: *mut *mut OceanCorrection) -> i32;
    pub fn ocean_correction_destroy(c: *mut OceanCorrection);
    pub fn ocean_propagate(p: *mut OceanPropagation, locals: *const OceanPropagateLocals, stream: *mut c_void) -> i32;
    pub fn ocean_fft_rows(fft: *mut OceanFft, field: i32, stream: *mut c_void) -> i32;
    pub fn ocean_fft_cols(fft: *mut OceanFft, field: i32, stream: *mut c_void) -> i32;
    pub fn ocean_correct(c: *mut OceanCorrection, locals: *const OceanCorrectionLocals, stream: *mut c_void) -> i32;
    pub fn ocean_frame(ctx: *mut OceanContext, time: f32, stream: *mut c_void) -> i32;
    pub fn ocean_frame_ex(ctx: *mut OceanContext, locals: *const OceanPropagateLocals, stream: *mut c_void) -> i32;
    pub fn ocean_context_create_tiles(device: i32, resolution: i32, tiles: i32, out: *mut *mut OceanContext) -> i32;
    pub fn ocean_context_tiles(ctx: *const OceanContext) -> i32;
    pub fn ocean_upload_spectrum_tile(ctx: *mut OceanContext, tile: i32, h0_re_im: *const f32, omega: *const f32) -> i32;
    pub fn ocean_upload_spectrum_device(ctx: *mut OceanContext, tile: i32, h0_device: *const c_void, omega_device: *const c_void,
                                        stream: *mut c_void) -> i32;
    pub fn ocean_frame_tiles(ctx: *mut OceanContext, time: f32, out_base_device: *mut c_void, out_stride_bytes: i64, stream: *mut c_void) -> i32;
    pub fn ocean_frame_batch(ctx: *mut OceanContext, t0: f32, dt: f32, count: i32, out_base_device: *mut c_void, out_stride_bytes: i64,
                             stream: *mut c_void) -> i32;
    pub fn ocean_batch_device_ptr(ctx: *mut OceanContext) -> *mut c_void;
    pub fn ocean_batch_normals_device_ptr(ctx: *mut OceanContext) -> *mut c_void;
    pub fn ocean_read_batch_normals(ctx: *mut OceanContext, index: i32, host_xyz0: *mut f32) -> i32;
    pub fn ocean_read_batch_displacement(ctx: *mut OceanContext, index: i32, host_rgba: *mut f32) -> i32;
    pub fn ocean_time_frame_batch(ctx: *mut OceanContext, launches: i32, count: i32, t0: f32, dt: f32, out_ms: *mut f32) -> i32;
    pub fn ocean_normals(ctx: *mut OceanContext, source_channel: i32, stream: *mut c_void) -> i32;
    pub fn ocean_set_frame_normals(ctx: *mut OceanContext, source_channel: i32) -> i32;
    pub fn ocean_frame_normals(ctx: *const OceanContext) -> i32;
    pub fn ocean_normals_device_ptr(ctx: *mut OceanContext) -> *mut c_void;
    pub fn ocean_read_normals(ctx: *mut OceanContext, host_xyz0: *mut f32) -> i32;
    pub fn ocean_positions(ctx: *mut OceanContext, verts: i32, offset_x: f32, offset_z: f32, stream: *mut c_void) -> i32;
    pub fn ocean_read_positions(ctx: *mut OceanContext, host_xyz1: *mut f32) -> i32;
    pub fn ocean_sync(ctx: *mut OceanContext) -> i32;
    pub fn ocean_checksum_displacement(ctx: *mut OceanContext, stream: *mut c_void, out_sum: *mut u64) -> i32;
    pub fn ocean_packed_bytes(ctx: *const OceanContext, format: i32) -> i64;
    pub fn ocean_pack_displacement(ctx: *mut OceanContext, format: i32, device_out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ocean_set_quirks(ctx: *mut OceanContext, quirks: u32) -> i32;
    pub fn ocean_quirks(ctx: *const OceanContext) -> u32;
    pub fn ocean_set_intermediate(ctx: *mut OceanContext, mode: i32) -> i32;
    pub fn ocean_intermediate(ctx: *const OceanContext) -> i32;
    pub fn ocean_read_displacement(ctx: *mut OceanContext, host_rgba: *mut f32) -> i32;
    pub fn ocean_read_field(ctx: *mut OceanContext, field: i32, host_re_im: *mut f32) -> i32;
    pub fn ocean_write_field(ctx: *mut OceanContext, field: i32, host_re_im: *const f32) -> i32;
    pub fn ocean_displacement_device_ptr(ctx: *mut OceanContext) -> *mut c_void;
    pub fn ocean_bind_displacement(ctx: *mut OceanContext, device_rgba: *mut c_void) -> i32;
    pub fn ocean_bind_displacement_fd(ctx: *mut OceanContext, fd: i32, allocation_bytes: u64, offset_bytes: u64) -> i32;
    pub fn ocean_stream(ctx: *mut OceanContext) -> *mut c_void;
    pub fn ocean_time_frames(ctx: *mut OceanContext, frames: i32, t0: f32, dt: f32, out_ms: *mut f32) -> i32;
    pub fn ocean_time_frame_batches(ctx: *mut OceanContext, batches: i32, frames_per_batch: i32, t0: f32, dt: f32, batch_ms: *mut f32) -> i32;
    pub fn ocean_frame_times(ctx: *mut OceanContext, frames: i32, t0: f32, dt: f32, pass1_ms: *mut f32, pass2_ms: *mut f32, period_ms: *mut f32) -> i32;
    pub fn ocean_frame_times_ex(ctx: *mut OceanContext, frames: i32, t0: f32, dt: f32, pass1_ms: *mut f32, pass2_ms: *mut f32, normals_ms: *mut f32,
                                period_ms: *mut f32) -> i32;
    pub fn ocean_profile_frame(ctx: *mut OceanContext, time: f32, cap: i32, names: *mut *const c_char,
                               ms: *mut f32, out_n: *mut i32) -> i32;
    pub fn ocean_profile_staged(ctx: *mut OceanContext, time: f32, cap: i32, names: *mut *const c_char,
                                ms: *mut f32, out_n: *mut i32) -> i32;

    // ---- one N x N tile sharded by row blocks over the GPUs of a node (include/ocean_hip.h "sharded tile") ----
    pub fn ocean_shard_create(device_ordinal: i32, resolution: i32, rank: i32, world: i32, out: *mut *mut OceanShard) -> i32;
    pub fn ocean_shard_destroy(shard: *mut OceanShard);
    pub fn ocean_shard_last_error(shard: *const OceanShard) -> *const c_char;
    pub fn ocean_shard_upload(shard: *mut OceanShard, h0_own_rows: *const f32, h0_partner_rows: *const f32,
                              omega_own_rows: *const f32) -> i32;
    pub fn ocean_shard_rows(shard: *mut OceanShard, locals: *const OceanPropagateLocals, send_device: *mut c_void,
                            stream: *mut c_void) -> i32;
    pub fn ocean_shard_cols(shard: *mut OceanShard, recv_device: *const c_void, out_rgba_t_device: *mut c_void,
                            stream: *mut c_void) -> i32;
    pub fn ocean_shard_sync(shard: *mut OceanShard) -> i32;
    pub fn ocean_shard_stream(shard: *mut OceanShard) -> *mut c_void;
    pub fn ocean_tile_exchange_bytes(ctx: *const OceanContext, world: i32) -> i64;
    pub fn ocean_tile_pass1(ctx: *mut OceanContext, locals: *const OceanPropagateLocals, rank: i32, world: i32, part: i32, parts: i32,
                            send_part_device: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ocean_tile_pass2(ctx: *mut OceanContext, rank: i32, world: i32, parts: i32, recv_device: *const c_void, out_rows_device: *mut c_void,
                            stream: *mut c_void) -> i32;
}
