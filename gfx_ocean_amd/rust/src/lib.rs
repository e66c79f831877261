//! `ocean_hip`: the reference's `mod ocean` / `mod fft` names over the MI355X C ABI.
//!
//! In gfx-ocean, `src/render.rs:223-225` becomes
//! ```ignore
//! let device = ocean_hip::Device::new(0, RESOLUTION as i32)?;      // replaces the gfx_hal buffers of :607-670
//! let mut fft = ocean_hip::fft::Fft::init(&device)?;
//! let mut propagation = ocean_hip::ocean::Propagation::init(&device)?;
//! let mut correction = ocean_hip::ocean::Correction::init(&device)?;
//! device.upload_spectrum(&spectrum, &omega)?;                      // replaces :742-924
//! ```
//! and the dispatch block `:1101-1310` becomes
//! ```ignore
//! propagation.dispatch(&PropagateLocals { time, resolution: RESOLUTION as i32, domain_size: DOMAIN_SIZE })?;
//! fft.row_pass(fft::FIELD_ALL)?;
//! fft.col_pass(fft::FIELD_ALL)?;
//! correction.dispatch(&CorrectionLocals { resolution: RESOLUTION as u32 })?;
//! // or, fused:  device.frame(time)?;
//! ```
pub mod ffi;

use std::error::Error;
use std::ffi::CStr;
use std::fmt;
use std::ptr;

/// SURVEY 8a quirk switches (include/ocean_hip.h `OCEAN_QUIRK_*`).
pub const QUIRK_Q1_UINT_WAVE_INDEX: u32 = 1;
pub const QUIRK_Q2_MIRROR_NO_CONJ: u32 = 2;
pub const QUIRKS_REFERENCE: u32 = 3;
/// Precision of the intermediate (include/ocean_hip.h `OCEAN_INTER_*`).
pub const INTER_F32: i32 = 0;
pub const INTER_BFP16: i32 = 1;

#[derive(Debug)]
pub struct OceanError { pub status: i32, pub message: String }
impl fmt::Display for OceanError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result { write!(f, "ocean_hip status {}: {}", self.status, self.message) }
}
impl Error for OceanError {}

/// One GPU + the buffers of the path (the slice of `Renderer` in src/render.rs:72-101 the compute path owns).
pub struct Device { ctx: *mut ffi::OceanContext }

impl Device {
    pub fn new(ordinal: i32, resolution: i32) -> Result<Self, Box<dyn Error>> {
        let mut ctx = ptr::null_mut();
        let st = unsafe { ffi::ocean_context_create(ordinal, resolution, &mut ctx) };
        if st != 0 { return Err(Box::new(error(ptr::null(), st))); }
        Ok(Device { ctx })
    }
    /// A context with only the buffers `frame` needs (include/ocean_hip.h `OCEAN_CTX_FUSED_ONLY`: 40 instead of 76-100 bytes per
    /// texel); the staged stage objects then fail with `OCEAN_E_STATE`.
    pub fn fused_only(ordinal: i32, resolution: i32) -> Result<Self, Box<dyn Error>> {
        let mut ctx = ptr::null_mut();
        let st = unsafe { ffi::ocean_context_create_ex(ordinal, resolution, 1, &mut ctx) };
        if st != 0 { return Err(Box::new(error(ptr::null(), st))); }
        Ok(Device { ctx })
    }
    /// `tiles` independent tiles' static inputs in one context (N <= 1024): `upload_spectrum_tile` each, then `frame_tiles` computes
    /// one frame of every tile in one launch pair (include/ocean_hip.h `ocean_context_create_tiles`).
    pub fn with_tiles(ordinal: i32, resolution: i32, tiles: i32) -> Result<Self, Box<dyn Error>> {
        let mut ctx = ptr::null_mut();
        let st = unsafe { ffi::ocean_context_create_tiles(ordinal, resolution, tiles, &mut ctx) };
        if st != 0 { return Err(Box::new(error(ptr::null(), st))); }
        Ok(Device { ctx })
    }
    pub fn upload_spectrum_tile(&self, tile: i32, spectrum: &[[f32; 2]], omega: &[f32]) -> Result<(), Box<dyn Error>> {
        let n2 = self.texels()?;
        if spectrum.len() != n2 || omega.len() != n2 {
            return Err(Box::new(OceanError { status: -1, message: format!(
                "upload_spectrum_tile: expected {} texels, got {} and {}", n2, spectrum.len(), omega.len()) }));
        }
        self.check(unsafe { ffi::ocean_upload_spectrum_tile(self.ctx, tile, spectrum.as_ptr() as *const f32, omega.as_ptr()) })
    }
    /// The upload's device-side half alone (`copy_buffer`, src/render.rs:896-915): the spectrum is read from memory the GPU can
    /// read (device, managed or registered host memory), asynchronously on `stream` (null = the context's).
    /// # Safety
    /// `h0_device` / `omega_device` must address N*N `[f32; 2]` / N*N `f32` that stay valid until the stream has passed the call.
    pub unsafe fn upload_spectrum_device(&self, tile: i32, h0_device: *const std::ffi::c_void, omega_device: *const std::ffi::c_void,
                                         stream: *mut std::ffi::c_void) -> Result<(), Box<dyn Error>> {
        self.check(ffi::ocean_upload_spectrum_device(self.ctx, tile, h0_device, omega_device, stream))
    }
    /// One frame of every tile at `time` into library-owned maps (`read_batch_displacement(k, ..)`).
    pub fn frame_tiles(&self, time: f32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_frame_tiles(self.ctx, time, ptr::null_mut(), 0, ptr::null_mut()) })
    }
    fn check(&self, st: i32) -> Result<(), Box<dyn Error>> {
        if st == 0 { Ok(()) } else { Err(Box::new(error(self.ctx, st))) }
    }
    /// bincode-decoded `Vec<[f32; 2]>` / `Vec<f32>` exactly as src/render.rs:769-771, :808-810 produce them.
    /// The C ABI reads exactly N*N complex and N*N real values: anything else is rejected here, so the safe
    /// function cannot read past a short slice.
    pub fn upload_spectrum(&self, spectrum: &[[f32; 2]], omega: &[f32]) -> Result<(), Box<dyn Error>> {
        let n2 = self.texels()?;
        if spectrum.len() != n2 || omega.len() != n2 {
            return Err(Box::new(OceanError { status: -1, message: format!(
                "upload_spectrum: expected {} spectrum and omega entries, got {} and {}", n2, spectrum.len(), omega.len()) }));
        }
        self.check(unsafe { ffi::ocean_upload_spectrum(self.ctx, spectrum.as_ptr() as *const f32, omega.as_ptr()) })
    }
    /// Resolution of the context (the reference's RESOLUTION, src/render.rs:44).
    pub fn resolution(&self) -> Result<usize, Box<dyn Error>> {
        let n = unsafe { ffi::ocean_resolution(self.ctx) };
        if n <= 0 { return Err(Box::new(error(self.ctx, n))); }
        Ok(n as usize)
    }
    fn texels(&self) -> Result<usize, Box<dyn Error>> { let n = self.resolution()?; Ok(n * n) }
    pub fn frame(&self, time: f32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_frame(self.ctx, time, ptr::null_mut()) })
    }
    /// `count` time steps `t0 + dt * i` of this tile into library-owned maps (one launch pair at N <= 1024:
    /// include/ocean_hip.h `ocean_frame_batch`); read them with `read_batch_displacement`.
    pub fn frame_batch(&self, t0: f32, dt: f32, count: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_frame_batch(self.ctx, t0, dt, count, ptr::null_mut(), 0, ptr::null_mut()) })
    }
    pub fn read_batch_displacement(&self, index: i32, rgba: &mut [f32]) -> Result<(), Box<dyn Error>> {
        let want = self.texels()? * 4;
        if rgba.len() != want {
            return Err(Box::new(OceanError { status: -1, message: format!(
                "read_batch_displacement: expected a slice of {} floats, got {}", want, rgba.len()) }));
        }
        self.check(unsafe { ffi::ocean_read_batch_displacement(self.ctx, index, rgba.as_mut_ptr()) })
    }
    /// The frame with its normal field as one workload (shader/ocean.frag:50-66; channel 0 = disp_x as the reference, quirk Q5):
    /// `Some(channel)` switches it on for every following frame, `None` off (include/ocean_hip.h `ocean_set_frame_normals`).
    pub fn set_frame_normals(&self, source_channel: Option<i32>) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_set_frame_normals(self.ctx, source_channel.unwrap_or(-1)) })
    }
    /// N*N float4 (n.x, n.y, n.z, 0) of the last frame's normal field.
    pub fn read_normals(&self, xyz0: &mut [f32]) -> Result<(), Box<dyn Error>> {
        let want = self.texels()? * 4;
        if xyz0.len() != want {
            return Err(Box::new(OceanError { status: -1, message: format!(
                "read_normals: expected a slice of {} floats, got {}", want, xyz0.len()) }));
        }
        self.check(unsafe { ffi::ocean_read_normals(self.ctx, xyz0.as_mut_ptr()) })
    }
    /// Milliseconds of `frames` back-to-back frames between two HIP events on the context stream.
    pub fn time_frames(&self, frames: i32, t0: f32, dt: f32) -> Result<f32, Box<dyn Error>> {
        let mut ms = 0f32;
        self.check(unsafe { ffi::ocean_time_frames(self.ctx, frames, t0, dt, &mut ms) })?;
        Ok(ms)
    }
    /// Per-batch milliseconds of `batches` x `frames_per_batch` back-to-back frames (the frame-time distribution).
    pub fn time_frame_batches(&self, batches: i32, frames_per_batch: i32, t0: f32, dt: f32) -> Result<Vec<f32>, Box<dyn Error>> {
        let mut ms = vec![0f32; batches.max(0) as usize];
        self.check(unsafe { ffi::ocean_time_frame_batches(self.ctx, batches, frames_per_batch, t0, dt, ms.as_mut_ptr()) })?;
        Ok(ms)
    }
    /// (pass 1, pass 2, period) milliseconds of every frame of a back-to-back loop, from events bound to the dispatches.
    pub fn frame_times(&self, frames: i32, t0: f32, dt: f32) -> Result<(Vec<f32>, Vec<f32>, Vec<f32>), Box<dyn Error>> {
        let k = frames.max(0) as usize;
        let (mut a, mut b, mut c) = (vec![0f32; k], vec![0f32; k], vec![0f32; k]);
        self.check(unsafe { ffi::ocean_frame_times(self.ctx, frames, t0, dt, a.as_mut_ptr(), b.as_mut_ptr(), c.as_mut_ptr()) })?;
        Ok((a, b, c))
    }
    /// Render every following frame straight into device memory another API exported as a file descriptor -- the `VkDeviceMemory`
    /// behind `displacement_map` (src/render.rs:820-869), exported with VK_KHR_external_memory_fd; the descriptor is consumed.
    pub fn bind_displacement_fd(&self, fd: std::os::unix::io::RawFd, allocation_bytes: u64, offset_bytes: u64) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_bind_displacement_fd(self.ctx, fd, allocation_bytes, offset_bytes) })
    }
    /// SURVEY 8a Q1/Q2 switches; `QUIRKS_REFERENCE` (default) is the shipped shaders' arithmetic.
    pub fn set_quirks(&self, quirks: u32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_set_quirks(self.ctx, quirks) })
    }
    /// Writes N*N RGBA32F texels; the slice must hold exactly N*N*4 floats.
    pub fn read_displacement(&self, rgba: &mut [f32]) -> Result<(), Box<dyn Error>> {
        let want = self.texels()? * 4;
        if rgba.len() != want {
            return Err(Box::new(OceanError { status: -1, message: format!(
                "read_displacement: expected a slice of {} floats, got {}", want, rgba.len()) }));
        }
        self.check(unsafe { ffi::ocean_read_displacement(self.ctx, rgba.as_mut_ptr()) })
    }
    /// Opt-in precision of the intermediate between the two fused launches: `INTER_F32` (default) or `INTER_BFP16`
    /// (int16 mantissas + block scales, N = 8192; include/ocean_hip.h `OCEAN_INTER_*`).
    pub fn set_intermediate(&self, mode: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { ffi::ocean_set_intermediate(self.ctx, mode) })
    }
    /// Order-independent 64-bit checksum of the current displacement map, computed on the device.
    pub fn checksum(&self) -> Result<u64, Box<dyn Error>> {
        let mut sum = 0u64;
        self.check(unsafe { ffi::ocean_checksum_displacement(self.ctx, ptr::null_mut(), &mut sum) })?;
        Ok(sum)
    }
    /// One tile over several GPUs (fused scheme): pass 1 of piece `part` of `parts` of rank `rank`'s half-spectrum columns
    /// into the caller's device send buffer, and pass 2 of its rows from the receive buffer (see include/ocean_hip.h).
    /// The pointers are device addresses owned by the caller (e.g. RCCL buffers): hence `unsafe`.
    pub unsafe fn tile_pass1(&self, locals: &ocean::PropagateLocals, rank: i32, world: i32, part: i32, parts: i32,
                             send_part_device: *mut std::ffi::c_void, stream: *mut std::ffi::c_void) -> Result<(), Box<dyn Error>> {
        self.check(ffi::ocean_tile_pass1(self.ctx, locals, rank, world, part, parts, send_part_device, stream))
    }
    pub unsafe fn tile_pass2(&self, rank: i32, world: i32, parts: i32, recv_device: *const std::ffi::c_void,
                             out_rows_device: *mut std::ffi::c_void, stream: *mut std::ffi::c_void) -> Result<(), Box<dyn Error>> {
        self.check(ffi::ocean_tile_pass2(self.ctx, rank, world, parts, recv_device, out_rows_device, stream))
    }
    pub fn raw(&self) -> *mut ffi::OceanContext { self.ctx }
}
impl Drop for Device { fn drop(&mut self) { unsafe { ffi::ocean_context_destroy(self.ctx) } } }

fn error(ctx: *const ffi::OceanContext, status: i32) -> OceanError {
    let message = unsafe { CStr::from_ptr(ffi::ocean_last_error(ctx)) }.to_string_lossy().into_owned();
    OceanError { status, message }
}

pub mod ocean {
    //! src/ocean.rs
    use super::*;
    pub use crate::ffi::OceanCorrectionLocals as CorrectionLocals;
    pub use crate::ffi::OceanPropagateLocals as PropagateLocals;

    /// src/ocean.rs:15-177
    pub struct Propagation<'d> { h: *mut ffi::OceanPropagation, device: &'d Device }
    impl<'d> Propagation<'d> {
        pub fn init(device: &'d Device) -> Result<Self, Box<dyn Error>> {
            let mut h = ptr::null_mut();
            device.check(unsafe { ffi::ocean_propagation_init(device.raw(), &mut h) })?;
            Ok(Propagation { h, device })
        }
        /// bind pipeline + descriptor set + dispatch [N/16, N/16, 1] (src/render.rs:1101-1130)
        pub fn dispatch(&mut self, locals: &PropagateLocals) -> Result<(), Box<dyn Error>> {
            self.device.check(unsafe { ffi::ocean_propagate(self.h, locals, ptr::null_mut()) })
        }
        pub fn destroy(self) { unsafe { ffi::ocean_propagation_destroy(self.h) } }
    }

    /// src/ocean.rs:184-328
    pub struct Correction<'d> { h: *mut ffi::OceanCorrection, device: &'d Device }
    impl<'d> Correction<'d> {
        pub fn init(device: &'d Device) -> Result<Self, Box<dyn Error>> {
            let mut h = ptr::null_mut();
            device.check(unsafe { ffi::ocean_correction_init(device.raw(), &mut h) })?;
            Ok(Correction { h, device })
        }
        /// src/render.rs:1280-1287
        pub fn dispatch(&mut self, locals: &CorrectionLocals) -> Result<(), Box<dyn Error>> {
            self.device.check(unsafe { ffi::ocean_correct(self.h, locals, ptr::null_mut()) })
        }
        pub fn destroy(self) { unsafe { ffi::ocean_correction_destroy(self.h) } }
    }
}

pub mod fft {
    //! src/fft.rs
    use super::*;
    /// desc_sets[0,1,2] -> dx_spec, dy_spec, dz_spec (src/render.rs:971-988)
    pub const FIELD_DX: i32 = 0;
    pub const FIELD_DY: i32 = 1;
    pub const FIELD_DZ: i32 = 2;
    pub const FIELD_ALL: i32 = -1;

    /// src/fft.rs:7-111
    pub struct Fft<'d> { h: *mut ffi::OceanFft, device: &'d Device }
    impl<'d> Fft<'d> {
        pub fn init(device: &'d Device) -> Result<Self, Box<dyn Error>> {
            let mut h = ptr::null_mut();
            device.check(unsafe { ffi::ocean_fft_init(device.raw(), &mut h) })?;
            Ok(Fft { h, device })
        }
        /// src/render.rs:1158-1179
        pub fn row_pass(&mut self, field: i32) -> Result<(), Box<dyn Error>> {
            self.device.check(unsafe { ffi::ocean_fft_rows(self.h, field, ptr::null_mut()) })
        }
        /// src/render.rs:1210-1231
        pub fn col_pass(&mut self, field: i32) -> Result<(), Box<dyn Error>> {
            self.device.check(unsafe { ffi::ocean_fft_cols(self.h, field, ptr::null_mut()) })
        }
        pub fn destroy(self) { unsafe { ffi::ocean_fft_destroy(self.h) } }
    }
}

pub mod shard {
    //! One N x N tile over several GPUs (include/ocean_hip.h "sharded tile"; no counterpart in the reference, whose
    //! only ordering constraint -- all row passes, barrier, all column passes, src/render.rs:1158-1231 -- becomes the
    //! all-to-all the caller runs between `rows` and `cols` (RCCL `ncclSend`/`ncclRecv` group, or any other transport).
    use super::*;
    use std::os::raw::c_void;

    pub struct Shard { h: *mut ffi::OceanShard, n: usize, rows: usize }

    impl Shard {
        pub fn new(device: i32, resolution: i32, rank: i32, world: i32) -> Result<Self, Box<dyn Error>> {
            let mut h = ptr::null_mut();
            let st = unsafe { ffi::ocean_shard_create(device, resolution, rank, world, &mut h) };
            if st != 0 { return Err(Box::new(shard_error(ptr::null(), st))); }
            Ok(Shard { h, n: resolution as usize, rows: (resolution / world) as usize })
        }
        fn check(&self, st: i32) -> Result<(), Box<dyn Error>> {
            if st == 0 { Ok(()) } else { Err(Box::new(shard_error(self.h, st))) }
        }
        /// The rank's rows of h0 and omega, and the opposite row block of h0 (where its "-k" partners live).
        pub fn upload(&self, h0_own: &[[f32; 2]], h0_partner: &[[f32; 2]], omega_own: &[f32]) -> Result<(), Box<dyn Error>> {
            let block = self.rows * self.n;
            if h0_own.len() != block || h0_partner.len() != block || omega_own.len() != block {
                return Err(Box::new(OceanError { status: -1, message: format!("upload: expected {} entries per block", block) }));
            }
            self.check(unsafe { ffi::ocean_shard_upload(self.h, h0_own.as_ptr() as *const f32, h0_partner.as_ptr() as *const f32, omega_own.as_ptr()) })
        }
        /// Propagate + row pass on the rank's rows into `send` (device memory, [world][3][rows][cols] complex fp32).
        /// # Safety
        /// `send` must be a device allocation of 3 * rows * N complex fp32 values; `stream` a hipStream_t or null.
        pub unsafe fn rows(&self, locals: &ffi::OceanPropagateLocals, send: *mut c_void, stream: *mut c_void) -> Result<(), Box<dyn Error>> {
            self.check(ffi::ocean_shard_rows(self.h, locals, send, stream))
        }
        /// Column pass + correction on the received block; `out_t` = float4[(N / world) * N], column block transposed.
        /// # Safety
        /// `recv` and `out_t` must be device allocations of the sizes above, `out_t` 16-byte aligned.
        pub unsafe fn cols(&self, recv: *const c_void, out_t: *mut c_void, stream: *mut c_void) -> Result<(), Box<dyn Error>> {
            self.check(ffi::ocean_shard_cols(self.h, recv, out_t, stream))
        }
        pub fn sync(&self) -> Result<(), Box<dyn Error>> { self.check(unsafe { ffi::ocean_shard_sync(self.h) }) }
    }
    impl Drop for Shard { fn drop(&mut self) { unsafe { ffi::ocean_shard_destroy(self.h) } } }

    fn shard_error(h: *const ffi::OceanShard, status: i32) -> OceanError {
        let message = unsafe { CStr::from_ptr(ffi::ocean_shard_last_error(h)) }.to_string_lossy().into_owned();
        OceanError { status, message }
    }
}
