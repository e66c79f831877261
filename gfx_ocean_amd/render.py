"""The compute slice of the reference's `Renderer` (src/render.rs): device + buffers
(`OceanDevice`) and the per-frame recorder (`OceanRenderer.render`, src/render.rs:1101-1310)."""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import OCEAN_OK, OceanError, load_library
from .fft import FIELD_ALL, FIELD_DX, FIELD_DY, FIELD_DZ, Fft

QUIRK_Q1, QUIRK_Q2, QUIRKS_REFERENCE = 1, 2, 3      # include/ocean_hip.h OCEAN_QUIRK_*
CTX_FUSED_ONLY, CTX_TILE_RANK, CTX_TILE_BANDS = 1, 2, 4   # include/ocean_hip.h OCEAN_CTX_*
INTER_F32, INTER_BFP16 = 0, 1                         # include/ocean_hip.h OCEAN_INTER_*
PACK_RGBA32F, PACK_RGB32F, PACK_HEIGHT32F = 0, 1, 2   # include/ocean_hip.h OCEAN_PACK_*
PACK_BYTES_PER_TEXEL = {PACK_RGBA32F: 16, PACK_RGB32F: 12, PACK_HEIGHT32F: 4}
from .ocean import DOMAIN_SIZE, Correction, CorrectionLocals, Propagation, PropagateLocals


class OceanDevice:
    """One GPU + the buffers of the path (initial_spec, omega, dx/dy/dz_spec, displacement map:
    src/render.rs:607-670, 820-869)."""

    def __init__(self, resolution: int, device_ordinal: int = 0, flags: int = 0, tiles: int = 1, tiles_context: bool = False):
        """flags: 0 = both paths' buffers; CTX_FUSED_ONLY = the fused frame's only (40 instead of 100 / 76 B/texel; the staged
        dispatches then raise OCEAN_E_STATE); CTX_TILE_RANK = one rank of a sharded tile (static inputs only, 12 B/texel).
        tiles > 1 (or tiles_context: also for ONE tile) = ocean_context_create_tiles."""
        lib = load_library()
        ctx = ctypes.c_void_p()
        if tiles > 1 or tiles_context:   # K independent tiles' static inputs in one fused-only context: one frame of each per launch pair (N <= 1024)
            st = lib.ocean_context_create_tiles(int(device_ordinal), int(resolution), int(tiles), ctypes.byref(ctx))
        else:
            st = lib.ocean_context_create_ex(int(device_ordinal), int(resolution), int(flags), ctypes.byref(ctx))
        if st != OCEAN_OK:
            raise OceanError(st, (lib.ocean_last_error(None) or b"").decode())
        self._ctx = ctx
        self.tiles = int(tiles)
        self.resolution = int(resolution)
        self.device_ordinal = int(device_ordinal)

    @classmethod
    def for_tile_rank(cls, resolution: int, rank: int, world: int, device_ordinal: int = 0):
        """One rank of a sharded tile with only the input lines its pass 1 reads backed by memory (ocean_context_create_tile_rank:
        12 / world instead of 12 B/texel; fp32 spectrum; ocean_tile_pass1/2 for THIS rank and world only)."""
        lib = load_library()
        ctx = ctypes.c_void_p()
        st = lib.ocean_context_create_tile_rank(int(device_ordinal), int(resolution), int(rank), int(world), ctypes.byref(ctx))
        if st != OCEAN_OK:
            raise OceanError(st, (lib.ocean_last_error(None) or b"").decode())
        self = cls.__new__(cls)
        self._ctx, self.tiles, self.resolution, self.device_ordinal = ctx, 1, int(resolution), int(device_ordinal)
        return self

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, status: int):
        if status != OCEAN_OK:
            raise OceanError(status, (load_library().ocean_last_error(self._ctx) or b"").decode())

    @property
    def alive(self) -> bool:
        return bool(self._ctx)

    def destroy(self):
        """Frees the device buffers.  Stage objects created from this device refuse further use (their
        `_handle()` checks `alive`; the C ABI additionally rejects handles of a destroyed context)."""
        if self._ctx:
            load_library().ocean_context_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    @property
    def stream(self):
        return load_library().ocean_stream(self._ctx)

    def sync(self):
        self._check(load_library().ocean_sync(self._ctx))

    # -- upload (src/render.rs:742-924) ----------------------------------------------------------
    def upload_spectrum(self, h0: np.ndarray, omega: np.ndarray, spectrum_fp16: bool = False, tile: int = 0):
        """spectrum_fp16=True: BASELINE config 5 -- h0 kept in HBM as scaled fp16 pairs for the fused
        path (fp32 arithmetic); read_spectrum() then returns the dequantised values in use.
        tile: which tile of a context of several (OceanDevice(.., tiles=K)) these inputs belong to."""
        n = self.resolution
        h0 = np.ascontiguousarray(h0, dtype=np.complex64)
        omega = np.ascontiguousarray(omega, dtype=np.float32)
        if h0.shape != (n, n) or omega.shape != (n, n):
            raise OceanError(-1, f"expected ({n},{n}) arrays, got {h0.shape} and {omega.shape}")
        if tile or self.tiles > 1:
            if spectrum_fp16:
                raise OceanError(-1, "a context of several tiles stores fp32 spectra")
            self._check(load_library().ocean_upload_spectrum_tile(self._ctx, int(tile), h0.ctypes.data, omega.ctypes.data))
            return
        fn = load_library().ocean_upload_spectrum_f16 if spectrum_fp16 else load_library().ocean_upload_spectrum
        self._check(fn(self._ctx, h0.ctypes.data, omega.ctypes.data))

    def upload_spectrum_device(self, h0_ptr: int, omega_ptr: int, tile: int = 0, stream=None):
        """The upload's device-side half alone (copy_buffer, src/render.rs:896-915): the spectrum is read from memory the GPU can
        read -- `h0_ptr` / `omega_ptr` are addresses (e.g. `tensor.data_ptr()`) of c32[N*N] / f32[N*N] in device, managed or
        registered host memory.  Asynchronous on `stream` (None = the context's); fp32 storage."""
        self._check(load_library().ocean_upload_spectrum_device(self._ctx, int(tile), ctypes.c_void_p(int(h0_ptr)),
                                                                ctypes.c_void_p(int(omega_ptr)), stream))

    def read_spectrum(self) -> np.ndarray:
        n = self.resolution
        out = np.empty((n, n), dtype=np.complex64)
        self._check(load_library().ocean_read_spectrum(self._ctx, out.ctypes.data))
        return out

    @property
    def spectrum_scale_log2(self) -> int:
        return int(load_library().ocean_spectrum_scale_log2(self._ctx))

    # -- quirk switches (include/ocean_hip.h OCEAN_QUIRK_*) ---------------------------
    @property
    def quirks(self) -> int:
        return int(load_library().ocean_quirks(self._ctx))

    def set_quirks(self, quirks: int):
        """QUIRKS_REFERENCE (default) = the shipped shaders; clear QUIRK_Q1 for a signed wave index,
        QUIRK_Q2 for the conjugated (N+1-g) % N partner.  Non-reference settings run the staged kernels."""
        self._check(load_library().ocean_set_quirks(self._ctx, int(quirks)))

    # -- precision of the intermediate (include/ocean_hip.h OCEAN_INTER_*) ------------------------------
    def set_intermediate(self, mode: int):
        """INTER_F32 (default) or INTER_BFP16: int16 intermediate with block scales, opt-in, N = 8192 (config 5)."""
        self._check(load_library().ocean_set_intermediate(self._ctx, int(mode)))

    @property
    def intermediate(self) -> int:
        return int(load_library().ocean_intermediate(self._ctx))

    # -- fused frame ---------------------------------------------------------------------------------
    def frame(self, time: float, domain_size: float = DOMAIN_SIZE, stream=None):
        loc = PropagateLocals(time, self.resolution, domain_size)._c()
        self._check(load_library().ocean_frame_ex(self._ctx, ctypes.byref(loc), stream))

    def frame_batch(self, t0: float, dt: float, count: int, out_ptr=None, out_stride_bytes: int = 0, stream=None):
        """`count` time steps t0 + i dt of this tile, each into its own map; at N <= 1024 ONE launch pair (ocean_frame_batch).
        out_ptr None: library-owned maps (read_batch_displacement)."""
        self._check(load_library().ocean_frame_batch(self._ctx, float(t0), float(dt), int(count), out_ptr, int(out_stride_bytes), stream))

    def frame_tiles(self, time: float, out_ptr=None, out_stride_bytes: int = 0, stream=None):
        """One frame of EVERY tile of this context at `time` in one launch pair (ocean_frame_tiles); maps as frame_batch."""
        self._check(load_library().ocean_frame_tiles(self._ctx, float(time), out_ptr, int(out_stride_bytes), stream))

    def read_batch_displacement(self, index: int) -> np.ndarray:
        n = self.resolution
        out = np.empty((n, n, 4), dtype=np.float32)
        self._check(load_library().ocean_read_batch_displacement(self._ctx, int(index), out.ctypes.data))
        return out

    def read_batch_normals(self, index: int) -> np.ndarray:
        """The normal field of frame / tile `index` of the last batch (set_frame_normals on; N <= 1024)."""
        n = self.resolution
        out = np.empty((n, n, 4), dtype=np.float32)
        self._check(load_library().ocean_read_batch_normals(self._ctx, int(index), out.ctypes.data))
        return out

    def time_frame_batch(self, launches: int, count: int, t0: float = 0.0, dt: float = 1.0 / 60.0) -> float:
        """Milliseconds of `launches` back-to-back ocean_frame_batch calls of `count` frames each (ocean_time_frame_batch)."""
        ms = ctypes.c_float()
        self._check(load_library().ocean_time_frame_batch(self._ctx, int(launches), int(count), float(t0), float(dt), ctypes.byref(ms)))
        return float(ms.value)

    # -- normal field (shader/ocean.frag:50-66) ----------------------------------------
    def normals(self, source_channel: int = 0, stream=None) -> np.ndarray:
        """Finite-difference normals of the current displacement map; channel 0 = disp_x as the
        reference does (quirk Q5), 1 = height.  -> float32 [N, N, 4] = (n.x, n.y, n.z, 0)."""
        lib = load_library()
        if stream is not None:    # checked BEFORE anything is launched: the readback synchronises the context stream only
            raise OceanError(-1, "normals(): pass stream=None (the readback synchronises the context stream)")
        self._check(lib.ocean_normals(self._ctx, int(source_channel), None))
        n = self.resolution
        out = np.empty((n, n, 4), dtype=np.float32)
        self._check(lib.ocean_read_normals(self._ctx, out.ctypes.data))
        return out

    def set_frame_normals(self, source_channel):
        """The frame with its normal field as one workload (BASELINE config 3): channel 0..2 on, None / -1 off.  While on,
        every frame is followed by the normal-field kernel, fed by a dense plane of the source channel that pass 2 stores
        (ocean_set_frame_normals)."""
        self._check(load_library().ocean_set_frame_normals(self._ctx, -1 if source_channel is None else int(source_channel)))

    @property
    def frame_normals(self) -> int:
        return int(load_library().ocean_frame_normals(self._ctx))

    def read_normals(self) -> np.ndarray:
        """The normal field the last frame (set_frame_normals) or normals() call left in HBM -> float32 [N, N, 4]."""
        n = self.resolution
        out = np.empty((n, n, 4), dtype=np.float32)
        self._check(load_library().ocean_read_normals(self._ctx, out.ctypes.data))
        return out

    # -- vertex-stage positions (shader/ocean.vert:21-25) ----------------------------------
    def positions(self, verts: int = 128, offset=(0.0, 0.0), stream=None) -> np.ndarray:
        """World positions of a verts x verts patch (src/render.rs:494-508) displaced by the current map,
        sampled bilinearly with wrap as the reference's sampler does.  -> float32 [verts, verts, 4]."""
        lib = load_library()
        if stream is not None:    # same rule as normals(): the readback below waits for the context stream only
            raise OceanError(-1, "positions(): pass stream=None (the readback synchronises the context stream)")
        self._check(lib.ocean_positions(self._ctx, int(verts), float(offset[0]), float(offset[1]), None))
        out = np.empty((verts, verts, 4), np.float32)
        self._check(lib.ocean_read_positions(self._ctx, out.ctypes.data))
        return out

    def read_displacement(self) -> np.ndarray:
        n = self.resolution
        out = np.empty((n, n, 4), dtype=np.float32)
        self._check(load_library().ocean_read_displacement(self._ctx, out.ctypes.data))
        return out

    def read_field(self, field: int) -> np.ndarray:
        n = self.resolution
        out = np.empty((n, n), dtype=np.complex64)
        self._check(load_library().ocean_read_field(self._ctx, int(field), out.ctypes.data))
        return out

    def write_field(self, field: int, data: np.ndarray):
        n = self.resolution
        data = np.ascontiguousarray(data, dtype=np.complex64)
        if data.shape != (n, n):
            raise OceanError(-1, f"expected ({n},{n}) array, got {data.shape}")
        self._check(load_library().ocean_write_field(self._ctx, int(field), data.ctypes.data))

    def checksum(self, stream=None) -> int:
        """Order-independent 64-bit checksum of the current displacement map, computed on the device
        (ocean_checksum_displacement): equal maps <=> equal sums.  Synchronous."""
        out = ctypes.c_uint64()
        self._check(load_library().ocean_checksum_displacement(self._ctx, stream, ctypes.byref(out)))
        return int(out.value)

    def packed_bytes(self, fmt: int) -> int:
        v = int(load_library().ocean_packed_bytes(self._ctx, int(fmt)))
        if v < 0:
            raise OceanError(v, "unknown pack format")
        return v

    def pack_displacement(self, fmt: int, device_ptr, stream=None):
        """Packed copy of the current map into caller device memory (PACK_RGB32F: 12 B/texel, PACK_HEIGHT32F: 4):
        the payload of the final gather of a multi-GPU run."""
        self._check(load_library().ocean_pack_displacement(self._ctx, int(fmt), device_ptr, stream))

    def displacement_device_ptr(self) -> int:
        return int(load_library().ocean_displacement_device_ptr(self._ctx) or 0)

    def bind_displacement(self, device_ptr):
        self._check(load_library().ocean_bind_displacement(self._ctx, device_ptr))

    def bind_displacement_fd(self, fd: int, allocation_bytes: int, offset_bytes: int = 0):
        """Frames into memory another API exported as a file descriptor (a Vulkan VkDeviceMemory via VK_KHR_external_memory_fd, a
        HIP allocation via hipMemExportToShareableHandle); the descriptor is consumed (ocean_bind_displacement_fd)."""
        self._check(load_library().ocean_bind_displacement_fd(self._ctx, int(fd), int(allocation_bytes), int(offset_bytes)))

    # -- measurement ------------------------------------------------------------------------------------
    def time_frames(self, frames: int, t0: float = 0.0, dt: float = 1.0 / 60.0) -> float:
        """Total milliseconds (HIP events on the context stream) of `frames` fused frames."""
        ms = ctypes.c_float()
        self._check(load_library().ocean_time_frames(self._ctx, int(frames), float(t0), float(dt), ctypes.byref(ms)))
        return float(ms.value)

    def time_frame_batches(self, batches: int, frames_per_batch: int = 10, t0: float = 0.0, dt: float = 1.0 / 60.0):
        """Milliseconds of each of `batches` consecutive batches of fused frames (one stream event between batches, one sync
        at the end): the frame-time distribution of an undisturbed loop (ocean_time_frame_batches)."""
        ms = (ctypes.c_float * int(batches))()
        self._check(load_library().ocean_time_frame_batches(self._ctx, int(batches), int(frames_per_batch), float(t0), float(dt), ms))
        return [float(v) for v in ms]

    def frame_times(self, frames: int, t0: float = 0.0, dt: float = 1.0 / 60.0):
        """Per-frame (pass1_ms, pass2_ms, period_ms) lists of a back-to-back loop of `frames` fused frames, from events bound
        to the dispatches (ocean_frame_times)."""
        arr = [(ctypes.c_float * int(frames))() for _ in range(3)]
        self._check(load_library().ocean_frame_times(self._ctx, int(frames), float(t0), float(dt), *arr))
        return tuple([float(v) for v in a] for a in arr)

    def frame_times_ex(self, frames: int, t0: float = 0.0, dt: float = 1.0 / 60.0):
        """(pass1_ms, pass2_ms, normals_ms, period_ms) per frame; normals_ms is None unless the frame carries the normal
        field (ocean_frame_times_ex)."""
        nrm = self.frame_normals >= 0
        arr = [(ctypes.c_float * int(frames))() for _ in range(4)]
        args = [arr[0], arr[1], arr[2] if nrm else None, arr[3]]
        self._check(load_library().ocean_frame_times_ex(self._ctx, int(frames), float(t0), float(dt), *args))
        return tuple(None if a is None else [float(v) for v in a] for a in args)

    def _profile(self, fn, time):
        cap = 16
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int32()
        self._check(fn(self._ctx, float(time), cap, names, ms, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def profile_frame(self, time: float = 0.0):
        return self._profile(load_library().ocean_profile_frame, time)

    def profile_staged(self, time: float = 0.0):
        return self._profile(load_library().ocean_profile_staged, time)


class OceanRenderer:
    """`Renderer::new` + `Renderer::render` restricted to the compute path.

    `render(time)` records exactly the reference's sequence (src/render.rs:1122-1310):
    propagate -> row pass x {dx,dy,dz} -> col pass x {dx,dy,dz} -> correction, stream order
    standing in for the four pipeline barriers.  `render_fused(time)` is the 2-launch path.
    """

    def __init__(self, resolution: int = 512, device_ordinal: int = 0, domain_size: float = DOMAIN_SIZE):
        self.device = OceanDevice(resolution, device_ordinal)
        self.fft = Fft.init(self.device)                  # src/render.rs:223
        self.propagation = Propagation.init(self.device)  # src/render.rs:224
        self.correction = Correction.init(self.device)    # src/render.rs:225
        self.domain_size = float(domain_size)

    def upload(self, h0, omega, spectrum_fp16: bool = False):
        self.device.upload_spectrum(h0, omega, spectrum_fp16)

    def render(self, time: float, stream=None):
        n = self.device.resolution
        self.propagation.dispatch(PropagateLocals(time, n, self.domain_size), stream)   # :1101-1130
        self.fft.row_pass(FIELD_ALL, stream)                                            # :1158-1179
        self.fft.col_pass(FIELD_ALL, stream)                                            # :1210-1231
        self.correction.dispatch(CorrectionLocals(n), stream)                           # :1280-1287

    def render_fused(self, time: float, stream=None):
        self.device.frame(time, self.domain_size, stream)

    def displacement(self) -> np.ndarray:
        return self.device.read_displacement()

    def dispose(self):
        """src/render.rs:1383-1438."""
        self.fft.destroy()
        self.propagation.destroy()
        self.correction.destroy()
        self.device.destroy()
