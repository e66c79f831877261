"""ctypes driver of tests/hipemu/libocean_emu.so: the product kernels executed on the CPU."""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# OCEAN_EMU_FLAGS="-DOCEAN_DMA_MIN_N=256" builds (and loads) the emulation with the LDS-DMA loader at every size
_FLAGS = os.environ.get("OCEAN_EMU_FLAGS", "").split()
_TAG = ("_" + "_".join(f.replace("-D", "").replace("=", "") for f in _FLAGS)) if _FLAGS else ""
_SO = os.path.join(_ROOT, "tests", "hipemu", f"libocean_emu{_TAG}.so")
_SRC = [os.path.join(_ROOT, "tests", "hipemu", "emu_kernels.cpp"),
        os.path.join(_ROOT, "tests", "hipemu", "hip", "hip_runtime.h"),
        os.path.join(_ROOT, "gfx_ocean_amd", "csrc", "ocean_kernels.hpp"),
        os.path.join(_ROOT, "gfx_ocean_amd", "csrc", "ocean_staged_kernels.hpp"),
        os.path.join(_ROOT, "gfx_ocean_amd", "csrc", "fft_core.hpp"),
        os.path.join(_ROOT, "tests", "hipemu", "ocean_device_intrinsics.hpp")]
_LIB = None


def build(force=False):
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRC)
    if force or stale:
        subprocess.check_call([
            "g++", "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC",
            "-I", os.path.join(_ROOT, "tests", "hipemu"), "-I", os.path.join(_ROOT, "gfx_ocean_amd", "csrc"), *_FLAGS,
            _SRC[0], "-o", _SO])


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(_SO)
        _LIB.emu_frame_half.argtypes = ([ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 5 +
                                        [ctypes.c_size_t] * 3 + [ctypes.c_int] + [ctypes.c_float] * 2 + [ctypes.c_void_p])
        _LIB.emu_fft_lines.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _LIB.emu_propagate.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_float] * 2 + [ctypes.c_uint32]
        _LIB.emu_correct.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def twiddles(n):
    i = np.arange(n, dtype=np.float64)
    return np.exp(2j * np.pi * i / n).astype(np.complex64)


def fft_lines(field, col):
    f = np.ascontiguousarray(field, np.complex64).copy()
    n = f.shape[0]
    tw = twiddles(n)
    assert lib().emu_fft_lines(n, int(col), _p(f), _p(tw)) == 0
    return f


def propagate(h0, omega, time, L=1000.0, quirks=3):
    n = h0.shape[0]
    h0 = np.ascontiguousarray(h0, np.complex64)
    omega = np.ascontiguousarray(omega, np.float32)
    outs = [np.empty((n, n), np.complex64) for _ in range(3)]
    assert lib().emu_propagate(n, _p(h0), _p(omega), *map(_p, outs), time, L, quirks) == 0
    return tuple(outs)


def correct(h, dx, dz):
    n = h.shape[0]
    a = [np.ascontiguousarray(x, np.complex64) for x in (h, dx, dz)]
    out = np.empty((n, n, 4), np.float32)
    assert lib().emu_correct(n, *map(_p, a), _p(out)) == 0
    return out


def chunk():
    """(columns, rows) of a chunk in the loaded build (4 x 4 unless an A/B variant was requested)."""
    return lib().emu_chunk_w(), lib().emu_chunk_r()


def _layout(columns, n, P, layout, pad):
    """(sx, sy, fs) in elements -- mirrors ocean_context_create: chunks are CHUNK_W columns x CHUNK_R rows
    (16 elements) whatever the number of lines P a pass-1 workgroup owns."""
    cw, cr = chunk()
    gx, gy = columns // cw, n // cr
    if layout == "p1":
        sy, sx = 16, gy * 16 + pad
        return sx, sy, sx * gx
    sx, sy = 16, gx * 16 + pad
    return sx, sy, sy * gy


def inter_layout(n, P, layout="p2", pad=32):
    return _layout(n, n, P, layout, pad)


def unpack_inter(inter, n, P, lay, f, columns=None, cmajor=False):
    """Intermediate field f -> natural [y, x] array (columns < n for the half-spectrum path).  cmajor: the chunks of the
    split geometry (k_half_pass1_split -> k_half_pass2_real) are column-major inside."""
    sx, sy, fs = lay[:3]
    columns = columns or n
    cw, cr = chunk()
    X, Y, r, c = np.meshgrid(np.arange(columns // cw), np.arange(n // cr), np.arange(cr), np.arange(cw), indexing="ij")
    if len(lay) == 4:                                   # block layout of the half-spectrum path
        b = lay[3]
        idx = f * fs + (Y >> b) * sy + X * sx + (Y & ((1 << b) - 1)) * (cw * cr) + ((c * cr + r) if cmajor else (r * cw + c))
    else:
        idx = f * fs + X * sx + Y * sy + r * cw + c
    out = np.empty((n, columns), np.complex64)
    out[(Y * cr + r).ravel(), (X * cw + c).ravel()] = inter[idx.ravel()]
    return out


def half_layout(n, P, layout="p2", pad=32, bshift=None):
    """(sx, sy, fs, bshift) of the half-spectrum intermediate (N/2 columns) -- mirrors ocean_context_create:
    chunk (X, Y) at (Y / B) * sy + X * sx + (Y % B) * 16, B = 2^bshift.  layout "p2" = B 1, "p1" = B N/4 (all chunk
    rows), "default" = what the build ships for this N (Geo::inter_bshift); an explicit bshift overrides."""
    cw, cr = chunk()
    gx, gy = (n // 2) // cw, n // cr
    if bshift is None:
        if layout == "default":
            bshift = lib().emu_inter_bshift(n)
        else:
            bshift = 0 if layout == "p2" else gy.bit_length() - 1
    bshift = min(bshift, gy.bit_length() - 1)
    B = 1 << bshift
    sx = B * 16
    sy = gx * sx + pad
    return sx, sy, sy * (gy // B), bshift


def uses_dma(n, P=None, split=False):
    """True when fused pass 1 of this geometry streams its inputs through the LDS-DMA ring (half_load_AB_dma)."""
    return bool(lib().emu_uses_dma(n, 22 if split else int(P or 0)))


def quantize_f16(h0):
    """Host side of the config-5 upload (mirrors ocean_api.hip upload_common): -> (packed uint32 [N,N]
    = fp16 re | fp16 im << 16 of h0 * 2^s, dequantised complex64 [N,N], s)."""
    h0 = np.ascontiguousarray(h0, np.complex64)
    mx = float(np.abs(h0.view(np.float32)).max())
    s = 14 - int(np.floor(np.log2(mx))) if mx > 0 else 0
    scaled = (h0.view(np.float32).astype(np.float32) * np.float32(2.0 ** s)).astype(np.float16)
    deq = (scaled.astype(np.float32) * np.float32(2.0 ** -s)).view(np.complex64).reshape(h0.shape)
    bits = scaled.view(np.uint16).reshape(h0.shape + (2,)).astype(np.uint32)
    return (bits[..., 0] | (bits[..., 1] << 16)).astype(np.uint32), deq, s


def frame_half(h0, omega, time, L=1000.0, layout="p2", return_inter=False, spectrum_fp16=False, P=None, split=False, bshift=None,
               inter16=False, plane_channel=None, batch=None):
    """split=True: every line as two interleaved N/2 transforms (the N = 8192 geometry, P = 2; with P=1 the N = 16384 geometry:
    one column per pass-1 workgroup); inter16=True (split, P = 2 only): the 16-bit block-floating intermediate (OCEAN_INTER_BFP16)."""
    # h0 / omega [K, n, n] with batch=(K, 0.0): K TILES per launch pair (ocean_frame_tiles) instead of K time steps of one
    tiles = h0.ndim == 3
    n = h0.shape[-1]
    P = (1 if P == 1 else 2) if split else (P or lib().emu_frame_p(n))
    descale = 1.0
    if spectrum_fp16:
        packed, _, s = quantize_f16(h0)
        h0T = np.ascontiguousarray(packed.T, np.uint32)
        descale = 2.0 ** -s
    elif tiles:
        h0T = np.ascontiguousarray(np.swapaxes(h0, 1, 2), np.complex64)
    else:
        h0T = np.ascontiguousarray(h0.T, np.complex64)
    omT = np.ascontiguousarray(np.swapaxes(omega, -1, -2), np.float32)
    sx, sy, fs, bshift = half_layout(n, P, layout, bshift=bshift)
    # batch = (count, dt): `count` time steps time + i dt in ONE launch pair (ocean_frame_batch, N <= 1024) -> out [count, n, n, 4]
    count, dt = batch if batch else (1, 0.0)
    inter = np.full(count * 3 * fs, np.nan + 1j * np.nan, np.complex64)
    nyq = np.full(count * 6 * n, np.nan, np.float32)   # scratch: the Nyquist column's three spectra (per frame)
    out = np.full((count, n, n, 4) if batch else (n, n, 4), np.nan, np.float32)
    lib().emu_set_batch.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_uint, ctypes.c_size_t]
    lib().emu_set_batch(count, dt, 3 * fs, n * n)
    lib().emu_set_batch_tiles.argtypes = [ctypes.c_size_t, ctypes.c_uint]
    lib().emu_set_batch_tiles(n * n * 8 if tiles else 0, n * n if tiles else 0)
    tw = twiddles(n)
    assert split or not inter16
    scales = np.full(3 * (n // 64) * (n // 4), np.nan, np.float32) if inter16 else None
    psel = (23 if inter16 else (21 if P == 1 else 22)) if split else int(P)
    # pass 2 runs as its PLANE instance (the frame with the normal field): the source channel as a dense fp32 plane
    plane = np.full((count, n, n) if batch else (n, n), np.nan, np.float32)       # (a batch: one plane per frame)
    lib().emu_set_plane.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib().emu_set_plane(_p(plane), int(plane_channel or 0))
    assert lib().emu_frame_half(n, psel, _p(h0T), int(spectrum_fp16), descale, _p(omT), _p(inter),
                                _p(nyq), _p(out), _p(tw), sx, sy, fs, bshift, time, L, _p(scales) if inter16 else None) == 0
    lib().emu_set_plane(None, 0)
    lib().emu_set_batch(1, 0.0, 0, 0)
    if plane_channel is not None:
        return out, plane
    if return_inter:
        return out, inter, nyq, (P, (sx, sy, fs, bshift))
    return out


def cols4(field, s):
    """The two-step staged column pass (k_cols4_a in place, k_cols4_b out of place): column transforms of `field` [NF, NF]."""
    f = np.ascontiguousarray(field, np.complex64).copy()
    nf = f.shape[0]
    dst = np.full_like(f, np.nan + 1j * np.nan)
    tw = twiddles(nf)
    lib().emu_cols4.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib().emu_cols4(nf, int(s), _p(f), _p(dst), _p(tw)) == 0
    return dst


def cols4_correct(h, dx, dz, s):
    """Step A of the three fields, then k_cols4_b_correct (step B of the three + correction.comp): the RGBA map [NF, NF, 4]."""
    f = [np.ascontiguousarray(v, np.complex64).copy() for v in (h, dx, dz)]
    nf = f[0].shape[0]
    out = np.full((nf, nf, 4), np.nan, np.float32)
    tw = twiddles(nf)
    lib().emu_cols4_correct.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    assert lib().emu_cols4_correct(nf, int(s), _p(f[0]), _p(f[1]), _p(f[2]), _p(out), _p(tw)) == 0
    return out


def normals_plane(plane, bands=False):
    """k_normals_plane (bands=True: k_normals_plane_bands, the N >= 8192 kernel; n >= 1024 here): the normal field from the dense
    source-channel plane of the fused pass 2."""
    plane = np.ascontiguousarray(plane, np.float32)
    n = plane.shape[0]
    out = np.full((n, n, 4), np.nan, np.float32)
    fn = lib().emu_normals_plane_bands if bands else lib().emu_normals_plane
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(n, _p(plane), _p(out)) == 0
    return out


def normals(rgba, channel=0):
    rgba = np.ascontiguousarray(rgba, np.float32)
    n = rgba.shape[0]
    out = np.empty((n, n, 4), np.float32)
    lib().emu_normals.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert lib().emu_normals(n, _p(rgba), _p(out), int(channel)) == 0
    return out


def positions(rgba, verts=128, offset=(0.0, 0.0)):
    rgba = np.ascontiguousarray(rgba, np.float32)
    n = rgba.shape[0]
    out = np.empty((verts, verts, 4), np.float32)
    lib().emu_positions.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    assert lib().emu_positions(n, _p(rgba), _p(out), int(verts), float(offset[0]), float(offset[1])) == 0
    return out


def staged_chunked(fields, do_cols=True):
    """The staged row pass (natural -> chunked), column pass (in place on the chunks) and both consumers
    (k_unchunk, k_correct_chunked) for the three natural fields (height, disp_x, disp_z).
    -> (natural fields after the passes, RGBA from the chunked correction)."""
    n = fields[0].shape[0]
    sx, sy, fs = inter_layout(n, 4)
    tw = twiddles(n)
    L = lib()
    L.emu_stage.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t] * 3
    L.emu_unchunk.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_size_t] * 3
    L.emu_correct_chunked.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t] * 3
    chunked, natural = [], []
    for f in fields:
        nat = np.ascontiguousarray(f, np.complex64).copy()
        chk = np.full(fs, np.nan + 1j * np.nan, np.complex64)
        assert L.emu_stage(n, 0, _p(nat), _p(chk), _p(tw), sx, sy, fs) == 0
        if do_cols:
            assert L.emu_stage(n, 1, None, _p(chk), _p(tw), sx, sy, fs) == 0
        back = np.full((n, n), np.nan + 1j * np.nan, np.complex64)
        assert L.emu_unchunk(n, _p(chk), _p(back), sx, sy, fs) == 0
        chunked.append(chk)
        natural.append(back)
    out = np.full((n, n, 4), np.nan, np.float32)
    assert L.emu_correct_chunked(n, *map(_p, chunked), _p(out), sx, sy, fs) == 0
    return natural, out


class EmuShardBackend:
    """Backend of gfx_ocean_amd.sharded.ShardedTile that runs the shard kernels on the CPU (numpy buffers wrapped as
    torch CPU tensors for gloo).  Test infrastructure: the product backend is HipShardBackend."""

    def __init__(self, n, rank, world):
        import torch
        self.torch = torch
        self.n, self.rank, self.world, self.rows = n, rank, world, n // world
        self.tw = twiddles(n)
        self.fld = np.zeros((3, self.rows, n), np.complex64)
        L = lib()
        L.emu_shard_rows.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6 + [ctypes.c_float] * 2
        L.emu_shard_cols.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4

    def upload(self, h0_own, h0_partner, omega_own):
        self.h0_own, self.h0_partner, self.omega = h0_own, h0_partner, omega_own

    def alloc_exchange(self):
        return self.torch.zeros((self.world, 3, self.rows, self.rows, 2), dtype=self.torch.float32)

    def alloc_out(self):
        return self.torch.zeros((self.rows, self.n, 4), dtype=self.torch.float32)

    def rows_pass(self, time, domain_size, send):
        buf = send.numpy()
        assert lib().emu_shard_rows(self.n, self.rank, self.world, _p(self.h0_own), _p(self.h0_partner), _p(self.omega),
                                    _p(self.fld), _p(buf), _p(self.tw), float(time), float(domain_size)) == 0

    def cols_pass(self, recv, out):
        assert lib().emu_shard_cols(self.n, self.rank, self.world, _p(recv.numpy()), _p(self.fld), _p(out.numpy()), _p(self.tw)) == 0

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def to_numpy(self, out):
        return out.numpy().copy()

    def destroy(self):
        pass


class EmuTileBackend:
    """Backend of gfx_ocean_amd.sharded.FusedShardedTile that runs the fused kernels of ocean_tile_pass1 / ocean_tile_pass2
    on the CPU (numpy buffers wrapped as torch CPU tensors for gloo).  Test infrastructure only."""

    def __init__(self, n, rank, world, psel=None, parts=1):
        import torch                                               # psel 22 / 21: the split geometry (the product's at N = 8192 / 16384)
        self.torch = torch
        self.n, self.rank, self.world, self.rows, self.parts = n, rank, world, n // world, parts
        self.tw = twiddles(n)
        self.psel = int(psel if psel is not None else {512: 1, 4096: 4}.get(n, 2))      # Launch<N>::default_psel()
        self.nyq = np.full(6 * n, np.nan, np.float32)
        lib().emu_tile.argtypes = ([ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 5 +
                                   [ctypes.c_float] * 2)
        self.exchange_floats = 3 * (n // 2) * (n // world) * 2

    def upload(self, h0, omega):
        self.h0T = np.ascontiguousarray(h0.T, np.complex64)
        self.omT = np.ascontiguousarray(omega.T, np.float32)

    def alloc_exchange(self):
        return self.torch.full((self.parts, self.world, self.exchange_floats // self.world // self.parts), float("nan"),
                               dtype=self.torch.float32)

    def alloc_out(self):
        return self.torch.full((self.rows, self.n, 4), float("nan"), dtype=self.torch.float32)

    def pass1(self, time, domain_size, send_part, part=0):
        assert send_part.is_contiguous()
        assert lib().emu_tile(self.n, 1, self.psel, self.rank, self.world, int(part), self.parts, _p(self.h0T), 0, 1.0, _p(self.omT),
                              _p(send_part.numpy()), _p(self.nyq), None, _p(self.tw), float(time), float(domain_size)) == 0

    def pass2(self, recv, out):
        assert lib().emu_tile(self.n, 2, self.psel, self.rank, self.world, 0, self.parts, None, 0, 1.0, None, _p(recv.numpy()), None,
                              _p(out.numpy()), _p(self.tw), 0.0, 0.0) == 0

    def exchange(self, dist, recv_part, send_part):
        dist.all_to_all_single(recv_part, send_part)

    def join_exchanges(self):
        pass

    def synchronize(self):
        pass

    def to_numpy(self, out):
        return out.numpy().copy()
