"""CPU oracle for the gfx-ocean hot path (propagate -> row iFFT -> col iFFT -> correction).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``gfx_ocean_amd/``) never does and fails loudly when its HIP library is missing.

PARITY PINNING.  The reference has no tests, no golden outputs and no CPU path, and it cannot
be built or run in the build container (Rust + gfx-hal + Vulkan are absent), so parity is
UNPINNED BY THE REFERENCE'S OWN TESTS.  What pins this oracle instead:
  * the reference's two input files (tests/golden/{spectrum,omega}.bin, data);
  * oracle/spirv_interp.py, which executes the reference's *shipped SPIR-V binaries*
    (shader/spv/*.comp.spv, the code the reference actually runs) in the build container on those
    inputs; its outputs are committed as tests/golden/spirv_frame512_t*.npz and the
    ``*_literal`` functions below reproduce them BIT FOR BIT at every stage
    (tests/test_oracle.py::test_literal_oracle_reproduces_the_shipped_spirv_bit_for_bit);
  * two independent formulations that must agree to ~1e-6
    (``*_literal`` = fp32 restatement of the shaders instruction by instruction,
    ``frame_f64`` = closed-form fp64 with numpy's pocketfft);
  * the survey-time KAT table (tests/golden/kat_survey.json).
Driver-defined precision (sin, cos, length, division, FMA contraction) is fixed to "correctly
rounded fp32, no contraction"; the reference's own GPU result is implementation-defined at that level.

Every function cites the reference file:line it restates.  Array convention:
row-major, ``index = x + N*y`` with x = gl_GlobalInvocationID.x fastest
(shader/propagate.comp:43), so numpy arrays are indexed ``a[y, x]``.
"""
from __future__ import annotations

import struct
import numpy as np

# shader/propagate.comp:6, shader/fft_row.comp:5 -- `const float pi = 3.1415926;`
# (fp32 value 0x40490FDA, one ulp below the correctly rounded pi)
PI_F32 = np.float32(3.1415926)
assert PI_F32.view(np.uint32) == 0x40490FDA

DOMAIN_SIZE = np.float32(1000.0)  # src/render.rs:46
RESOLUTION = 512                  # src/render.rs:44


# --------------------------------------------------------------------------
# data/*.bin decode -- src/render.rs:769-771 (omega), :808-810 (spectrum)
# bincode 1.3.1 Vec<T>: u64-LE element count, then the LE payload.
# --------------------------------------------------------------------------
def read_bincode_f32(path: str, lanes: int) -> np.ndarray:
    with open(path, "rb") as f:
        raw = f.read()
    (count,) = struct.unpack_from("<Q", raw, 0)
    payload = np.frombuffer(raw, dtype="<f4", offset=8)
    if payload.size != count * lanes:
        raise ValueError(f"{path}: header says {count} x {lanes} f32, payload has {payload.size}")
    return payload.reshape(count, lanes) if lanes > 1 else payload.copy()


def load_reference_inputs(spectrum_path: str, omega_path: str):
    """-> (h0 complex64 [N,N], omega float32 [N,N]) for the shipped 512x512 data."""
    spec = read_bincode_f32(spectrum_path, 2)
    omega = read_bincode_f32(omega_path, 1)
    n = int(round(np.sqrt(omega.size)))
    assert n * n == omega.size == spec.shape[0]
    h0 = np.ascontiguousarray(spec, dtype=np.float32).view(np.complex64).reshape(n, n)
    return h0, omega.astype(np.float32).reshape(n, n)


def centre_crop(a: np.ndarray, n: int) -> np.ndarray:
    """the N=256 case is the centre crop [128:384]^2 of the 512 data."""
    big = a.shape[0]
    o = (big - n) // 2
    return np.ascontiguousarray(a[o:o + n, o:o + n])


# --------------------------------------------------------------------------
# helpers: fp32 transcendental semantics = correctly rounded result of the
# fp32 argument (SURVEY 7 "Transcendentals").
# --------------------------------------------------------------------------
def _cos32(x32):
    return np.cos(x32.astype(np.float64)).astype(np.float32)


def _sin32(x32):
    return np.sin(x32.astype(np.float64)).astype(np.float32)


def _cmul32(ar, ai, br, bi):
    # complex_mul, shader/propagate.comp:12-14: (a.x*b.x - a.y*b.y, a.y*b.x + a.x*b.y)
    return (ar * br - ai * bi).astype(np.float32), (ai * br + ar * bi).astype(np.float32)


def wave_vector_q1(n: int):
    """float(uint(2*g - N - 1)) -- shader/propagate.comp:45-46 (quirk Q1: the
    subtraction is evaluated in uint and wraps; ConvertUToF rounds to nearest even)."""
    g = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        xu = (np.uint32(2) * g - np.uint32(n) - np.uint32(1)).astype(np.uint32)
    return xu.astype(np.float32)  # numpy uint32->float32 is RNE, same as v_cvt_f32_u32


# --------------------------------------------------------------------------
# a1: shader/propagate.comp:42-72
# --------------------------------------------------------------------------
def propagate_literal(h0: np.ndarray, omega: np.ndarray, time, domain_size=DOMAIN_SIZE):
    """fp32 restatement.  Returns (height, disp_x, disp_z) complex64 [N,N]
    (= bindings 3,4,5 = dy_spec, dx_spec, dz_spec, src/render.rs:949-954)."""
    n = h0.shape[0]
    t = np.float32(time)
    L = np.float32(domain_size)
    xf = wave_vector_q1(n)
    kx = ((PI_F32 * xf) / L).astype(np.float32)[None, :].repeat(n, 0)   # :50-53, x from gid.x
    ky = ((PI_F32 * xf) / L).astype(np.float32)[:, None].repeat(n, 1)   # y from gid.y

    # index_neg = (N-gy-1)*N + N-gx-1 (:48) -> flip both axes, no conjugate (Q2)
    h0r = h0.real.astype(np.float32)
    h0i = h0.imag.astype(np.float32)
    hnr = h0r[::-1, ::-1]
    hni = h0i[::-1, ::-1]

    d = (omega.astype(np.float32) * t).astype(np.float32)               # :55
    c = _cos32(d)
    s = _sin32(d)
    pr, pi_ = _cmul32(h0r, h0i, c, s)                                   # :59-60 h0*disp_pos
    nr, ni = _cmul32(hnr, hni, c, (-s).astype(np.float32))              # :61 h0[neg]*disp_neg
    hr = (pr + nr).astype(np.float32)
    hi = (pi_ + ni).astype(np.float32)

    ln = np.sqrt((kx * kx + ky * ky).astype(np.float32)).astype(np.float32)  # length(k) :65
    ok = ln > np.float32(1.0e-10)
    safe = np.where(ok, ln, np.float32(1.0))
    knx = np.where(ok, (kx / safe).astype(np.float32), np.float32(0.0))
    kny = np.where(ok, (ky / safe).astype(np.float32), np.float32(0.0))

    zero = np.zeros_like(hr)
    dxr, dxi = _cmul32(zero, (-knx).astype(np.float32), hr, hi)         # :70
    dzr, dzi = _cmul32(zero, (-kny).astype(np.float32), hr, hi)         # :71
    mk = lambda r, i: (r + 1j * i).astype(np.complex64)
    return mk(hr, hi), mk(dxr, dxi), mk(dzr, dzi)


# --------------------------------------------------------------------------
# a2/a3/a4: shader/fft_row.comp:25-63, shader/fft_col.comp:44-63
# --------------------------------------------------------------------------
def _stockham_lines_literal(lines: np.ndarray) -> np.ndarray:
    """Radix-2 Stockham of every row of `lines` ([batch, N] complex64), literal:
    log2(N) stages, ping-pong buffers, theta = (pi*float(k))/float(bs) in fp32
    (fft_row.comp:32), w = (cos, sin), dest = 2j-k (:36).  N generic (Q4:
    512->N, 256->N/2, 9->log2 N)."""
    n = lines.shape[1]
    half = n // 2
    stages = n.bit_length() - 1
    assert 1 << stages == n
    sr = lines.real.astype(np.float32).copy()
    si = lines.imag.astype(np.float32).copy()
    j = np.arange(half, dtype=np.uint32)
    for i in range(stages):
        bs = np.uint32(1 << i)
        k = j & (bs - np.uint32(1))
        theta = ((PI_F32 * k.astype(np.float32)) / np.float32(bs)).astype(np.float32)
        c = _cos32(theta)[None, :]
        s = _sin32(theta)[None, :]
        in0r, in0i = sr[:, :half], si[:, :half]
        in1r, in1i = sr[:, half:], si[:, half:]
        tr, ti = _cmul32(in1r, in1i, c, s)
        dest = ((j << np.uint32(1)) - k).astype(np.int64)
        dr = np.empty_like(sr)
        di = np.empty_like(si)
        dr[:, dest] = (in0r + tr).astype(np.float32)
        di[:, dest] = (in0i + ti).astype(np.float32)
        dr[:, dest + int(bs)] = (in0r - tr).astype(np.float32)
        di[:, dest + int(bs)] = (in0i - ti).astype(np.float32)
        sr, si = dr, di
    return (sr + 1j * si).astype(np.complex64)


def fft_rows_literal(field: np.ndarray) -> np.ndarray:
    """fft_row.comp main: line y = field[y, :] (index = gid.x + N*gid.y)."""
    return _stockham_lines_literal(field)


def fft_cols_literal(field: np.ndarray) -> np.ndarray:
    """fft_col.comp main: line x = field[:, x] (index = gid.y + N*gid.x ... element m at x + N*m)."""
    return np.ascontiguousarray(_stockham_lines_literal(np.ascontiguousarray(field.T)).T)


# --------------------------------------------------------------------------
# a5: shader/correction.comp:24-35
# --------------------------------------------------------------------------
def correction_literal(height: np.ndarray, disp_x: np.ndarray, disp_z: np.ndarray) -> np.ndarray:
    """-> float32 [N,N,4] = (dx, h, dz, 0) * sign, sign = -1 where (x+y) even."""
    n = height.shape[0]
    g = np.arange(n)
    sign = np.where(((g[None, :] + g[:, None]) % 2) == 0, np.float32(-1.0), np.float32(1.0))
    out = np.zeros((n, n, 4), dtype=np.float32)
    out[..., 0] = disp_x.real.astype(np.float32) * sign
    out[..., 1] = height.real.astype(np.float32) * sign
    out[..., 2] = disp_z.real.astype(np.float32) * sign
    return out


# --------------------------------------------------------------------------
# a8: the frame recorder order, src/render.rs:1122-1310
# --------------------------------------------------------------------------
def frame_literal(h0, omega, time, domain_size=DOMAIN_SIZE, return_stages=False):
    h, dx, dz = propagate_literal(h0, omega, time, domain_size)
    stages = {"propagate": (h, dx, dz)}
    h, dx, dz = (fft_rows_literal(f) for f in (h, dx, dz))
    stages["rows"] = (h, dx, dz)
    h, dx, dz = (fft_cols_literal(f) for f in (h, dx, dz))
    stages["cols"] = (h, dx, dz)
    out = correction_literal(h, dx, dz)
    return (out, stages) if return_stages else out


# --------------------------------------------------------------------------
# Independent fp64 formulation (the tolerance anchor).
# --------------------------------------------------------------------------
QUIRK_Q1, QUIRK_Q2, QUIRKS_REFERENCE = 1, 2, 3


def propagate_f64(h0, omega, time, domain_size=DOMAIN_SIZE, quirks=QUIRKS_REFERENCE):
    """quirks: bit 0 = Q1 (uint wave index, shader/propagate.comp:45-46), bit 1 = Q2 (partner N-1-g, not
    conjugated, :48,59-62); both set = the reference.  Off: signed wave index; partner (N+1-g) % N, conjugated
    (include/ocean_hip.h OCEAN_QUIRK_*)."""
    n = h0.shape[0]
    if quirks & QUIRK_Q1:
        xf = wave_vector_q1(n).astype(np.float64)       # Q1 incl. the uint->f32 rounding
    else:
        xf = 2.0 * np.arange(n, dtype=np.float64) - n - 1
    kx = np.broadcast_to(xf[None, :], (n, n))
    ky = np.broadcast_to(xf[:, None], (n, n))
    d = (omega.astype(np.float32) * np.float32(time)).astype(np.float32).astype(np.float64)
    h0c = h0.astype(np.complex128)
    if quirks & QUIRK_Q2:
        partner = h0c[::-1, ::-1]
    else:
        p = (n + 1 - np.arange(n)) % n
        partner = np.conj(h0c[np.ix_(p, p)])
    h = h0c * np.exp(1j * d) + partner * np.exp(-1j * d)
    ln = np.hypot(kx, ky)                               # pi/L cancels in k/|k|
    return h, (-1j * kx / ln) * h, (-1j * ky / ln) * h


def frame_f64(h0, omega, time, domain_size=DOMAIN_SIZE, quirks=QUIRKS_REFERENCE):
    """N^2 * ifft2 (= unnormalised e^{+i} DFT on both axes), sign, real, pack.  float64 [N,N,4]."""
    n = h0.shape[0]
    h, dx, dz = propagate_f64(h0, omega, time, domain_size, quirks)
    g = np.arange(n)
    sign = np.where(((g[None, :] + g[:, None]) % 2) == 0, -1.0, 1.0)
    out = np.zeros((n, n, 4), dtype=np.float64)
    for c, f in ((0, dx), (1, h), (2, dz)):
        out[..., c] = (np.fft.ifft2(f) * (n * n)).real * sign
    return out


def ifft_lines_f64(lines):
    """Unnormalised inverse DFT of each row: X[n] = sum_m x[m] e^{+2 pi i m n / N}."""
    return np.fft.ifft(lines.astype(np.complex128), axis=-1) * lines.shape[-1]


# --------------------------------------------------------------------------
# 8f #1: the "normal field" -- shader/ocean.frag:50-66, restated at texel centres with the
# sampler's Tile wrap (src/render.rs:398).  Channel R (= disp_x) is differentiated, not height
# (quirk Q5); dim literal 512 -> N; height_scale = 180 (:19).
# --------------------------------------------------------------------------
HEIGHT_SCALE = np.float32(180.0)


def normals_literal(rgba: np.ndarray, channel: int = 0) -> np.ndarray:
    """rgba float32 [N,N,4] -> float32 [N,N,4] = (n.x, n.y, n.z, 0).  channel 0 = reference (Q5)."""
    f = np.float32
    r = rgba[..., channel].astype(f)
    n = r.shape[0]
    x0, x1 = np.roll(r, 1, axis=1), np.roll(r, -1, axis=1)      # textureOffset(-1,0) / (+1,0), wrap
    z0, z1 = np.roll(r, 1, axis=0), np.roll(r, -1, axis=0)
    d = f(2.0) / f(n)                                             # diff = 2 / dim (:52)

    def normalize(v):
        ln = np.sqrt((v[0] * v[0] + v[1] * v[1] + v[2] * v[2]).astype(f)).astype(f)
        return [(c / ln).astype(f) for c in v]

    zero = np.zeros_like(r)
    na = normalize([np.full_like(r, -d), ((x1 - x0).astype(f) / HEIGHT_SCALE).astype(f), zero])   # :64
    nb = normalize([zero, ((z1 - z0).astype(f) / HEIGHT_SCALE).astype(f), np.full_like(r, d)])    # :65
    cx = (na[1] * nb[2] - na[2] * nb[1]).astype(f)
    cy = (na[2] * nb[0] - na[0] * nb[2]).astype(f)
    cz = (na[0] * nb[1] - na[1] * nb[0]).astype(f)
    nn = normalize([cx, cy, cz])                                                                  # :66
    out = np.zeros(rgba.shape[:2] + (4,), f)
    out[..., 0], out[..., 1], out[..., 2] = nn
    return out


def normals_f64(rgba: np.ndarray, channel: int = 0) -> np.ndarray:
    r = rgba[..., channel].astype(np.float64)
    n = r.shape[0]
    dx = (np.roll(r, -1, axis=1) - np.roll(r, 1, axis=1)) / 180.0
    dz = (np.roll(r, -1, axis=0) - np.roll(r, 1, axis=0)) / 180.0
    d = 2.0 / n
    na = np.stack([np.full_like(r, -d), dx, np.zeros_like(r)], -1)
    nb = np.stack([np.zeros_like(r), dz, np.full_like(r, d)], -1)
    na /= np.linalg.norm(na, axis=-1, keepdims=True)
    nb /= np.linalg.norm(nb, axis=-1, keepdims=True)
    c = np.cross(na, nb)
    c /= np.linalg.norm(c, axis=-1, keepdims=True)
    out = np.zeros(rgba.shape[:2] + (4,))
    out[..., :3] = c
    return out


# --------------------------------------------------------------------------
# 8f #2: vertex-stage positions, shader/ocean.vert:21-25 with the sampler of src/render.rs:398
# --------------------------------------------------------------------------
def positions_f64(rgba: np.ndarray, verts: int = 128, offset=(0.0, 0.0)) -> np.ndarray:
    """pos = a_Pos + texture(map, a_Uv).xyz / (3.5, 3.0, 3.5) + (offset.x, 0, offset.y); a_Pos = (x, 0, z),
    a_Uv = (x, z) / (verts - 1) (src/render.rs:494-508); Filter::Linear + WrapMode::Tile = bilinear at texel
    coordinates uv * N - 0.5 with wrap (ideal weights; real samplers quantise them to ~8 bits)."""
    n = rgba.shape[0]
    r = rgba.astype(np.float64)
    g = np.arange(verts, dtype=np.float32) / np.float32(verts - 1)   # `(x as f32) / (V - 1) as f32`, src/render.rs:503-504
    t = g.astype(np.float32) * np.float32(n) - np.float32(0.5)
    t = t.astype(np.float64)
    f = np.floor(t)
    w = t - f
    i0 = f.astype(np.int64) % n
    i1 = (i0 + 1) % n
    X0, X1, WX = i0[None, :], i1[None, :], w[None, :, None]
    Y0, Y1, WY = i0[:, None], i1[:, None], w[:, None, None]
    s = (r[Y0, X0] * (1 - WX) * (1 - WY) + r[Y0, X1] * WX * (1 - WY) + r[Y1, X0] * (1 - WX) * WY + r[Y1, X1] * WX * WY)
    out = np.ones((verts, verts, 4))
    vx = np.arange(verts, dtype=np.float64)
    out[..., 0] = vx[None, :] + s[..., 0] / 3.5 + offset[0]
    out[..., 1] = s[..., 1] / 3.0
    out[..., 2] = vx[:, None] + s[..., 2] / 3.5 + offset[1]
    return out


# --------------------------------------------------------------------------
# Parity metric, SURVEY 8d: normalised max and relative L2 per channel.
# --------------------------------------------------------------------------
def parity_errors(a, b):
    """a: candidate, b: oracle; arrays [...,C] (or complex [...]).  -> (nmax, rel_l2) arrays per channel."""
    a = np.asarray(a)
    b = np.asarray(b)
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        a = np.stack([a.real, a.imag], -1)
        b = np.stack([b.real, b.imag], -1)
        a = a.reshape(-1, 1, 2).reshape(-1, 2).reshape(-1)[:, None]
        b = b.reshape(-1, 1, 2).reshape(-1, 2).reshape(-1)[:, None]
    a = a.reshape(-1, a.shape[-1]).astype(np.float64)
    b = b.reshape(-1, b.shape[-1]).astype(np.float64)
    nmax, rl2 = [], []
    for c in range(b.shape[1]):
        den_m = np.max(np.abs(b[:, c]))
        den_2 = np.linalg.norm(b[:, c])
        if den_m == 0.0:
            nmax.append(float(np.max(np.abs(a[:, c]))))
            rl2.append(float(np.linalg.norm(a[:, c])))
        else:
            nmax.append(float(np.max(np.abs(a[:, c] - b[:, c])) / den_m))
            rl2.append(float(np.linalg.norm(a[:, c] - b[:, c]) / den_2))
    return np.array(nmax), np.array(rl2)
