"""The product's kernel sources (gfx_ocean_amd/csrc/*.hpp) executed on the CPU by the host
emulation harness (tests/hipemu) and compared with the oracle: index algebra, plans, LDS
exchanges, chunked intermediate layout, quirks.  The GPU tier (-m gpu) repeats this on gfx950."""
import numpy as np
import pytest

import emu
import gfx_ocean_amd as g
from conftest import assert_parity
from oracle import ocean_oracle as oc


@pytest.fixture(scope="module", autouse=True)
def _build():
    """Both builds of the emulation at once (each is a minute of g++): the default one, and the one with the LDS-DMA loader at
    every size that test_emu_dma_loader_at_every_size runs in its own process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1] + '/tests'); import emu; emu.build()", root],
                             env=dict(os.environ, OCEAN_EMU_FLAGS="-DOCEAN_DMA_MIN_N=256"))
    emu.build()
    assert other.wait() == 0


def test_emu_propagate_matches_literal(ref_inputs_256):
    h0, om = ref_inputs_256
    for t in (0.0, 3.0, 1000.0):
        ref = oc.propagate_literal(h0, om, t)
        got = emu.propagate(h0, om, t)
        for a, b in zip(got, ref):
            assert oc.parity_errors(a, b)[0].max() <= 1e-6


@pytest.mark.parametrize("quirks", [0, 1, 2, 3])
def test_emu_propagate_quirk_switches(ref_inputs_256, quirks):
    """Q1 (uint wave index) and Q2 (partner N-1-g, not conjugated) are switchable; 3 = reference."""
    h0, om = ref_inputs_256
    got = emu.propagate(h0, om, 2.0, quirks=quirks)
    ref = oc.propagate_f64(h0, om, 2.0, quirks=quirks)
    for a, b, name in zip(got, ref, ("height", "disp_x", "disp_z")):
        assert_parity(a, b, 2e-6, f"quirks={quirks} {name}")
    if quirks != 3:       # and they do change the result
        refq = oc.propagate_f64(h0, om, 2.0)
        assert max(oc.parity_errors(a, b)[0].max() for a, b in zip(got, refq)) > 1e-2


def test_quirk_free_spectrum_of_a_hermitian_field_gives_a_real_surface():
    """With Q1 and Q2 off the propagation is the textbook one: if h0 is Hermitian under the (N+1-g) % N pairing
    the propagated spectra are too, except on the two self-paired lines g in {0, 1} x anything."""
    n = 64
    rng = np.random.default_rng(7)
    a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
    p = (n + 1 - np.arange(n)) % n
    h0 = ((a + np.conj(a[np.ix_(p, p)])) / 2).astype(np.complex64)
    om = rng.uniform(0.1, 3.0, (n, n)).astype(np.float32)
    om = ((om + om[np.ix_(p, p)]) / 2).astype(np.float32)
    h, dx, dz = oc.propagate_f64(h0, om, 1.5, quirks=0)
    assert np.abs(h - np.conj(h[np.ix_(p, p)])).max() < 1e-6 * np.abs(h).max()
    for f in (dx, dz):                                   # k(p(g)) = -k(g) needs g >= 2 on both axes
        d = (f - np.conj(f[np.ix_(p, p)]))[2:, 2:]
        assert np.abs(d).max() < 1e-6 * np.abs(f).max()
    # ... which is not the case with the reference's quirks
    hq = oc.propagate_f64(h0, om, 1.5)[0]
    assert np.abs(hq - np.conj(hq[np.ix_(p, p)])).max() > 1e-2 * np.abs(hq).max()


@pytest.mark.parametrize("n,col", [(256, 0), (256, 1), (512, 0), (512, 1), (1024, 0), (1024, 1)])
def test_emu_fft_lines(n, col):
    rng = np.random.default_rng(100 + n + col)
    x = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
    got = emu.fft_lines(x, col)
    ref = oc.ifft_lines_f64(x.T if col else x)
    ref = ref.T if col else ref
    assert_parity(got, ref, 2e-6, f"emu fft n={n} col={col}")


@pytest.mark.parametrize("nf,s", [(2048, 8)])                        # ((4096, 16) passes too: 30 s; the GPU tier runs S = 16 and 32)
def test_emu_two_step_column_pass(nf, s):
    """The staged column pass of N >= 8192 (k_cols4_a: sub-transforms of sixteen columns + twiddles, in place; k_cols4_b: the
    S-point step over consecutive rows, out of place) with the product's S = 8 and 16 at sizes the emulation can run."""
    rng = np.random.default_rng(nf)
    x = np.zeros((nf, nf), np.complex64)
    cols = rng.choice(nf, 48, replace=False)              # a few columns carry data (the reference transform of all of them is slow)
    x[:, cols] = (rng.standard_normal((nf, 48)) + 1j * rng.standard_normal((nf, 48))).astype(np.complex64)
    got = emu.cols4(x, s)
    ref = oc.ifft_lines_f64(np.ascontiguousarray(x[:, cols].T)).T
    assert_parity(got[:, cols], ref, 2e-6, f"two-step column pass nf={nf} s={s}")
    rest = np.delete(got, cols, axis=1)
    assert np.all(rest == 0)


def test_emu_two_step_column_pass_with_the_correction():
    """k_cols4_b_correct (the deferred second step of the three column passes + correction.comp:24-35 in one kernel) is the map
    k_correct makes of the three k_cols4_b outputs, bit for bit (real parts of the same S-point step, sign by the row's parity)."""
    nf, s = 1024, 4                                         # (2048, 8) passes too: 47 s; the GPU tier runs S = 16 and 32
    rng = np.random.default_rng(7)
    fields = []
    for i in range(3):
        x = np.zeros((nf, nf), np.complex64)
        cols = np.concatenate([[0, 1, nf - 1], rng.choice(nf, 13, replace=False)])
        x[:, cols] = (rng.standard_normal((nf, len(cols))) + 1j * rng.standard_normal((nf, len(cols)))).astype(np.complex64)
        fields.append(x)
    got = emu.cols4_correct(*fields, s)
    h, dx, dz = (emu.cols4(x, s) for x in fields)
    want = oc.correction_literal(h, dx, dz)
    assert np.array_equal(got, want)
    assert np.all(got[..., 3] == 0)


def test_emu_correct(ref_inputs_256):
    h0, om = ref_inputs_256
    h, dx, dz = oc.propagate_literal(h0, om, 1.0)
    assert np.array_equal(emu.correct(h, dx, dz), oc.correction_literal(h, dx, dz))


@pytest.mark.parametrize("channel", [0, 1])
def test_emu_normals(ref_inputs_256, channel):
    """k_normals vs the restatement of shader/ocean.frag:50-66."""
    h0, om = ref_inputs_256
    rgba = oc.frame_literal(h0, om, 2.0)
    got = emu.normals(rgba, channel)
    ref = oc.normals_literal(rgba, channel)
    assert np.abs(got - ref).max() <= 2e-6
    assert np.abs(got[..., :3] - oc.normals_f64(rgba, channel)[..., :3]).max() <= 2e-6
    assert np.allclose(np.linalg.norm(got[..., :3], axis=-1), 1.0, atol=1e-6) and np.all(got[..., 3] == 0)


@pytest.mark.parametrize("n,channel,split", [(256, 0, False), (256, 1, False), (512, 2, False), (512, 0, True)])
def test_emu_frame_with_normal_plane(n, channel, split, ref_inputs, ref_inputs_256):
    """The frame with the normal field (ocean_set_frame_normals): the PLANE instances of both pass-2 kernels store the very floats of
    the map's source channel, and k_normals_plane gives the normals of k_normals bit for bit."""
    h0, om = ref_inputs_256 if n == 256 else ref_inputs
    rgba, plane = emu.frame_half(h0, om, 2.0, plane_channel=channel, split=split)
    assert np.array_equal(plane, rgba[..., channel])
    got = emu.normals_plane(plane)
    assert np.array_equal(got, emu.normals(rgba, channel))
    assert np.abs(got - oc.normals_literal(rgba, channel)).max() <= 2e-6


@pytest.mark.parametrize("n,channel", [(1024, 0), (2048, 1)])
def test_emu_normals_plane_rows_per_wave_variants(n, channel):
    """k_normals_plane with 4 and 8 rows per wave (N = 1024, 2048) on a random map: the normals of k_normals bit for bit, wrap included."""
    rng = np.random.default_rng(n)
    rgba = rng.standard_normal((n, n, 4)).astype(np.float32) * 3.0
    got = emu.normals_plane(np.ascontiguousarray(rgba[..., channel]))
    assert np.array_equal(got, emu.normals(rgba, channel))
    assert np.abs(got - oc.normals_literal(rgba, channel)).max() <= 2e-6
    if n == 2048:                                        # the kernel of N >= 8192: one band of whole rows per workgroup
        assert np.array_equal(emu.normals_plane(np.ascontiguousarray(rgba[..., channel]), bands=True), got)


@pytest.mark.parametrize("n", [256])
def test_emu_frame_batch(n, ref_inputs, ref_inputs_256):
    """ocean_frame_batch at the latency-bound sizes: K time steps as blockIdx.y of ONE launch pair, every frame with its own
    intermediate, Nyquist scratch and map -- each bit-identical to the plain frame at t0 + dt * i (fp32, no FMA)."""
    h0, om = ref_inputs_256 if n == 256 else ref_inputs
    t0, dt, K = np.float32(1.5), np.float32(1.0 / 60.0), 3
    outs, planes = emu.frame_half(h0, om, float(t0), batch=(K, float(dt)), plane_channel=1)
    assert outs.shape == (K, n, n, 4) and np.array_equal(planes, outs[..., 1])      # one source-channel plane per frame of the batch
    for i in range(K):
        ti = np.float32(t0 + np.float32(dt * np.float32(i)))
        assert np.array_equal(outs[i], emu.frame_half(h0, om, float(ti))), i
    assert not np.array_equal(outs[0], outs[1])


def test_emu_frame_tiles(ref_inputs_256):
    """ocean_frame_tiles: K independent tiles as blockIdx.y of one launch pair, each with its own transposed inputs."""
    h0, om = ref_inputs_256
    h1, o1 = g.synth.make_inputs(256, seed=6)
    outs = emu.frame_half(np.stack([h0, h1]), np.stack([om, o1]), 1.5, batch=(2, 0.0))
    assert np.array_equal(outs[0], emu.frame_half(h0, om, 1.5)) and np.array_equal(outs[1], emu.frame_half(h1, o1, 1.5))
    assert not np.array_equal(outs[0], outs[1])


def test_emu_split_line_geometry_block_layout(ref_inputs):
    """The split kernels (N = 8192 geometry) with the intermediate in blocks of 8 chunk rows."""
    h0, om = ref_inputs
    out = emu.frame_half(h0, om, 2.5, split=True, bshift=3)
    assert_parity(out[..., :3], oc.frame_f64(h0, om, 2.5)[..., :3], 5e-6, "split frame, blocks of 8 chunk rows")


@pytest.mark.parametrize("n", [512])                               # (1024 passes too: 46 s of host emulation; the GPU tier runs 8192)
def test_emu_split_line_geometry(n, ref_inputs):
    """The N = 8192 kernels (every line as two interleaved N/2 transforms, last radix-2 step at read-out) at sizes
    the emulation can run: same frame and same intermediate as the plain kernels."""
    if n == 512:
        h0, om = ref_inputs
    else:
        import gfx_ocean_amd as g
        h0, om = g.synth.make_inputs(n)
    out, inter, _, (P, lay) = emu.frame_half(h0, om, 2.5, return_inter=True, split=True)
    ref = oc.frame_f64(h0, om, 2.5)
    assert_parity(out[..., :3], ref[..., :3], 5e-6, f"split frame n={n}")
    assert np.all(out[..., 3] == 0.0)
    plain, inter_p, _, _ = emu.frame_half(h0, om, 2.5, return_inter=True, P=2)
    for f in range(3):
        a = emu.unpack_inter(inter, n, P, lay, f, columns=n // 2, cmajor=True)
        b = emu.unpack_inter(inter_p, n, 2, lay, f, columns=n // 2)
        assert_parity(a, b, 5e-6, f"split intermediate field {f}")


@pytest.mark.parametrize("n", [1024])                              # (2048 passes too; 170 s of host emulation)
def test_emu_split_one_column_per_workgroup(n):
    """The N = 16384 geometry of the split kernels (two N/2-point sub-lines fill the LDS: ONE column per pass-1 workgroup,
    8-byte chunk pieces, the ring with the left neighbour's lines as half of what it stages) at sizes the emulation can
    run: same frame as the oracle, same intermediate as two columns per workgroup; fp16-stored spectrum too."""
    import gfx_ocean_amd as g                                       # (a workgroup is whole waves from N = 1024 on: 2 x 32 threads)
    h0, om = g.synth.make_inputs(n, seed=5)
    out, inter, _, (P, lay) = emu.frame_half(h0, om, 2.5, return_inter=True, split=True, P=1)
    assert P == 1
    assert_parity(out[..., :3], oc.frame_f64(h0, om, 2.5)[..., :3], 5e-6, f"split frame, one column per workgroup, n={n}")
    assert np.all(out[..., 3] == 0.0)
    for f in range(3):                                              # ... and the intermediate is the plain kernels'
        _, inter_p, _, _ = (None, None, None, None) if f else emu.frame_half(h0, om, 2.5, return_inter=True, P=2)
        if f == 0:
            plain = inter_p
        assert_parity(emu.unpack_inter(inter, n, P, lay, f, columns=n // 2, cmajor=True), emu.unpack_inter(plain, n, 2, lay, f, columns=n // 2), 5e-6,
                      f"intermediate field {f}")
    # (the fp16-stored spectrum with this geometry: tests/test_gpu_parity.py::test_fp16_spectrum_16384_sampled_texels; here 20 s more)


def test_emu_split_fp16_spectrum(ref_inputs):
    h0, om = ref_inputs
    _, deq, _ = emu.quantize_f16(h0)
    out = emu.frame_half(h0, om, 1.0, spectrum_fp16=True, split=True)
    assert_parity(out[..., :3], oc.frame_f64(deq, om, 1.0)[..., :3], 5e-6, "split fp16")


@pytest.mark.parametrize("verts,offset", [(128, (0.0, 0.0)), (128, (127.0, 127.0)), (33, (5.0, -2.0))])
def test_emu_positions(ref_inputs_256, verts, offset):
    """k_positions vs the restatement of shader/ocean.vert:21-25 (bilinear sampler with Tile wrap)."""
    h0, om = ref_inputs_256
    rgba = oc.frame_literal(h0, om, 2.0)
    got = emu.positions(rgba, verts, offset)
    ref = oc.positions_f64(rgba, verts, offset)
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max() / 8)      # fp32 on coordinates up to ~260
    assert np.all(got[..., 3] == 1.0)
    # the last grid line samples uv = 1: wraps onto the first texel column, as the Tile sampler does
    assert np.abs((got[:, -1, 1] - got[:, 0, 1])).max() < 1e-6


@pytest.mark.parametrize("P", [4, 2])
@pytest.mark.parametrize("t", [0.0, 3.0])
def test_emu_half_spectrum_frame(ref_inputs_256, t, P):
    """The shipped fused path (real-output algorithm: half the column FFTs, two row FFTs), with
    4 or 2 lines per pass-1 workgroup (whole chunks vs half chunk rows)."""
    h0, om = ref_inputs_256
    out = emu.frame_half(h0, om, t, P=P)
    assert not np.isnan(out).any()
    assert_parity(out[..., :3], oc.frame_f64(h0, om, t)[..., :3], 5e-6, "emu half-spectrum frame")
    assert np.all(out[..., 3] == 0.0)


@pytest.mark.parametrize("domain", [250.0, 1.0e13, 1.0e21])
def test_emu_domain_size_and_the_zero_wave_vector_guard(ref_inputs_256, domain):
    """PropagateLocals.domain_size: k = pi x / L is normalised behind `length(k) > 1e-10` (shader/propagate.comp:64-67), so L
    cancels unless the guard trips -- for part of the quadrant whose wave index does not wrap at L = 1e13, for every texel at
    1e21 (both displacement channels exactly zero).  The fused kernels' `|k|^2 > 1e-20` against the C restatement's literal
    guard (the GPU tier repeats this at 512 / 2048 / 8192 for both paths)."""
    from oracle import c_oracle as cc
    h0, om = ref_inputs_256
    ref = cc.FrameRunner(h0, om, domain_size=domain).frame(1.5).copy()
    out = emu.frame_half(h0, om, 1.5, L=domain)
    assert_parity(out[..., 1:2], ref[..., 1:2], 5e-6, "height")
    if domain >= 1.0e21:
        assert np.all(ref[..., (0, 2)] == 0.0) and np.all(out[..., (0, 2)] == 0.0)
    else:
        assert_parity(out[..., :3], ref[..., :3], 5e-6, f"emu half-spectrum frame, L = {domain}")
    if domain == 1.0e13:
        plain = cc.FrameRunner(h0, om).frame(1.5)
        assert np.abs(ref[..., 0] - plain[..., 0]).max() > 1e-3 * np.abs(plain[..., 0]).max()     # the guard changed the frame


@pytest.mark.parametrize("P", [2, 1])
def test_emu_half_spectrum_frame_512(ref_inputs, P):
    """N = 512 as shipped: ONE column per pass-1 workgroup, three field groups (FPAR), stores from registers; one row per
    pass-2 workgroup with 8 elements per thread and two transform groups (PPAR).  And two columns per workgroup."""
    h0, om = ref_inputs
    assert_parity(emu.frame_half(h0, om, 10.0, P=P)[..., :3], oc.frame_f64(h0, om, 10.0)[..., :3], 5e-6, f"emu half 512 P={P}")


def test_emu_fp16_spectrum_config5(ref_inputs_256):
    """BASELINE config 5 semantics at a small N: fp16-stored h0 (scaled), fp32 arithmetic; parity
    against the oracle fed the SAME quantised inputs, and the quantisation error itself is ~3e-4."""
    h0, om = ref_inputs_256
    _, deq, s = emu.quantize_f16(h0)
    assert 2 ** 14 <= np.abs(h0.view(np.float32)).max() * 2.0 ** s < 2 ** 15
    out = emu.frame_half(h0, om, 1.0, spectrum_fp16=True)
    assert_parity(out[..., :3], oc.frame_f64(deq, om, 1.0)[..., :3], 5e-6, "fp16 spectrum vs oracle on quantised inputs")
    nmax, rl2 = oc.parity_errors(out[..., :3], oc.frame_f64(h0, om, 1.0)[..., :3])
    assert 1e-5 < rl2.max() < 2e-3          # fp16 rounding of the inputs is visible, as SURVEY 7 predicts


@pytest.mark.parametrize("bshift", [0, 2, 30])
@pytest.mark.parametrize("P", [4, 2])
def test_emu_half_intermediate_layout(ref_inputs_256, P, bshift):
    """k_half_pass1 writes columns kx < N/2 of FFT_y(2 S(F)) as 4 x 4 chunks (128 bytes), whole or in
    halves; column 0 carries two real columns, (kx = 0, kx = N/2) as (re, im), and the scratch holds the
    Nyquist column's three symmetrised spectra.  bshift: the chunk rows in blocks of 1 (pass-2-contiguous), 4, or
    all of them (pass-1-contiguous, what N = 4096 ships) -- the frame must not care."""
    h0, om = ref_inputs_256
    n = 256
    out, inter, nyq, (P_, lay) = emu.frame_half(h0, om, 2.0, return_inter=True, P=P, bshift=bshift)
    assert_parity(out[..., :3], oc.frame_f64(h0, om, 2.0)[..., :3], 5e-6, f"frame, blocks of 2^{bshift} chunk rows")
    H, DX, DZ = oc.propagate_f64(h0, om, 2.0)
    for f, F in ((0, DX), (1, H), (2, DZ)):
        Fm = np.conj(np.roll(np.roll(F[::-1, ::-1], 1, axis=0), 1, axis=1))
        G = np.fft.ifft(F + Fm, axis=0) * n                       # 2 S(F), transformed along y
        got = emu.unpack_inter(inter, n, P, lay, f, columns=n // 2)
        want = G[:, :n // 2].copy()
        assert np.abs(G[:, 0].imag).max() < 1e-9 * np.abs(G).max() and np.abs(G[:, n // 2].imag).max() < 1e-9 * np.abs(G).max()
        want[:, 0] = G[:, 0].real + 1j * G[:, n // 2].real        # the two real columns share one transform
        assert_parity(got, want, 5e-6, f"half intermediate field {f}")
        spec = nyq.view(np.complex64)[f * n:(f + 1) * n]
        assert_parity(spec[:, None], (F + Fm)[:, n // 2][:, None], 5e-6, f"nyquist spectrum field {f}")
    assert np.isnan(inter.real).sum() >= 3 * (lay[2] - n * n // 2)


@pytest.mark.parametrize("n", [256, 512])
def test_emu_staged_chunked_handoff(n, ref_inputs, ref_inputs_256):
    """The staged path's chunked hand-off (k_stage_rows -> k_stage_cols -> k_correct_chunked / k_unchunk, N <= 4096):
    row pass and column pass against the fp64 line transforms, the correction against the literal shader."""
    h0, om = ref_inputs if n == 512 else ref_inputs_256
    fields = oc.propagate_literal(h0, om, 2.5)                       # height, disp_x, disp_z
    rows_only, _ = emu.staged_chunked(fields, do_cols=False)
    for got, f in zip(rows_only, fields):
        assert_parity(got, oc.ifft_lines_f64(f), 5e-6, f"chunked row pass n={n}")
    both, rgba = emu.staged_chunked(fields)
    refs = [oc.ifft_lines_f64(oc.ifft_lines_f64(f).T).T for f in fields]
    for got, ref in zip(both, refs):
        assert not np.isnan(got.view(np.float32)).any()
        assert_parity(got, ref, 5e-6, f"chunked row + column pass n={n}")
    want = oc.correction_literal(*[b for b in both])                 # same inputs, the literal correction
    assert np.array_equal(rgba, want)


def test_emu_dma_loader_at_every_size():
    """The LDS-DMA loader of fused pass 1 (half_load_AB_dma: N >= 2048 in the product) in a build of the emulation that
    enables it at every size, so that the ring's index algebra (pieces, slots, the three edge values carried from piece
    to piece, the deferred element of piece 0), its barrier protocol and its 1 KiB instruction slots are checked against
    the oracle at sizes the CPU can run: whole lines with four columns per workgroup (256: one piece of eight elements),
    two (512) and two with one-wave lines (1024, four pieces); the split geometry at 512 / 1024; fp32 and fp16-stored
    spectrum.  Own process: the flags are read at import."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import emu, gfx_ocean_amd as g
from oracle import ocean_oracle as oc
from conftest import GOLDEN, assert_parity
h0, om = oc.load_reference_inputs(GOLDEN + "/spectrum.bin", GOLDEN + "/omega.bin")
h256, o256 = oc.centre_crop(h0, 256), oc.centre_crop(om, 256)
h1k, o1k = g.synth.make_inputs(1024, seed=4)
for (h, o, kw, what) in ((h0, om, dict(split=True), "split 512"),
                         (h256, o256, dict(), "lines 256 P=4"), (h0, om, dict(P=2), "lines 512 P=2"), (h1k, o1k, dict(), "lines 1024")):
    assert emu.uses_dma(h.shape[0], kw.get("P"), kw.get("split", False)), what
    out = emu.frame_half(h, o, 2.5, **kw)
    assert_parity(out[..., :3], oc.frame_f64(h, o, 2.5)[..., :3], 5e-6, what)
_, deq, _ = emu.quantize_f16(h0)
for kw in (dict(split=True), dict(P=2)):
    out = emu.frame_half(h0, om, 1.0, spectrum_fp16=True, **kw)
    assert_parity(out[..., :3], oc.frame_f64(deq, om, 1.0)[..., :3], 5e-6, "fp16 spectrum " + str(kw))
print("DMA_EMU_OK")
"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OCEAN_EMU_FLAGS="-DOCEAN_DMA_MIN_N=256")
    p = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0 and "DMA_EMU_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_emu_default_build_loads_small_sizes_into_registers():
    """... and the default build keeps the latency-bound sizes on the register loader (half_load_AB); the split kernels
    always use the ring."""
    for n in (256, 512, 1024):
        assert not emu.uses_dma(n) and not emu.uses_dma(n, 2)
    assert emu.uses_dma(512, split=True)


@pytest.mark.parametrize("n", [512])
def test_emu_bfp16_intermediate(n, ref_inputs):
    """The opt-in 16-bit block-floating intermediate (OCEAN_INTER_BFP16; split kernels): int16 mantissas, one power-of-two
    scale per wave store.  Against the fp32-intermediate frame the error is the quantisation's (a few 1e-5 normalised max,
    tolerance 1e-4), and it is really different from it (not silently the fp32 path)."""
    h0, om = ref_inputs if n == 512 else g.synth.make_inputs(n, seed=8)
    ref = oc.frame_f64(h0, om, 2.5)[..., :3]
    out32 = emu.frame_half(h0, om, 2.5, split=True)[..., :3]
    out16 = emu.frame_half(h0, om, 2.5, split=True, inter16=True)[..., :3]
    assert not np.isnan(out16).any()
    nmax, rl2 = assert_parity(out16, ref, 1e-4, f"bfp16 intermediate n={n}")
    assert nmax.max() < 6e-5
    assert oc.parity_errors(out16, out32)[0].max() > 2e-6             # quantised: not the fp32 intermediate
