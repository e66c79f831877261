"""GPU tier: the HIP path, called through the C ABI, against the oracle.

Tolerance: north_star states "within 1e-4 relative fp32"; measured per channel as normalised max
and relative L2, both <= 1e-4 against the fp64 oracle.  Stage-level checks use the
fp32 literal oracle where the comparison is elementwise.
"""
import os

import numpy as np
import pytest

import gfx_ocean_amd as g
from conftest import GOLDEN, assert_parity
from oracle import c_oracle as cc
from oracle import ocean_oracle as oc

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def r512(ref_inputs):
    r = g.OceanRenderer(512)
    r.upload(*ref_inputs)
    yield r
    r.dispose()


@pytest.fixture(scope="module")
def r256(ref_inputs_256):
    r = g.OceanRenderer(256)
    r.upload(*ref_inputs_256)
    yield r
    r.dispose()


def test_native_library_is_loaded():
    lib = g.load_library()
    assert lib.ocean_abi_version() == 4
    with open("/proc/self/maps") as f:
        assert "libocean_hip.so" in f.read()


@pytest.mark.parametrize("t", [0.0, 1.0, 10.0, 100.0])
def test_staged_frame_reference_data_512(r512, ref_inputs, t):
    """config 2: N=512, data/*.bin, the reference's 8-dispatch sequence."""
    r512.render(t)
    out = r512.displacement()
    assert_parity(out[..., :3], oc.frame_f64(*ref_inputs, t)[..., :3], TOL, f"staged t={t}")
    assert np.all(out[..., 3] == 0.0)


@pytest.mark.parametrize("t", [0.0, 1.0, 10.0, 100.0, 1000.0])
def test_fused_frame_reference_data_512(r512, ref_inputs, t):
    r512.render_fused(t)
    out = r512.displacement()
    nmax, rl2 = assert_parity(out[..., :3], oc.frame_f64(*ref_inputs, t)[..., :3], TOL, f"fused t={t}")
    assert nmax.max() < 1e-5          # fp32 headroom: expect ~1e-6
    assert np.all(out[..., 3] == 0.0)


def test_stage_by_stage_512(r512, ref_inputs):
    """Each dispatch against the literal fp32 oracle of the same stage."""
    h0, om = ref_inputs
    t = 3.25
    n = 512
    dev = r512.device
    r512.propagation.dispatch(g.PropagateLocals(t, n))
    h, dx, dz = oc.propagate_literal(h0, om, t)
    for f, ref in ((g.FIELD_DY, h), (g.FIELD_DX, dx), (g.FIELD_DZ, dz)):
        assert_parity(dev.read_field(f), ref, 2e-6, f"propagate field {f}")
    r512.fft.row_pass(g.FIELD_ALL)
    rows = {f: oc.ifft_lines_f64(v) for f, v in ((g.FIELD_DY, h), (g.FIELD_DX, dx), (g.FIELD_DZ, dz))}
    for f, ref in rows.items():
        assert_parity(dev.read_field(f), ref, 5e-6, f"row pass field {f}")
    r512.fft.col_pass(g.FIELD_DX)           # single-set dispatch, like one loop iteration of render.rs:1210-1231
    got = dev.read_field(g.FIELD_DX)
    ref = oc.ifft_lines_f64(rows[g.FIELD_DX].T).T
    assert_parity(got, ref, 5e-6, "col pass dx")
    assert_parity(dev.read_field(g.FIELD_DY), rows[g.FIELD_DY], 5e-6, "dy untouched by col_pass(dx)")
    r512.fft.col_pass(g.FIELD_DY)
    r512.fft.col_pass(g.FIELD_DZ)
    r512.correction.dispatch(g.CorrectionLocals(n))
    assert_parity(r512.displacement()[..., :3], oc.frame_f64(h0, om, t)[..., :3], TOL, "after correction")


def test_staged_field_layout_state_machine(r512):
    """The staged calls hand a field from the row pass to the column pass in the chunked layout (N <= 4096,
    k_stage_rows / k_stage_cols); ocean_read_field / ocean_write_field and every order of calls must still behave
    as the reference's in-place passes on a natural buffer: read after any stage, inject then transform columns
    first, transform rows twice."""
    n = 512
    dev = r512.device
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
    rows_of = lambda v: oc.ifft_lines_f64(v)                        # noqa: E731
    cols_of = lambda v: oc.ifft_lines_f64(np.ascontiguousarray(v.T)).T   # noqa: E731
    # rows -> read (natural copy made, chunked copy still current) -> cols on the chunks -> read
    dev.write_field(g.FIELD_DX, a)
    r512.fft.row_pass(g.FIELD_DX)
    assert_parity(dev.read_field(g.FIELD_DX), rows_of(a), 5e-6, "rows")
    r512.fft.col_pass(g.FIELD_DX)
    assert_parity(dev.read_field(g.FIELD_DX), cols_of(rows_of(a)), 5e-6, "rows, read, cols")
    # inject, then the column pass first (natural-layout kernel), then the row pass
    dev.write_field(g.FIELD_DY, a)
    r512.fft.col_pass(g.FIELD_DY)
    assert_parity(dev.read_field(g.FIELD_DY), cols_of(a), 5e-6, "cols on an injected field")
    r512.fft.row_pass(g.FIELD_DY)
    assert_parity(dev.read_field(g.FIELD_DY), rows_of(cols_of(a)), 5e-6, "cols then rows")
    # two row passes in a row (the second starts from the chunked result of the first), without a read between
    dev.write_field(g.FIELD_DZ, a)
    r512.fft.row_pass(g.FIELD_DZ)
    r512.fft.row_pass(g.FIELD_DZ)
    assert_parity(dev.read_field(g.FIELD_DZ), rows_of(rows_of(a)), 5e-6, "rows twice")
    # correction with the three fields in three different states (chunked, natural, chunked)
    r512.fft.col_pass(g.FIELD_DZ)
    r512.correction.dispatch(g.CorrectionLocals(n))
    want = oc.correction_literal(dev.read_field(g.FIELD_DY), dev.read_field(g.FIELD_DX), dev.read_field(g.FIELD_DZ))
    assert np.array_equal(r512.displacement(), want)


def test_config1_n256_centre_crop(r256, ref_inputs_256):
    for t in (0.0, 1.0):
        r256.render_fused(t)
        fused = r256.displacement()
        r256.render(t)
        staged = r256.displacement()
        ref = oc.frame_f64(*ref_inputs_256, t)
        assert_parity(fused[..., :3], ref[..., :3], TOL, "N=256 fused")
        assert_parity(staged[..., :3], ref[..., :3], TOL, "N=256 staged")


@pytest.mark.parametrize("name,n,t", [("frame512_t0", 512, 0.0), ("frame512_t1", 512, 1.0),
                                     ("frame512_t10", 512, 10.0), ("frame256_t1", 256, 1.0)])
def test_committed_golden_vectors(r512, r256, name, n, t):
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    r = r512 if n == 512 else r256
    r.render_fused(t)
    out = r.displacement()[..., :3]
    assert np.abs(out[:64, :64] - gold["crop"]).max() <= TOL * gold["max"].max()
    for (x, y), v in zip(gold["probes_xy"], gold["probes"]):
        assert np.allclose(out[y, x], v, atol=TOL * gold["max"].max())
    assert np.allclose(np.sqrt((out.astype(np.float64) ** 2).sum((0, 1))), gold["l2"], rtol=TOL)


@pytest.mark.parametrize("t", [0, 1, 10])
def test_against_the_reference_spirv_outputs(r512, t):
    """HIP (fused and staged) vs vectors obtained by executing the reference's shipped
    shader/spv/*.comp.spv on data/*.bin (tests/golden/make_spirv_golden.py)."""
    gold = np.load(os.path.join(GOLDEN, f"spirv_frame512_t{t}.npz"))
    scale = gold["max"].max()
    for mode in ("fused", "staged"):
        (r512.render_fused if mode == "fused" else r512.render)(float(t))
        out = r512.displacement()
        assert np.abs(out[:64, :64] - gold["crop"]).max() <= TOL * scale, mode
        assert np.abs(out[::8, ::8] - gold["sub8"]).max() <= TOL * scale, mode
        ch = out[..., :3].astype(np.float64)
        assert np.allclose(np.sqrt((ch ** 2).sum((0, 1))), gold["l2"], rtol=TOL), mode


@pytest.mark.parametrize("n", [1024, 2048])
def test_synthetic_against_c_oracle(n):
    """config 3 (N=2048, three iFFTs per frame) and N=1024: synthetic inputs, C oracle + fp64."""
    h0, om = g.synth.make_inputs(n)
    r = g.OceanRenderer(n)
    try:
        r.upload(h0, om)
        t = 2.5
        ref64 = oc.frame_f64(h0, om, t)
        refc = cc.FrameRunner(h0, om).frame(t)
        r.render_fused(t)
        fused = r.displacement()
        r.render(t)
        staged = r.displacement()
        assert_parity(fused[..., :3], ref64[..., :3], TOL, f"N={n} fused vs fp64")
        assert_parity(staged[..., :3], ref64[..., :3], TOL, f"N={n} staged vs fp64")
        assert_parity(fused[..., :3], refc[..., :3], TOL, f"N={n} fused vs C oracle")
    finally:
        r.dispose()


@pytest.mark.parametrize("n", [512, 2048, 8192])        # the register loader, the LDS-DMA loader, the split kernels
@pytest.mark.parametrize("domain", [250.0, 1.0e13, 1.0e21])
def test_domain_size_and_the_zero_wave_vector_guard(n, domain):
    """PropagateLocals.domain_size (src/ocean.rs:8-13; the reference always feeds 1000, src/render.rs:46,1110) scales the wave
    vector k = pi x / L, which shader/propagate.comp:64-67 normalises behind the guard `length(k) > 1e-10`: the scale cancels
    unless the guard trips.  L = 250: as 1000 up to rounding.  L = 1e13: the guard trips for part of the one quadrant whose wave
    index does not wrap (quirk Q1: |x| <= N there, 4.29e9 elsewhere).  L = 1e21: it trips for every texel -- both displacement
    channels are exactly zero, the height is untouched.  Both paths against the C restatement of the shaders, which evaluates
    the guard as the shader does."""
    h0, om = g.synth.make_inputs(n, seed=77)
    cc.set_threads(min(32, cc.max_threads()))
    refc = cc.FrameRunner(h0, om, domain_size=domain).frame(1.5).copy()
    ref1000 = cc.FrameRunner(h0, om).frame(1.5).copy()
    r = g.OceanRenderer(n, domain_size=domain)
    try:
        r.upload(h0, om)
        r.render_fused(1.5)
        fused = r.displacement()
        r.render(1.5)
        staged = r.displacement()
    finally:
        r.dispose()
    for name, out in (("fused", fused), ("staged", staged)):
        assert np.all(out[..., 3] == 0.0)
        assert_parity(out[..., 1:2], refc[..., 1:2], TOL, f"N={n} L={domain} {name} height")
        if domain >= 1.0e21:
            assert np.all(refc[..., 0] == 0.0) and np.all(refc[..., 2] == 0.0)       # the oracle: k_norm = 0 everywhere
            assert np.all(out[..., 0] == 0.0) and np.all(out[..., 2] == 0.0), f"N={n} {name}: the guard did not trip everywhere"
        else:
            assert_parity(out[..., :3], refc[..., :3], TOL, f"N={n} L={domain} {name}")
    if domain == 1.0e13:        # ... and the guard did change the frame (a test that cannot fail is not one)
        assert np.abs(refc[..., 0] - ref1000[..., 0]).max() > 1e-3 * np.abs(ref1000[..., 0]).max()
    if domain == 250.0:
        assert_parity(refc[..., :3], ref1000[..., :3], 1e-5, "the scale cancels in k / |k|")


@pytest.mark.parametrize("n", [4096, 8192])
def test_full_size_against_c_oracle(n):
    """BASELINE configs 4 and 5 at their full sizes, every texel: the fused frame against the C restatement of
    the four shaders (fp32, radix-2 Stockham, sincosf per butterfly) on the box's host cores."""
    h0, om = g.synth.make_inputs(n)
    cc.set_threads(min(32, cc.max_threads()))            # the strided column pass is fastest on ~32 threads
    refc = cc.FrameRunner(h0, om).frame(0.75)
    r = g.OceanRenderer(n)
    try:
        r.upload(h0, om)
        r.render_fused(0.75)
        fused = r.displacement()
        nmax, rl2 = assert_parity(fused[..., :3], refc[..., :3], TOL, f"N={n} fused vs C oracle")
        assert nmax.max() < 2e-5                          # two fp32 paths: expect a few 1e-6
        assert np.all(fused[..., 3] == 0.0)
    finally:
        r.dispose()


def test_full_size_large_time_4096():
    """Phases omega * t of thousands of radians at the headline size (the fused kernels reduce the fp32 phase to
    revolutions with a two-constant product before the hardware sin/cos): every texel against the C oracle's sincosf."""
    n, t = 4096, 1000.0
    h0, om = g.synth.make_inputs(n, seed=11)
    cc.set_threads(min(32, cc.max_threads()))
    refc = cc.FrameRunner(h0, om).frame(t)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om)
        d.frame(t)
        out = d.read_displacement()
        nmax, rl2 = assert_parity(out[..., :3], refc[..., :3], TOL, "N=4096 t=1000 fused vs C oracle")
        assert nmax.max() < 3e-5
    finally:
        d.destroy()


@pytest.mark.parametrize("t", [2.0e4, 2.0e5, 2.0e6])
def test_phase_range_of_the_fused_propagate(t, ref_inputs):
    """the reference feeds wall-clock seconds (src/lib.rs:139-141) into `d = omega * t`
    (shader/propagate.comp:55-57); the fused kernels reduce that fp32 phase to revolutions with a two-constant product
    (propagate_height) before the hardware sin/cos.  On the reference's own inputs at t = 2e4 .. 2e6 s (|omega t| up to
    ~1e7 rad: 23 days of run time) the frame stays within the tolerance of the oracle, whose phase is the same
    fl32(omega * t) followed by a correctly rounded sin/cos -- the bound is stated next to ocean_frame in
    include/ocean_hip.h.  (Beyond that the PHASE ITSELF is the problem, for the reference as well: one ulp of
    fl32(omega t) at 1e7 rad is 1 rad.)  The staged propagate (ocml sincosf) is held to the same frames."""
    h0, om = ref_inputs
    assert float(np.abs(om).max()) * t > 0.9 * t                    # phases really reach ~ t radians
    ref = oc.frame_f64(h0, om, t)
    r = g.OceanRenderer(512)
    try:
        r.upload(h0, om)
        r.render_fused(t)
        fused = r.displacement()
        r.render(t)
        staged = r.displacement()
    finally:
        r.dispose()
    nmax_f, _ = assert_parity(fused[..., :3], ref[..., :3], TOL, f"fused, t = {t:g}")
    nmax_s, _ = assert_parity(staged[..., :3], ref[..., :3], TOL, f"staged, t = {t:g}")
    print(f"phase range t = {t:g}: |omega t| <= {float(np.abs(om).max()) * t:.3g} rad; normalised max fused {nmax_f.max():.2e}, staged {nmax_s.max():.2e}")
    assert nmax_f.max() < 2e-5 and nmax_s.max() < 2e-5


@pytest.mark.parametrize("n", [4096, 8192])
def test_full_size_properties(n):
    """BASELINE full sizes through size-independent properties (the oracle would take minutes):
    fused == staged; impulse spectrum -> closed-form plane wave; linearity in h0; FFT of the
    intermediate fields against fp64 on sampled lines."""
    rng = np.random.default_rng(n)
    h0, om = g.synth.make_inputs(n)
    r = g.OceanRenderer(n)
    try:
        r.upload(h0, om)
        t = 1.75
        r.render_fused(t)
        fused = r.displacement()
        r.render(t)
        staged = r.displacement()
        assert_parity(fused[..., :3], staged[..., :3], 2e-5, f"N={n} fused vs staged")
        assert np.all(fused[..., 3] == 0.0)
        # sampled texels against a direct fp64 evaluation of the 2-D sum (O(N^2) per texel)
        H, DX, DZ = oc.propagate_f64(h0, om, t)
        k = np.arange(n)
        scale = np.abs(fused[..., :3]).max((0, 1))
        for (x, y) in [(0, 0), (1, n - 1), (n // 2 + 3, n // 3), (n - 1, n - 1)]:
            ey, ex = np.exp(2j * np.pi * k * y / n), np.exp(2j * np.pi * k * x / n)
            sgn = -1.0 if (x + y) % 2 == 0 else 1.0
            ref = np.array([(ey @ (F @ ex)).real for F in (DX, H, DZ)]) * sgn
            assert np.all(np.abs(fused[y, x, :3] - ref) <= TOL * scale), (x, y, fused[y, x, :3], ref)
        # row pass of the staged path on random data, sampled lines vs fp64
        a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
        r.device.write_field(g.FIELD_DZ, a)
        r.fft.row_pass(g.FIELD_DZ)
        rows = r.device.read_field(g.FIELD_DZ)
        for yy in (0, 17, n - 1):
            assert_parity(rows[yy], oc.ifft_lines_f64(a[yy][None])[0], 5e-6, f"row {yy}")
        r.fft.col_pass(g.FIELD_DZ)
        both = r.device.read_field(g.FIELD_DZ)
        for xx in (0, 5, n - 1):
            assert_parity(both[:, xx], oc.ifft_lines_f64(rows[:, xx][None])[0], 5e-6, f"col {xx}")
        # Parseval over the whole 2-D transform
        assert abs((np.abs(both.astype(np.complex128)) ** 2).sum() / (float(n) ** 2 * (np.abs(a.astype(np.complex128)) ** 2).sum()) - 1) < 1e-5
    finally:
        r.dispose()


def test_fp16_spectrum_16384_sampled_texels():
    """The fp16-stored spectrum with the one-column split geometry (N = 16384; the emulation runs this geometry at 1024 with
    the fp32 spectrum only): sampled texels of ocean_frame against direct fp64 evaluations of the 2-D sum on the DEQUANTISED
    spectrum the kernels use."""
    n, t = 16384, 1.25
    h0, om = g.synth.make_inputs(n, seed=77)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=True)
        deq = d.read_spectrum()
        d.frame(t)
        out = d.read_displacement()
    finally:
        d.destroy()
    H, DX, DZ = oc.propagate_f64(deq, om, t)
    k = np.arange(n)
    scale = np.abs(out[..., :3]).max((0, 1))
    for (x, y) in [(0, 0), (n // 2 + 3, n // 3), (n - 1, n - 1)]:
        ey, ex = np.exp(2j * np.pi * k * y / n), np.exp(2j * np.pi * k * x / n)
        sgn = -1.0 if (x + y) % 2 == 0 else 1.0
        ref = np.array([(ey @ (F @ ex)).real for F in (DX, H, DZ)]) * sgn
        assert np.all(np.abs(out[y, x, :3] - ref) <= TOL * scale), (x, y, out[y, x, :3], ref)
    assert np.all(out[..., 3] == 0.0)


def test_staged_column_pass_16384():
    """The two-step staged column pass (k_cols4_a / k_cols4_b) with S = 16 at N = 16384: sampled columns of random data vs fp64,
    twice in a row (the field and its second buffer swap roles), rows untouched by a column pass of another field."""
    n = 16384
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n), dtype=np.float32).astype(np.complex64)
    cols = (0, 5, 4097, n - 1)
    for xx in cols:
        a[:, xx] += 1j * rng.standard_normal(n).astype(np.float32)
    d = g.OceanDevice(n)
    fft = g.Fft.init(d)
    try:
        d.write_field(g.FIELD_DZ, a)
        fft.col_pass(g.FIELD_DZ)
        once = d.read_field(g.FIELD_DZ)
        for xx in cols:
            assert_parity(once[:, xx], oc.ifft_lines_f64(a[:, xx][None])[0], 5e-6, f"col {xx}")
        fft.col_pass(g.FIELD_DZ)                                  # from the second buffer back into the first
        twice = d.read_field(g.FIELD_DZ)
        for xx in cols[:2]:
            assert_parity(twice[:, xx], oc.ifft_lines_f64(once[:, xx][None])[0], 5e-6, f"col {xx}, second pass")
    finally:
        fft.destroy()
        d.destroy()


def test_staged_deferred_column_step_8192():
    """N >= 8192: the second step of a column pass waits for the field's next consumer.  Behind the three column passes that is
    ocean_correct (k_cols4_b_correct: step 2 of the three fields + correction.comp in one kernel); ocean_read_field, a row pass,
    another column pass or a correction with the fields in mixed states run k_cols4_b first.  Every order gives the map the
    separate kernels give, bit for bit, and the fields read back as the reference's in-place passes would leave them."""
    n = 8192
    h0, om = g.synth.make_inputs(n)
    r = g.OceanRenderer(n)
    dev = r.device
    try:
        r.upload(h0, om)
        r.render(0.5)                                              # 8 dispatches: ..., cols x3 (step 1 each), correction (step 2 + sign)
        one_kernel = r.displacement()
        fields = {f: dev.read_field(f) for f in (g.FIELD_DY, g.FIELD_DX, g.FIELD_DZ)}     # each read settles its field (k_cols4_b)
        assert np.array_equal(one_kernel, oc.correction_literal(fields[g.FIELD_DY], fields[g.FIELD_DX], fields[g.FIELD_DZ]))
        r.correction.dispatch(g.CorrectionLocals(n))               # nothing pending now: k_correct on the settled fields
        assert np.array_equal(r.displacement(), one_kernel)
        del fields
        # mixed states: one field settled by a read, two pending
        r.render(0.5)
        dx = dev.read_field(g.FIELD_DX)
        r.correction.dispatch(g.CorrectionLocals(n))
        assert np.array_equal(r.displacement(), one_kernel)
        assert np.array_equal(dev.read_field(g.FIELD_DX), dx)      # ... and a second read finds the same field
        # the fields behind the one-kernel correction are still the column pass's results
        r.render(0.5)
        assert np.array_equal(dev.read_field(g.FIELD_DX), dx)
        del dx
        # column pass behind a column pass, then a row pass, without a read between (sampled lines vs fp64)
        rng = np.random.default_rng(n)
        a = np.zeros((n, n), np.complex64)
        cols = (0, 3, n // 2 + 1, n - 1)
        for xx in cols:
            a[:, xx] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        dev.write_field(g.FIELD_DY, a)
        r.fft.col_pass(g.FIELD_DY)
        r.fft.col_pass(g.FIELD_DY)
        twice = dev.read_field(g.FIELD_DY)
        for xx in cols:
            once = oc.ifft_lines_f64(a[:, xx][None])[0]
            assert_parity(twice[:, xx], oc.ifft_lines_f64(once[None])[0], 5e-6, f"col {xx} twice")
        dev.write_field(g.FIELD_DY, a)
        r.fft.col_pass(g.FIELD_DY)
        r.fft.row_pass(g.FIELD_DY)                                 # settles, then transforms rows
        both = dev.read_field(g.FIELD_DY)
        want = np.zeros((n, len(cols)), np.complex128)
        for i, xx in enumerate(cols):
            want[:, i] = oc.ifft_lines_f64(a[:, xx][None])[0]
        phase = np.exp(2j * np.pi * np.outer(np.arange(n), np.array(cols)) / n)       # row yy of the column-transformed field has
        for yy in (0, 1, n - 1):                                                      # four non-zero entries: its transform in closed form
            assert_parity(both[yy], (want[yy][None, :] * phase).sum(1), 5e-6, f"cols then rows, row {yy}")
        # a written field discards what was pending
        r.fft.col_pass(g.FIELD_DY)
        dev.write_field(g.FIELD_DY, a)
        assert np.array_equal(dev.read_field(g.FIELD_DY), a)
    finally:
        r.dispose()


def test_upload_spectrum_from_device_memory():
    """ocean_upload_spectrum_device = the upload's device-side half (copy_buffer, src/render.rs:896-915): a spectrum in device
    memory, or in host memory registered with the runtime (a mapped staging buffer), asynchronous and stream-ordered.  Same maps
    as the host upload bit for bit: fused-only and full contexts (the staged path sees it too), a tile of a context of several,
    a replacement between two frames without a host wait; pageable host memory is refused."""
    from hipmem import DeviceBuffer, PinnedHostBuffer, Stream
    n = 1024
    h0, om = g.synth.make_inputs(n, seed=5)
    h1, om1 = g.synth.make_inputs(n, seed=6)
    ref = g.OceanDevice(n)
    dh, do = DeviceBuffer(h0.nbytes), DeviceBuffer(om.nbytes)
    ph, po = PinnedHostBuffer(h1.nbytes), PinnedHostBuffer(om1.nbytes)
    fo = full = tiles = None
    try:
        ref.upload_spectrum(h0, om)
        ref.frame(0.7)
        want0 = ref.read_displacement()
        ref.upload_spectrum(h1, om1)
        ref.frame(0.7)
        want1 = ref.read_displacement()
        dh.from_host(h0.view(np.float32))
        do.from_host(om)
        ph.view(np.float32)[:] = h1.view(np.float32).ravel()
        po.view(np.float32)[:] = om1.ravel()
        # fused-only context, device memory; then the replacement from the registered host buffer BETWEEN two frames, no host wait:
        # the first frame keeps the old spectrum, the second sees the new one (stream order)
        fo = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
        fo.upload_spectrum_device(dh.ptr, do.ptr)
        fo.frame(0.7)
        assert np.array_equal(fo.read_displacement(), want0)
        first = DeviceBuffer(want0.nbytes)
        fo.bind_displacement(first.ptr)
        fo.frame(0.7)
        fo.upload_spectrum_device(ph.ptr, po.ptr)
        fo.bind_displacement(None)
        fo.frame(0.7)
        assert np.array_equal(fo.read_displacement(), want1)
        assert np.array_equal(first.to_host().reshape(n, n, 4), want0)
        first.free()
        # the same replacement on a CALLER stream while frames are queued on the context's own: the upload waits for them (they
        # keep the old spectrum), and a frame launched on the context stream afterwards waits for the upload (the
        # re-layout used to race with frames on another stream than its own)
        side = Stream()
        fo.upload_spectrum_device(dh.ptr, do.ptr)
        queued = DeviceBuffer(want0.nbytes)
        fo.bind_displacement(queued.ptr)
        for _ in range(40):                                        # ~1 ms of frames in front of the upload
            fo.frame(0.7)
        fo.upload_spectrum_device(ph.ptr, po.ptr, stream=side.handle)
        fo.bind_displacement(None)
        fo.frame(0.7)                                              # on the context stream, behind the upload on `side`
        assert np.array_equal(fo.read_displacement(), want1)
        assert np.array_equal(queued.to_host().reshape(n, n, 4), want0)
        queued.free()
        side.destroy()
        # full context: the staged path's natural copies are written as well
        full = g.OceanRenderer(n)
        full.device.upload_spectrum_device(dh.ptr, do.ptr)
        assert np.array_equal(full.device.read_spectrum(), h0)
        full.render(0.7)
        staged = full.displacement()
        full.render_fused(0.7)
        assert np.array_equal(full.displacement(), want0)
        assert_parity(staged[..., :3], want0[..., :3], 2e-5, "staged frame after a device-side upload")
        # a context of several tiles: tile 1 from device memory, tile 0 from the host
        tiles = g.OceanDevice(n, tiles=2)
        tiles.upload_spectrum(h1, om1, tile=0)
        tiles.upload_spectrum_device(dh.ptr, do.ptr, tile=1)
        tiles.frame_tiles(0.7)
        assert np.array_equal(tiles.read_batch_displacement(0), want1)
        assert np.array_equal(tiles.read_batch_displacement(1), want0)
        # pageable host memory, NULL, a tile that does not exist
        with pytest.raises(g.OceanError) as e:
            fo.upload_spectrum_device(h0.ctypes.data, om.ctypes.data)
        assert e.value.status == -1 and "registered" in str(e.value)            # OCEAN_E_INVALID_ARG
        with pytest.raises(g.OceanError):
            fo.upload_spectrum_device(0, do.ptr)
        with pytest.raises(g.OceanError):
            fo.upload_spectrum_device(dh.ptr, do.ptr, tile=1)
        fo.frame(0.7)                                              # the context still works, with the spectrum it had
        assert np.array_equal(fo.read_displacement(), want1)
    finally:
        for d in (ref, fo, tiles):
            if d is not None:
                d.destroy()
        if full is not None:
            full.dispose()
        for b in (dh, do, ph, po):
            b.free()


def test_linearity_and_impulse_1024():
    n = 1024
    om = g.synth.dispersion(n)
    r = g.OceanRenderer(n)
    try:
        # a single non-zero h0 texel: output is a closed-form pair of plane waves
        h0 = np.zeros((n, n), np.complex64)
        h0[700, 650] = 1.0 + 0.5j
        r.upload(h0, om)
        r.render_fused(4.0)
        out = r.displacement()
        assert_parity(out[..., :3], oc.frame_f64(h0, om, 4.0)[..., :3], TOL, "impulse")
        # linearity: frame(a + b) == frame(a) + frame(b)
        a, _ = g.synth.make_inputs(n, seed=1)
        b, _ = g.synth.make_inputs(n, seed=2)
        outs = []
        for h in (a, b, (a + b).astype(np.complex64)):
            r.upload(h, om)
            r.render_fused(0.5)
            outs.append(r.displacement().astype(np.float64))
        assert_parity((outs[0] + outs[1])[..., :3], outs[2][..., :3], 1e-5, "linearity")
    finally:
        r.dispose()


@pytest.mark.parametrize("n,channel", [(512, 0), (512, 1), (1024, 0), (2048, 0), (2048, 1), (8192, 1)])   # rows per workgroup: 1, 2, 4, 8
def test_normal_field(r512, ref_inputs, n, channel):
    """normals of the displacement map (shader/ocean.frag:50-66, quirk Q5 for channel 0) at the
    reference's size and at BASELINE config 3's (N = 2048: height + displacement + normal)."""
    if n == 512:
        h0, om = ref_inputs
        r = r512
    else:
        h0, om = g.synth.make_inputs(n)
        r = g.OceanRenderer(n)
        r.upload(h0, om)
    try:
        r.render_fused(2.0)
        rgba = r.displacement()
        got = r.device.normals(channel)
        ref = oc.normals_literal(rgba, channel)              # same fp32 map in, so the comparison is elementwise
        assert np.abs(got - ref).max() <= 1e-5
        # and end to end against the fp64 oracle of the whole path: normals are O(1), tolerance absolute;
        # d(normal)/d(field) ~ N/360, so a 1e-6 field error becomes ~1e-4 at 512 and ~5e-4 at 2048
        ref64 = oc.normals_f64(oc.frame_f64(h0, om, 2.0), channel)
        assert np.abs(got[..., :3] - ref64[..., :3]).max() <= (1e-3 if n == 512 else 4e-3) * max(1, n // 2048)
        assert np.allclose(np.linalg.norm(got[..., :3], axis=-1), 1.0, atol=1e-5)
        assert np.all(got[..., 3] == 0.0)
    finally:
        if n != 512:
            r.dispose()


@pytest.mark.parametrize("n,channel,fp16", [(256, 0, False), (512, 1, False), (1024, 2, False), (2048, 0, False), (2048, 1, False),
                                            (4096, 0, False), (8192, 0, True), (8192, 1, False)])
def test_frame_with_normal_field(n, channel, fp16):
    """BASELINE config 3 as ONE workload (ocean_set_frame_normals): pass 2 also stores the source channel as a dense plane and the
    normal-field kernel differentiates that plane.  The map is the frame's bit for bit, and the normals are ocean_normals'
    (k_normals on the RGBA map) bit for bit -- every rows-per-wave variant, both pass-2 kernels."""
    h0, om = g.synth.make_inputs(n, seed=21)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=fp16)
        d.frame(2.0)
        want_sum = d.checksum()
        want = d.normals(channel)                                # k_normals from the RGBA map
        d.set_frame_normals(channel)
        assert d.frame_normals == channel
        d.frame(2.0)
        assert d.checksum() == want_sum             # the PLANE instance writes the same map
        got = d.read_normals()
        assert np.array_equal(got, want)
        rgba = d.read_displacement()
        assert np.abs(got - oc.normals_literal(rgba, channel)).max() <= 1e-5
        d.frame(3.0)                                             # another frame, another field
        got3 = d.read_normals()
        d.set_frame_normals(None)
        assert d.frame_normals == -1
        assert np.array_equal(got3, d.normals(channel)) and not np.array_equal(got3, got)
        p1, p2, nrm, per = d.frame_times_ex(3)
        assert nrm is None and len(p1) == 3
    finally:
        d.destroy()


def test_frame_with_normal_field_staged_quirks_and_errors():
    """With non-reference quirks the frame is the staged dispatches, and the normal field comes from the map itself."""
    n = 512
    h0, om = g.synth.make_inputs(n, seed=22)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om)
        with pytest.raises(g.OceanError):
            d.set_frame_normals(3)
        d.set_quirks(0)
        d.set_frame_normals(1)
        d.frame(1.0)
        got = d.read_normals()
        assert np.array_equal(got, d.normals(1))
        d.set_quirks(g.QUIRKS_REFERENCE)
        p1, p2, nrm, per = d.frame_times_ex(4)
        assert len(nrm) == 4 and all(v > 0 for v in nrm)
    finally:
        d.destroy()


@pytest.mark.parametrize("n,count,own,fp16", [(256, 8, True, False), (512, 8, True, False), (512, 5, False, True), (1024, 4, False, False),
                                              (1024, 64, True, False), (2048, 3, False, False)])
def test_frame_batch_is_bit_identical_to_single_frames(n, count, own, fp16):
    """ocean_frame_batch: K time steps of one tile per launch pair at N <= 1024 (blockIdx.y = frame; K ordinary launch pairs
    above).  Every map equals ocean_frame's at t0 + dt * i bit for bit; library-owned and caller-owned maps; the context's own
    map is untouched."""
    from hipmem import DeviceBuffer
    h0, om = g.synth.make_inputs(n, seed=31)
    d = g.OceanDevice(n)
    buf = None
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=fp16)
        t0, dt = np.float32(0.75), np.float32(1.0 / 60.0)
        d.frame(9.0)
        keep = d.checksum()
        stride = n * n * 16 + (0 if own else 4096)
        if own:
            d.frame_batch(float(t0), float(dt), count)
            maps = [d.read_batch_displacement(i) for i in range(count)]
        else:
            buf = DeviceBuffer(stride * count)
            buf.fill(0xFF)
            d.frame_batch(float(t0), float(dt), count, out_ptr=buf.ptr, out_stride_bytes=stride)
            d.sync()
            raw = buf.to_host(np.uint8)
            maps = [raw[i * stride:i * stride + n * n * 16].view(np.float32).reshape(n, n, 4) for i in range(count)]
            assert np.all(raw[n * n * 16:stride] == 0xFF)          # the padding between two maps is nobody's
        assert d.checksum() == keep
        for i in sorted({0, 1, count // 2, count - 1}):
            ti = np.float32(t0 + np.float32(dt * np.float32(i)))
            d.frame(float(ti))
            assert np.array_equal(maps[i], d.read_displacement()), (n, i)
        assert not np.array_equal(maps[0], maps[1])
        ms = d.time_frame_batch(3, count)
        assert ms > 0.0
        with pytest.raises(g.OceanError):
            d.frame_batch(0.0, 0.1, 65)
        d.set_frame_normals(0)
        if n > 1024:                                              # above 1024 a batch is K ordinary frames: the field comes from ocean_frame
            with pytest.raises(g.OceanError):
                d.frame_batch(0.0, 0.1, 2)
        else:                                                     # the batch carries the normal field: K planes, one more launch
            k = min(count, 4)
            d.frame_batch(float(t0), float(dt), k)
            fields = [d.read_batch_normals(i) for i in range(k)]
            again = [d.read_batch_displacement(i) for i in range(k)]
            for i in range(k):
                ti = np.float32(t0 + np.float32(dt * np.float32(i)))
                d.frame(float(ti))
                assert np.array_equal(fields[i], d.read_normals()) and np.array_equal(again[i], d.read_displacement()), (n, i)
            assert not np.array_equal(fields[0], fields[1])
            # a batch of ONE frame with the normal field (pass 2 then wrote the single frame's plane while the field
            # was taken from the batch's -- stale from the batch above; another time, so that the stale plane is a wrong one)
            t1 = float(np.float32(t0 + np.float32(5.0)))
            d.frame_batch(t1, float(dt), 1)
            one_field, one_map = d.read_batch_normals(0), d.read_batch_displacement(0)
            d.frame(t1)
            assert np.array_equal(one_field, d.read_normals()) and np.array_equal(one_map, d.read_displacement()), n
            assert not np.array_equal(one_field, fields[0])
            # ... and what that batch did NOT leave behind is not handed out (the buffers still hold the k frames from before)
            for call in (lambda: d.read_batch_normals(1), lambda: d.read_batch_displacement(1)):
                with pytest.raises(g.OceanError) as e:
                    call()
                assert e.value.status == -5, e.value
            d.set_frame_normals(None)
            d.frame_batch(float(t0), float(dt), 2)
            with pytest.raises(g.OceanError) as e:
                d.read_batch_normals(0)                            # the last batch carried no field
            assert e.value.status == -5
            assert np.array_equal(d.read_batch_displacement(1), again[1])
    finally:
        if buf is not None:
            buf.free()
        d.destroy()


@pytest.mark.parametrize("n", [2048, 8192])
def test_fused_only_and_tile_rank_contexts(n):
    """ocean_context_create_ex: a fused-only context allocates 40 instead of 100 (N <= 4096) / 76 B/texel and computes the same
    frame bit for bit; the staged entry points refuse with OCEAN_E_STATE and a message; a tile-rank context holds the static
    inputs only (12 B/texel).  Footprints by hipMemGetInfo."""
    import hipmem
    h0, om = g.synth.make_inputs(n, seed=41)
    n2 = n * n
    slack = 64 << 20                                              # allocator granularity, twiddles, scratch
    before = hipmem.free_bytes()
    full = g.OceanDevice(n)
    used_full = before - hipmem.free_bytes()
    full.upload_spectrum(h0, om)
    full.frame(1.25)
    want = full.checksum()
    full.destroy()
    assert abs(hipmem.free_bytes() - before) <= slack             # everything came back
    fo = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
    try:
        used_fo = before - hipmem.free_bytes()
        per_texel_full = 100 if n <= 4096 else 76
        assert abs(used_full - per_texel_full * n2) <= slack, (used_full / n2)
        assert abs(used_fo - 40 * n2) <= slack, (used_fo / n2)
        fo.upload_spectrum(h0, om)
        assert abs((before - hipmem.free_bytes()) - 40 * n2) <= slack      # the upload's staging buffer is gone again
        fo.frame(1.25)
        assert fo.checksum() == want
        fo.upload_spectrum(h0, om, spectrum_fp16=True)
        fo.frame(1.25)
        assert fo.checksum() != want
        lib = g.load_library()
        assert lib.ocean_context_flags(fo._ctx) == g.CTX_FUSED_ONLY
        for call in (lambda: fo.read_field(g.FIELD_DY), lambda: fo.set_quirks(0), lambda: fo.profile_staged(0.0), lambda: fo.read_spectrum()):
            with pytest.raises(g.OceanError) as e:
                call()
            assert e.value.status == -5 and "OCEAN_CTX_FUSED_ONLY" in str(e.value)
        prop = g.Propagation.init(fo)                             # a stage object of the staged path
        with pytest.raises(g.OceanError) as e:
            prop.dispatch(g.PropagateLocals(0.0, n, 1000.0))
        assert e.value.status == -5
    finally:
        fo.destroy()
    lib = g.load_library()
    tr = g.OceanDevice(n, flags=g.CTX_TILE_RANK)
    try:
        assert abs((before - hipmem.free_bytes()) - 12 * n2) <= slack
        tr.upload_spectrum(h0, om)
        assert lib.ocean_context_flags(tr._ctx) == (g.CTX_FUSED_ONLY | g.CTX_TILE_RANK)
        with pytest.raises(g.OceanError) as e:
            tr.frame(0.0)
        assert e.value.status == -5 and "OCEAN_CTX_TILE_RANK" in str(e.value)
    finally:
        tr.destroy()


@pytest.mark.parametrize("n,tiles", [(256, 8), (512, 3), (1024, 2)])
def test_frame_tiles_is_bit_identical_to_one_context_per_tile(n, tiles):
    """ocean_frame_tiles: K independent tiles (their own spectra and dispersion arrays) per launch pair at N <= 1024.  Tile k's map
    equals ocean_frame's on a context that holds that tile alone, bit for bit; a frame is refused until every tile is uploaded."""
    inputs = [g.synth.make_inputs(n, seed=100 + k) for k in range(tiles)]
    d = g.OceanDevice(n, tiles=tiles)
    try:
        assert g.load_library().ocean_context_tiles(d._ctx) == tiles
        for k, (h0, om) in enumerate(inputs[:-1]):
            d.upload_spectrum(h0, om, tile=k)
        with pytest.raises(g.OceanError):
            d.frame_tiles(1.0)                                    # the last tile has no inputs yet
        d.upload_spectrum(*inputs[-1], tile=tiles - 1)
        d.frame_tiles(2.25)
        maps = [d.read_batch_displacement(k) for k in range(tiles)]
        d.set_frame_normals(1)                                    # ... and with the normal field: one field per tile
        d.frame_tiles(2.25)
        tile_normals = [d.read_batch_normals(k) for k in range(tiles)]
        assert all(np.array_equal(d.read_batch_displacement(k), maps[k]) for k in range(tiles))
        d.set_frame_normals(None)
        for t_ in (0.5, 2.25):                                    # consecutive launches reuse the intermediates
            d.frame_tiles(t_)
        again = [d.read_batch_displacement(k) for k in range(tiles)]
        assert d.time_frame_batch(3, tiles) > 0
        with pytest.raises(g.OceanError):
            d.upload_spectrum(*inputs[0], spectrum_fp16=True)
    finally:
        d.destroy()
    for k, (h0, om) in enumerate(inputs):
        one = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
        try:
            one.upload_spectrum(h0, om)
            one.frame(2.25)
            want = one.read_displacement()
            want_normals = one.normals(1)
        finally:
            one.destroy()
        assert np.array_equal(maps[k], want) and np.array_equal(again[k], want), (n, k)
        assert np.array_equal(tile_normals[k], want_normals), (n, k)
    assert not np.array_equal(maps[0], maps[1])
    with pytest.raises(g.OceanError):
        g.OceanDevice(2048, tiles=2)                              # above 1024 one tile fills the chip: one context per tile
    # a context of ONE tile (which ocean_context_create_tiles allows) with the normal field: a batch of one frame
    d = g.OceanDevice(n, tiles=1, tiles_context=True)
    try:
        d.upload_spectrum(*inputs[0], tile=0)
        d.set_frame_normals(1)
        d.frame_tiles(0.125)                                      # leaves a plane of another time behind
        d.frame_tiles(2.25)
        assert np.array_equal(d.read_batch_normals(0), tile_normals[0]) and np.array_equal(d.read_batch_displacement(0), maps[0]), n
        with pytest.raises(g.OceanError):
            d.read_batch_normals(1)
    finally:
        d.destroy()


@pytest.mark.parametrize("verts,offset", [(128, (0.0, 0.0)), (128, (127.0, 127.0)), (257, (0.0, 0.0))])
def test_vertex_positions(r512, ref_inputs, verts, offset):
    """the vertex stage's positions (shader/ocean.vert:21-25; patch grid and offsets of
    src/render.rs:494-551) from the current displacement map."""
    r512.render_fused(2.0)
    rgba = r512.displacement()
    got = r512.device.positions(verts, offset)
    ref = oc.positions_f64(rgba, verts, offset)          # same fp32 map in
    assert np.abs(got - ref).max() <= 1e-4
    ref64 = oc.positions_f64(oc.frame_f64(*ref_inputs, 2.0), verts, offset)
    assert np.abs(got - ref64).max() <= 1e-3             # absolute, on coordinates up to ~260
    assert np.all(got[..., 3] == 1.0)


@pytest.mark.parametrize("n", [512, 8192])
def test_config5_fp16_spectrum(n, ref_inputs):
    """BASELINE config 5: fp16 spectrum / fp32 accumulate (N = 8192 is the configured size; 512 uses the
    reference data).  Parity against the oracle fed the same quantised inputs."""
    if n == 512:
        h0, om = ref_inputs
    else:
        h0, om = g.synth.make_inputs(n)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=True)
        s = d.spectrum_scale_log2
        assert 2 ** 14 <= np.abs(h0.view(np.float32)).max() * 2.0 ** s < 2 ** 15
        deq = d.read_spectrum()
        q = (h0.view(np.float32) * np.float32(2.0 ** s)).astype(np.float16).astype(np.float32) * np.float32(2.0 ** -s)
        assert np.array_equal(deq.view(np.float32), q.reshape(deq.view(np.float32).shape))   # RNE fp16, power-of-two scale
        t = 1.25
        d.frame(t)
        out = d.read_displacement()
        if n == 512:
            assert_parity(out[..., :3], oc.frame_f64(deq, om, t)[..., :3], TOL, "fp16 spectrum vs oracle(quantised)")
            nmax, rl2 = oc.parity_errors(out[..., :3], oc.frame_f64(h0, om, t)[..., :3])
            assert rl2.max() > 1e-5              # the quantisation is real: not the fp32 result
        else:
            # full size, EVERY texel: the C restatement of the four shaders on the host cores, fed the same
            # dequantised spectrum the kernels use (config 5 parity is judged on the quantised inputs)
            cc.set_threads(min(32, cc.max_threads()))
            refc = cc.FrameRunner(deq, om).frame(t)
            nmax, rl2 = assert_parity(out[..., :3], refc[..., :3], TOL, f"N={n} fp16 spectrum vs C oracle(quantised)")
            assert nmax.max() < 2e-5                      # two fp32 paths on identical inputs: a few 1e-6
            assert np.all(out[..., 3] == 0.0)
            # and sampled texels by direct fp64 summation of the quantised spectrum (independent of any FFT)
            H, DX, DZ = oc.propagate_f64(deq, om, t)
            k = np.arange(n)
            scale = np.abs(out[..., :3]).max((0, 1))
            for (x, y) in [(0, 0), (n // 2 + 3, n // 3), (n - 1, n - 1)]:
                ey, ex = np.exp(2j * np.pi * k * y / n), np.exp(2j * np.pi * k * x / n)
                sgn = -1.0 if (x + y) % 2 == 0 else 1.0
                ref = np.array([(ey @ (F @ ex)).real for F in (DX, H, DZ)]) * sgn
                assert np.all(np.abs(out[y, x, :3] - ref) <= TOL * scale), (x, y, out[y, x, :3], ref)
    finally:
        d.destroy()


@pytest.mark.parametrize("quirks", [0, 1, 2])
def test_quirk_switches(ref_inputs, quirks):
    """Q1/Q2 off through the C ABI; every entry point honours the setting (ocean_frame then runs the
    staged kernels), and the reference setting is restored bit for bit."""
    h0, om = ref_inputs
    r = g.OceanRenderer(512)
    try:
        r.upload(h0, om)
        r.render_fused(3.0)
        ref_out = r.displacement()
        r.device.set_quirks(quirks)
        assert r.device.quirks == quirks
        want = oc.frame_f64(h0, om, 3.0, quirks=quirks)[..., :3]
        r.render(3.0)
        staged = r.displacement()
        assert_parity(staged[..., :3], want, TOL, f"staged quirks={quirks}")
        r.render_fused(3.0)
        assert np.array_equal(r.displacement(), staged)
        assert oc.parity_errors(staged[..., :3], ref_out[..., :3])[0].max() > 1e-2
        with pytest.raises(g.OceanError):
            r.device.profile_frame(0.0)              # the fused kernels are reference-only
        with pytest.raises(g.OceanError):
            r.device.set_quirks(4)
        r.device.set_quirks(g.QUIRKS_REFERENCE)
        r.render_fused(3.0)
        assert np.array_equal(r.displacement(), ref_out)
    finally:
        r.dispose()


@pytest.mark.parametrize("n,normals", [(512, False), (2048, True), (4096, False)])
def test_frames_are_capturable_in_a_hip_graph(n, normals):
    """`ocean_frame` / `ocean_frame_batch` on a caller stream launch kernels and nothing else, so a consumer can record them into a
    hipGraph of its own (stream capture): K frames at K times into K maps, replayed twice, every map bit-identical to the
    directly launched frame -- also with the normal field (three kernels per frame) and through the LDS-DMA loader.  (A graph of
    the frame's launches is no faster than the launches -- DESIGN 4.3 -- this is about being embeddable.)"""
    from hipmem import CapturedGraph, DeviceBuffer, Stream
    h0, om = g.synth.make_inputs(n, seed=61)
    k = 4
    d = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
    maps = DeviceBuffer(k * n * n * 16)
    side = Stream()
    graph = None
    try:
        d.upload_spectrum(h0, om)
        if normals:
            d.set_frame_normals(0)
        times = [0.5 + 0.25 * i for i in range(k)]

        def record(stream):
            for i, t in enumerate(times):
                d.bind_displacement(maps.ptr + i * n * n * 16)
                d.frame(t, stream=stream)
            d.bind_displacement(None)

        graph = CapturedGraph(side, record)
        assert graph.nodes == k * (3 if normals else 2), graph.nodes   # kernel nodes only: no copies, no host nodes
        maps.fill(0xFF)                                                # capture launched nothing
        for _ in range(2):
            graph.launch()
        side.synchronize()
        got = maps.to_host().reshape(k, n, n, 4)
        last_normals = d.read_normals() if normals else None
        for i, t in enumerate(times):
            d.frame(t)
            assert np.array_equal(got[i], d.read_displacement()), (n, i)
        if normals:
            assert np.array_equal(last_normals, d.read_normals())      # the graph's last frame = the last direct frame
    finally:
        if graph is not None:
            graph.destroy()
        side.destroy()
        maps.free()
        d.destroy()


def test_two_contexts_share_the_gpu():
    """Two contexts (two tiles) on one device, their frames interleaved on their own streams with no host wait in between: each
    map equals the one its context computes alone (the reference keeps 3 frames in flight on one queue, src/lib.rs:86,150; a
    consumer with several tiles keeps several contexts)."""
    n = 1024
    ins = [g.synth.make_inputs(n, seed=s) for s in (71, 72)]
    alone = []
    for h0, om in ins:
        d = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
        try:
            d.upload_spectrum(h0, om)
            d.frame(3.5)
            alone.append(d.checksum())
        finally:
            d.destroy()
    a, b = (g.OceanDevice(n, flags=g.CTX_FUSED_ONLY) for _ in range(2))
    try:
        a.upload_spectrum(*ins[0])
        b.upload_spectrum(*ins[1])
        for i in range(50):
            a.frame(0.1 * i)
            b.frame(0.2 * i)
        a.frame(3.5)
        b.frame(3.5)
        assert [a.checksum(), b.checksum()] == alone and alone[0] != alone[1]
    finally:
        a.destroy()
        b.destroy()


def test_time_is_stateless(r512, ref_inputs):
    """No state but `time` (SURVEY 5 checkpoint/resume): frames are reproducible in any order."""
    r512.render_fused(5.0)
    a = r512.displacement()
    r512.render_fused(123.0)
    r512.render_fused(5.0)
    assert np.array_equal(a, r512.displacement())


def test_error_behaviour():
    with pytest.raises(g.OceanError) as e:
        g.OceanDevice(500)
    assert e.value.status == -2                      # OCEAN_E_UNSUPPORTED_N
    with pytest.raises(g.OceanError):
        g.OceanDevice(512, device_ordinal=99)
    d = g.OceanDevice(256)
    try:
        with pytest.raises(g.OceanError) as e:
            d.frame(0.0)                             # before upload
        assert e.value.status == -5
        with pytest.raises(g.OceanError):
            d.upload_spectrum(np.zeros((128, 128), np.complex64), np.zeros((128, 128), np.float32))
        p = g.Propagation.init(d)
        d.upload_spectrum(np.zeros((256, 256), np.complex64), np.zeros((256, 256), np.float32))
        with pytest.raises(g.OceanError):
            p.dispatch(g.PropagateLocals(0.0, 512))  # resolution mismatch
        p.destroy()
        with pytest.raises(g.OceanError):
            p.dispatch(g.PropagateLocals(0.0, 256))  # use after destroy
    finally:
        d.destroy()


_TORCH_INTEROP = r"""
import sys, numpy as np
import torch                      # FIRST: libocean_hip.so then binds to the HIP runtime torch loaded
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import gfx_ocean_amd as g
from oracle import ocean_oracle as oc
h0, om = oc.load_reference_inputs(sys.argv[1] + "/tests/golden/spectrum.bin", sys.argv[1] + "/tests/golden/omega.bin")
n = 512
d = g.OceanDevice(n)
d.upload_spectrum(h0, om)
out = torch.zeros((n, n, 4), dtype=torch.float32, device="cuda:0")
d.bind_displacement(out.data_ptr())
s = torch.cuda.Stream()
d.frame(1.0, stream=s.cuda_stream)
s.synchronize()
nmax, rl2 = oc.parity_errors(out.cpu().numpy()[..., :3], oc.frame_f64(h0, om, 1.0)[..., :3])
assert nmax.max() <= 1e-4 and rl2.max() <= 1e-4, (nmax, rl2)
d.bind_displacement(None)
d.destroy()
print("TORCH_INTEROP_OK", nmax.max())
"""


def test_torch_interop_stream_and_bound_output():
    """The C ABI takes raw device pointers / hipStream_t: drive it from torch memory and a torch
    stream.  Own process, torch imported first (one HIP runtime per process: torch bundles its own
    libamdhip64 with the same soname, so whichever loads first serves both)."""
    import subprocess
    import sys
    import importlib.util
    if importlib.util.find_spec("torch") is None:      # (not importorskip: importing torch HERE would map a second HIP runtime)
        pytest.skip("torch not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _TORCH_INTEROP, root], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "TORCH_INTEROP_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("n", [512, 2048])
def test_packed_displacement_for_the_gather(n, ref_inputs):
    """SURVEY 8e / VERDICT r02 #8: ocean_pack_displacement -- the map without its always-zero alpha (12 B/texel) or the
    height alone (4), bit for bit the channels of the RGBA map, written to caller device memory."""
    from hipmem import DeviceBuffer
    h0, om = ref_inputs if n == 512 else g.synth.make_inputs(n)
    d = g.OceanDevice(n)
    buf = DeviceBuffer(n * n * 16)
    try:
        d.upload_spectrum(h0, om)
        d.frame(1.5)
        rgba = d.read_displacement()
        assert d.packed_bytes(g.PACK_RGBA32F) == n * n * 16 and d.packed_bytes(g.PACK_RGB32F) == n * n * 12
        assert d.packed_bytes(g.PACK_HEIGHT32F) == n * n * 4
        d.pack_displacement(g.PACK_RGB32F, buf.ptr)
        d.sync()
        assert np.array_equal(buf.to_host(np.float32, n * n * 12).reshape(n, n, 3), rgba[..., :3])
        d.pack_displacement(g.PACK_HEIGHT32F, buf.ptr)
        d.sync()
        assert np.array_equal(buf.to_host(np.float32, n * n * 4).reshape(n, n), rgba[..., 1])
        d.pack_displacement(g.PACK_RGBA32F, buf.ptr)
        d.sync()
        assert np.array_equal(buf.to_host(np.float32).reshape(n, n, 4), rgba)
        with pytest.raises(g.OceanError):
            d.pack_displacement(7, buf.ptr)
        with pytest.raises(g.OceanError):
            d.pack_displacement(g.PACK_RGB32F, buf.ptr + 4)          # misaligned
    finally:
        buf.free()
        d.destroy()


def test_frame_time_distributions():
    """SURVEY 8d's "median + p10/p90" (VERDICT r03 #2): ocean_time_frame_batches (plain loop, one stream event per batch) and
    ocean_frame_times (per-dispatch events) agree with ocean_time_frames on what a frame costs, and reject bad arguments."""
    n = 2048
    d = g.OceanDevice(n)
    try:
        h0, om = g.synth.make_inputs(n, seed=2)
        with pytest.raises(g.OceanError):
            d.time_frame_batches(4, 5)                               # before upload: OCEAN_E_STATE
        d.upload_spectrum(h0, om)
        d.time_frames(200)                                            # clock ramp
        total = d.time_frames(200) / 200
        batches = d.time_frame_batches(20, 10)
        assert len(batches) == 20 and all(b > 0 for b in batches)
        per_frame = sorted(b / 10 for b in batches)
        assert 0.8 * total < per_frame[10] < 1.25 * total, (total, per_frame)
        p1, p2, period = d.frame_times(100)
        assert len(p1) == len(p2) == len(period) == 100 and min(p1) > 0 and min(p2) > 0
        med = lambda v: sorted(v)[len(v) // 2]
        assert med(p1) + med(p2) < 1.3 * total and med(period) >= 0.95 * (med(p1) + med(p2))
        with pytest.raises(g.OceanError):
            d.frame_times(0)
        with pytest.raises(g.OceanError):
            d.time_frame_batches(0, 10)
        d.frame(1.0)                                                  # the measurement loops leave the context usable
        assert np.isfinite(d.read_displacement()).all()
    finally:
        d.destroy()


def test_stale_stage_handle_of_a_reused_context_address_is_rejected():
    """a stage handle kept after its context is destroyed must not become valid again when a new context
    happens to be allocated at the same address (handles carry the generation of their context)."""
    seen = False
    for _ in range(64):
        if seen:
            break
        d = g.OceanDevice(256)
        p = g.Propagation.init(d)
        addr = d._ctx.value
        raw = p._h
        d.destroy()
        d2 = g.OceanDevice(256)
        try:
            d2.upload_spectrum(np.zeros((256, 256), np.complex64), np.zeros((256, 256), np.float32))
            loc = g.PropagateLocals(0.0, 256)._c()
            import ctypes
            assert g.load_library().ocean_propagate(raw, ctypes.byref(loc), None) == -1     # OCEAN_E_INVALID_ARG
            seen = seen or (d2._ctx.value == addr)
        finally:
            d2.destroy()
            g.load_library().ocean_propagation_destroy(raw)
            p._h = None
    # Address reuse is up to the allocator.  Without it the stale handle was rejected by the dead-context check alone and
    # the generation comparison was never what decided: say so instead of passing.
    if not seen:
        pytest.skip("the allocator never handed a new context the old address in 64 tries: generation check not exercised")


@pytest.mark.parametrize("f16", [True, False])
def test_config5_bfp16_intermediate_8192(f16):
    """SURVEY 8d "B_frame16" as an opt-in precision mode (ocean_set_intermediate(OCEAN_INTER_BFP16), N = 8192): int16
    intermediate with one power-of-two scale per 64 x 2 block.  EVERY texel against the C restatement of the shaders fed
    the spectrum the kernels use: within the 1e-4 tolerance (expected 3-4e-5: tools/inter16_numerics.py), really
    different from the fp32 intermediate, and switchable back bit for bit."""
    n, t = 8192, 1.25
    h0, om = g.synth.make_inputs(n)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=f16)
        src = d.read_spectrum() if f16 else h0
        d.frame(t)
        c32 = d.checksum()
        ref32 = d.read_displacement()
        assert d.intermediate == g.INTER_F32
        d.set_intermediate(g.INTER_BFP16)
        assert d.intermediate == g.INTER_BFP16
        d.frame(t)
        out = d.read_displacement()
        cc.set_threads(min(32, cc.max_threads()))
        refc = cc.FrameRunner(src, om).frame(t)
        nmax, rl2 = assert_parity(out[..., :3], refc[..., :3], TOL, "bfp16 intermediate vs C oracle")
        assert nmax.max() < 6e-5 and np.all(out[..., 3] == 0.0)
        q = oc.parity_errors(out[..., :3], ref32[..., :3])[0].max()
        assert 5e-6 < q < 6e-5                                       # the quantisation is real, and small
        d.frame(t)
        assert np.array_equal(out, d.read_displacement())             # reproducible
        d.set_intermediate(g.INTER_F32)
        d.frame(t)
        assert d.checksum() == c32                                     # and the default comes back bit for bit
    finally:
        d.destroy()


def test_bfp16_intermediate_is_8192_only():
    d = g.OceanDevice(4096)
    try:
        with pytest.raises(g.OceanError) as e:
            d.set_intermediate(g.INTER_BFP16)
        assert e.value.status == -2                                   # OCEAN_E_UNSUPPORTED_N
        with pytest.raises(g.OceanError):
            d.set_intermediate(7)
        assert d.intermediate == g.INTER_F32
    finally:
        d.destroy()
