"""The oracle against everything that pins it: the two independent formulations,
the survey KAT, the committed golden vectors, the C restatement, and structural properties."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity
from oracle import c_oracle as cc
from oracle import ocean_oracle as oc


@pytest.mark.parametrize("t", [0.0, 1.0, 10.0, 100.0, 1000.0])
def test_literal_vs_f64_on_reference_data(ref_inputs, t):
    h0, om = ref_inputs
    lit = oc.frame_literal(h0, om, t)
    f64 = oc.frame_f64(h0, om, t)
    nmax, rl2 = oc.parity_errors(lit[..., :3], f64[..., :3])
    assert np.all(nmax <= 2e-6) and np.all(rl2 <= 2e-6), (nmax, rl2)   # SURVEY 7 step 1 gate
    assert np.all(lit[..., 3] == 0.0)


def test_survey_kat(ref_inputs):
    h0, om = ref_inputs
    with open(os.path.join(GOLDEN, "kat_survey.json")) as f:
        kat = json.load(f)
    for t in ("0", "1", "10"):
        out = oc.frame_f64(h0, om, float(t))
        for x, y, dx, h, dz in kat[t]:
            assert np.allclose(out[y, x, :3], [dx, h, dz], atol=2e-6), (t, x, y, out[y, x, :3])
        agg = kat["aggregates"][t]
        ch = out[..., :3]
        assert np.allclose(ch.sum((0, 1)), agg["sum"], atol=2e-3)
        assert np.allclose(np.sqrt((ch ** 2).sum((0, 1))), agg["l2"], rtol=2e-6)
        if "max" in agg:
            assert np.allclose(np.abs(ch).max((0, 1)), agg["max"], atol=2e-6)


@pytest.mark.parametrize("name,n,t", [("frame512_t0", 512, 0.0), ("frame512_t1", 512, 1.0),
                                     ("frame512_t10", 512, 10.0), ("frame256_t1", 256, 1.0)])
def test_golden_fixtures(ref_inputs, name, n, t):
    h0, om = ref_inputs
    if n == 256:
        h0, om = oc.centre_crop(h0, 256), oc.centre_crop(om, 256)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    lit = oc.frame_literal(h0, om, t)[..., :3]
    assert np.abs(lit[:64, :64] - g["crop"]).max() <= 1e-4 * g["max"].max()
    for (x, y), v in zip(g["probes_xy"], g["probes"]):
        assert np.allclose(lit[y, x], v, atol=3e-5)
    assert np.allclose(np.sqrt((lit.astype(np.float64) ** 2).sum((0, 1))), g["l2"], rtol=1e-5)


def test_c_oracle_matches_numpy(ref_inputs_256):
    h0, om = ref_inputs_256
    t = 7.5
    h, dx, dz = oc.propagate_literal(h0, om, t)
    ch, cdx, cdz = cc.propagate(h0, om, t)
    for a, b in ((ch, h), (cdx, dx), (cdz, dz)):
        assert oc.parity_errors(a, b)[0].max() <= 5e-7
    assert oc.parity_errors(cc.fft_rows(h), oc.fft_rows_literal(h))[0].max() <= 1e-6
    assert oc.parity_errors(cc.fft_cols(h), oc.fft_cols_literal(h))[0].max() <= 1e-6
    r = cc.FrameRunner(h0, om)
    assert_parity(r.frame(t)[..., :3], oc.frame_f64(h0, om, t)[..., :3], 2e-6, "C oracle frame")
    assert np.array_equal(cc.correct(h, dx, dz), oc.correction_literal(h, dx, dz))


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192])
def test_literal_stockham_is_unnormalised_inverse_dft(n):
    """Q4: 512->N, 256->N/2, 9->log2 N reproduces N*ifft for every supported N."""
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    got = oc.fft_rows_literal(x)
    ref = np.fft.ifft(x.astype(np.complex128), axis=1) * n
    assert oc.parity_errors(got, ref)[0].max() <= 2e-6


def test_fft_properties():
    n = 256
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
    b = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
    fa, fb, fab = (oc.fft_rows_literal(v) for v in (a, b, (a + 2 * b).astype(np.complex64)))
    assert oc.parity_errors(fab, fa.astype(np.complex128) + 2 * fb)[0].max() <= 2e-6          # linearity
    imp = np.zeros((1, n), np.complex64)
    imp[0, 5] = 1
    k = np.arange(n)
    assert np.abs(oc.fft_rows_literal(imp)[0] - np.exp(2j * np.pi * 5 * k / n)).max() <= 2e-6   # impulse
    assert abs((np.abs(fa) ** 2).sum() / (n * (np.abs(a) ** 2).sum()) - 1) <= 1e-5            # Parseval
    rc = oc.fft_cols_literal(oc.fft_rows_literal(a))
    cr = oc.fft_rows_literal(oc.fft_cols_literal(a))
    assert oc.parity_errors(rc, cr)[0].max() <= 2e-6                                          # separability


def test_q1_uint_wrap_table():
    """at N=512 float(uint(2g-513)) rounds to multiples of 256 for g <= 256."""
    xf = oc.wave_vector_q1(512)
    assert np.all(xf[0:65] == 4294966784.0)
    assert np.all(xf[65:193] == 4294967040.0)
    assert np.all(xf[193:257] == 4294967296.0)
    assert np.array_equal(xf[257:], (2.0 * np.arange(257, 512) - 513.0).astype(np.float32))
    h0 = np.ones((512, 512), np.complex64)
    om = np.zeros((512, 512), np.float32)
    h, dx, dz = oc.propagate_literal(h0, om, 0.0)
    kx = -dx.imag / 2.0                # h = (2, 0) -> dx = (kn*h.y, -kn*h.x) = (0, -2 kn)
    kz = -dz.imag / 2.0
    assert np.allclose(kx[0, 0], 0.70711, atol=1e-5) and np.allclose(kz[0, 0], 0.70711, atol=1e-5)   # both wrap
    assert np.allclose(kx[400, 10], 1.0, atol=1e-6) and abs(kz[400, 10]) < 1e-6                      # only x wraps
    assert abs(kx[10, 400]) < 1e-6 and np.allclose(kz[10, 400], 1.0, atol=1e-6)                      # only y wraps
    assert np.allclose(np.hypot(kx[300:, 300:], kz[300:, 300:]), 1.0, atol=1e-6)                    # true quadrant


def test_centre_crop_is_self_consistent(ref_inputs):
    """mirror partner and k of a cropped index match the 512 formulas."""
    h0, om = ref_inputs
    h256 = oc.centre_crop(h0, 256)
    assert h256.shape == (256, 256)
    assert h256[255 - 10, 255 - 20] == h0[511 - (128 + 10), 511 - (128 + 20)]
    g = np.arange(512)
    assert np.array_equal((2 * g[128:384] - 513), 2 * np.arange(256) - 257)


def test_bincode_header(ref_inputs):
    h0, om = ref_inputs
    assert h0.shape == (512, 512) and om.shape == (512, 512)
    with open(os.path.join(GOLDEN, "omega.bin"), "rb") as f:
        assert int.from_bytes(f.read(8), "little") == 262144


@pytest.mark.parametrize("t", [0, 1, 10])
def test_literal_oracle_reproduces_the_shipped_spirv_bit_for_bit(ref_inputs, t):
    """tests/golden/spirv_frame512_t*.npz come from executing the reference's own
    shader/spv/*.comp.spv (oracle/spirv_interp.py, generator tests/golden/make_spirv_golden.py).
    The literal restatement must match them bit for bit, stage by stage."""
    import zlib
    h0, om = ref_inputs
    g = np.load(os.path.join(GOLDEN, f"spirv_frame512_t{t}.npz"))
    img, st = oc.frame_literal(h0, om, float(t), return_stages=True)
    assert np.array_equal(img[:64, :64].view(np.uint32), g["crop"].view(np.uint32))
    assert np.array_equal(img[::8, ::8].view(np.uint32), g["sub8"].view(np.uint32))
    assert zlib.crc32(img.tobytes()) == int(g["crc_image"])
    for key, name in (("propagate", "crc_propagate"), ("rows", "crc_fft_row"), ("cols", "crc_fft_col")):
        for buf, crc in zip(st[key], g[name]):      # order: height, disp_x, disp_z
            raw = np.ascontiguousarray(buf, np.complex64).view(np.float32).tobytes()
            assert zlib.crc32(raw) == int(crc), (key, t)


def test_spirv_interpreter_against_reference_binaries_when_present(ref_inputs):
    """Build container only: re-run the interpreter on /root/reference's binaries."""
    spv = "/root/reference/shader/spv"
    if not os.path.isdir(spv):
        pytest.skip("reference tree not present (GPU box)")
    from oracle import spirv_interp as si
    h0, om = ref_inputs
    img = si.run_reference_frame(spv, h0, om, 2.5)
    assert np.array_equal(img.view(np.uint32), oc.frame_literal(h0, om, 2.5).view(np.uint32))
