"""Host-side logic that needs no GPU: the module surface mirroring src/ocean.rs / src/fft.rs,
uniform-block layouts, bincode I/O, the synthetic generator, error behaviour without a device."""
import ctypes
import os

import numpy as np
import pytest

import gfx_ocean_amd as g
from gfx_ocean_amd import _lib
from conftest import GOLDEN


def test_module_surface_mirrors_reference():
    # src/ocean.rs: PropagateLocals, Propagation, CorrectionLocals, Correction; src/fft.rs: Fft
    for name in ("PropagateLocals", "Propagation", "CorrectionLocals", "Correction", "Fft"):
        assert hasattr(g, name)
    for cls in (g.Propagation, g.Correction, g.Fft):
        assert callable(getattr(cls, "init")) and callable(getattr(cls, "destroy"))
    assert g.RESOLUTION == 512 and g.DOMAIN_SIZE == 1000.0      # src/render.rs:44,46
    assert (g.FIELD_DX, g.FIELD_DY, g.FIELD_DZ) == (0, 1, 2)     # desc_sets order, src/render.rs:971-988


def test_uniform_block_layouts():
    # shader/propagate.comp:16-20 std140 offsets 0/4/8; shader/correction.comp:6-8
    P, C = _lib.PropagateLocalsC, _lib.CorrectionLocalsC
    assert ctypes.sizeof(P) == 12 and (P.time.offset, P.resolution.offset, P.domain_size.offset) == (0, 4, 8)
    assert ctypes.sizeof(C) == 4
    loc = g.PropagateLocals(1.5, 512)._c()
    assert (loc.time, loc.resolution, loc.domain_size) == (1.5, 512, 1000.0)


def test_bincode_roundtrip(tmp_path):
    h0, om = g.bincode.load_spectrum(os.path.join(GOLDEN, "spectrum.bin"), os.path.join(GOLDEN, "omega.bin"))
    assert h0.shape == (512, 512) and h0.dtype == np.complex64 and om.dtype == np.float32
    sp, op = str(tmp_path / "s.bin"), str(tmp_path / "o.bin")
    g.bincode.save_spectrum(sp, op, h0, om)
    with open(sp, "rb") as a, open(os.path.join(GOLDEN, "spectrum.bin"), "rb") as b:
        assert a.read() == b.read()                      # byte-identical to the reference's file
    with open(op, "rb") as a, open(os.path.join(GOLDEN, "omega.bin"), "rb") as b:
        assert a.read() == b.read()
    with open(sp, "ab") as f:
        f.write(b"\0")
    with pytest.raises(ValueError):
        g.bincode.read_vec_f32(sp, 2)


def test_synth_generator(ref_inputs):
    h0, om = g.synth.make_inputs(512)
    h0b, _ = g.synth.make_inputs(512)
    assert np.array_equal(h0, h0b)                                   # deterministic
    assert not np.array_equal(h0, g.synth.make_inputs(512, seed=513)[0])
    rh0, rom = ref_inputs
    assert np.abs(om - rom).max() < 1e-4                             # fitted dispersion
    assert np.argwhere(h0 == 0).tolist() == np.argwhere(rh0 == 0).tolist()
    assert 0.5 < np.sqrt((np.abs(h0) ** 2).mean()) / np.sqrt((np.abs(rh0) ** 2).mean()) < 2.0


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly, not fall back: no GPU here -> OceanError from create."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    if not os.path.exists(g.library_path()):
        with pytest.raises(g.OceanError):
            g.load_library()
    else:
        with pytest.raises(g.OceanError) as e:
            g.OceanDevice(512)
        assert e.value.status in (_lib.STATUS_NAMES.keys())


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gfx_ocean_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".h", ".cpp", ".rs")):
                with open(os.path.join(root, fn)) as f:
                    src = f.read()
                assert "import oracle" not in src and "from oracle" not in src and "libocean_oracle" not in src, fn
