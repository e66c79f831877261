"""gfx_ocean_amd -- MI355X-native ocean height-field path (propagate -> 2-D iFFT -> correction).

Host-side mirror of the reference's private ``ocean`` / ``fft`` modules
(src/ocean.rs, src/fft.rs) and of the compute slice of ``Renderer`` (src/render.rs),
over the C ABI in ``include/ocean_hip.h`` (HIP kernels in ``csrc/``).

There is no CPU fallback: importing works anywhere, but creating an
``OceanDevice`` raises ``OceanError`` if ``libocean_hip.so`` is missing or no
gfx950 device is present.
"""
from ._lib import OceanError, build_library, library_path, load_library  # noqa: F401
from .ocean import (Correction, CorrectionLocals, Propagation, PropagateLocals,  # noqa: F401
                    DOMAIN_SIZE, RESOLUTION)
from .fft import Fft  # noqa: F401
from .render import (OceanDevice, OceanRenderer, FIELD_DX, FIELD_DY, FIELD_DZ, FIELD_ALL,  # noqa: F401
                     QUIRK_Q1, QUIRK_Q2, QUIRKS_REFERENCE, PACK_RGBA32F, PACK_RGB32F, PACK_HEIGHT32F,
                     PACK_BYTES_PER_TEXEL, INTER_F32, INTER_BFP16, CTX_FUSED_ONLY, CTX_TILE_RANK, CTX_TILE_BANDS)
from . import bincode, synth  # noqa: F401

__all__ = [
    "OceanError", "build_library", "library_path", "load_library",
    "Correction", "CorrectionLocals", "Propagation", "PropagateLocals", "Fft",
    "OceanDevice", "OceanRenderer", "FIELD_DX", "FIELD_DY", "FIELD_DZ", "FIELD_ALL",
    "QUIRK_Q1", "QUIRK_Q2", "QUIRKS_REFERENCE", "PACK_RGBA32F", "PACK_RGB32F", "PACK_HEIGHT32F", "PACK_BYTES_PER_TEXEL", "INTER_F32", "INTER_BFP16", "CTX_FUSED_ONLY", "CTX_TILE_RANK", "CTX_TILE_BANDS",
    "DOMAIN_SIZE", "RESOLUTION", "bincode", "synth",
]
