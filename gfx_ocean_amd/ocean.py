"""Mirror of the reference's `mod ocean` (src/ocean.rs): PropagateLocals, Propagation,
CorrectionLocals, Correction -- same names, same init/destroy life cycle, over HIP."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

from ._lib import CorrectionLocalsC, OceanError, PropagateLocalsC, load_library

RESOLUTION = 512      # src/render.rs:44
DOMAIN_SIZE = 1000.0  # src/render.rs:46


@dataclass
class PropagateLocals:
    """src/ocean.rs:8-13 / shader/propagate.comp:16-20."""
    time: float
    resolution: int
    domain_size: float = DOMAIN_SIZE

    def _c(self):
        return PropagateLocalsC(float(self.time), int(self.resolution), float(self.domain_size))


@dataclass
class CorrectionLocals:
    """src/ocean.rs:179-182 / shader/correction.comp:6-8."""
    resolution: int

    def _c(self):
        return CorrectionLocalsC(int(self.resolution))


class _Stage:
    _init = _destroy = None

    def __init__(self, device, handle):
        self.device = device
        self._h = handle

    @classmethod
    def init(cls, device):
        """`unsafe fn init(device) -> Result<Self, Box<dyn Error>>` (src/ocean.rs:25,194; src/fft.rs:19)."""
        lib = load_library()
        h = ctypes.c_void_p()
        device._check(getattr(lib, cls._init)(device._ctx, ctypes.byref(h)))
        return cls(device, h)

    def destroy(self, device=None):
        """`unsafe fn destroy(self, device)` -- consumes the object (no Drop in the reference)."""
        if self._h:
            getattr(load_library(), self._destroy)(self._h)
            self._h = None

    def _handle(self):
        if not self._h:
            raise OceanError(-5, f"{type(self).__name__} used after destroy()")
        if not self.device.alive:
            raise OceanError(-5, f"{type(self).__name__} used after its OceanDevice was destroyed")
        return self._h


class Propagation(_Stage):
    """src/ocean.rs:15-177.  `dispatch` = bind pipeline + descriptor set + dispatch [N/16, N/16, 1]
    (src/render.rs:1123-1130)."""
    _init, _destroy = "ocean_propagation_init", "ocean_propagation_destroy"

    def dispatch(self, locals_: PropagateLocals, stream=None):
        c = locals_._c()
        self.device._check(load_library().ocean_propagate(self._handle(), ctypes.byref(c), stream))


class Correction(_Stage):
    """src/ocean.rs:184-328.  `dispatch` = src/render.rs:1280-1287."""
    _init, _destroy = "ocean_correction_init", "ocean_correction_destroy"

    def dispatch(self, locals_: CorrectionLocals, stream=None):
        c = locals_._c()
        self.device._check(load_library().ocean_correct(self._handle(), ctypes.byref(c), stream))
