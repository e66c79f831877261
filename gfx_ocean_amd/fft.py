"""Mirror of the reference's `mod fft` (src/fft.rs): `Fft` with a row pass and a column pass
over three field buffers (desc_sets[0,1,2] -> dx, dy, dz; src/render.rs:971-988)."""
from __future__ import annotations

from ._lib import load_library
from .ocean import _Stage

FIELD_DX, FIELD_DY, FIELD_DZ, FIELD_ALL = 0, 1, 2, -1


class Fft(_Stage):
    _init, _destroy = "ocean_fft_init", "ocean_fft_destroy"

    def row_pass(self, field: int = FIELD_ALL, stream=None):
        """bind row_pass; dispatch [1, N, 1] for each set (src/render.rs:1158-1179)."""
        self.device._check(load_library().ocean_fft_rows(self._handle(), int(field), stream))

    def col_pass(self, field: int = FIELD_ALL, stream=None):
        """bind col_pass; dispatch [1, N, 1] for each set (src/render.rs:1210-1231)."""
        self.device._check(load_library().ocean_fft_cols(self._handle(), int(field), stream))
