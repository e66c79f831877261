"""ctypes binding of oracle/libocean_oracle.so (the C restatement; test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libocean_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        vp, f32, i32 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int32
        L.oracle_propagate.argtypes = [vp, vp, i32, f32, f32, vp, vp, vp]
        L.oracle_fft_rows.argtypes = [vp, i32]
        L.oracle_fft_cols.argtypes = [vp, i32]
        L.oracle_correct.argtypes = [vp, vp, vp, i32, vp]
        L.oracle_frame.argtypes = [vp, vp, i32, f32, f32, vp, vp]
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        for fn in ("oracle_propagate", "oracle_fft_rows", "oracle_fft_cols", "oracle_correct",
                   "oracle_frame", "oracle_set_threads"):
            getattr(L, fn).restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def set_threads(n: int):
    lib().oracle_set_threads(int(n))


def propagate(h0, omega, time, domain_size=1000.0):
    n = h0.shape[0]
    h0 = np.ascontiguousarray(h0, np.complex64)
    omega = np.ascontiguousarray(omega, np.float32)
    outs = [np.empty((n, n), np.complex64) for _ in range(3)]
    lib().oracle_propagate(_p(h0), _p(omega), n, float(time), float(domain_size), *map(_p, outs))
    return tuple(outs)  # height, disp_x, disp_z


def fft_rows(field):
    f = np.ascontiguousarray(field, np.complex64).copy()
    lib().oracle_fft_rows(_p(f), f.shape[0])
    return f


def fft_cols(field):
    f = np.ascontiguousarray(field, np.complex64).copy()
    lib().oracle_fft_cols(_p(f), f.shape[0])
    return f


def correct(height, disp_x, disp_z):
    n = height.shape[0]
    args = [np.ascontiguousarray(a, np.complex64) for a in (height, disp_x, disp_z)]
    out = np.empty((n, n, 4), np.float32)
    lib().oracle_correct(*map(_p, args), n, _p(out))
    return out


class FrameRunner:
    """Keeps the scratch buffers so repeated frames time only the path itself."""

    def __init__(self, h0, omega, domain_size=1000.0):
        self.n = h0.shape[0]
        self.h0 = np.ascontiguousarray(h0, np.complex64)
        self.omega = np.ascontiguousarray(omega, np.float32)
        self.L = float(domain_size)
        self.work = np.empty((3, self.n, self.n), np.complex64)
        self.out = np.empty((self.n, self.n, 4), np.float32)

    def frame(self, time):
        lib().oracle_frame(_p(self.h0), _p(self.omega), self.n, float(time), self.L,
                           _p(self.work), _p(self.out))
        return self.out
