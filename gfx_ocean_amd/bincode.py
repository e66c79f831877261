"""bincode 1.3.1 `Vec<f32>` / `Vec<[f32; 2]>` reader and writer: u64-LE element count followed
by little-endian f32 payload -- the format of the reference's data/spectrum.bin and
data/omega.bin (decoded at src/render.rs:769-771, :808-810)."""
from __future__ import annotations

import struct

import numpy as np


def read_vec_f32(path: str, lanes: int = 1) -> np.ndarray:
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 8:
        raise ValueError(f"{path}: too short for a bincode Vec header")
    (count,) = struct.unpack_from("<Q", raw, 0)
    if len(raw) != 8 + count * lanes * 4:
        raise ValueError(f"{path}: header says {count} x {lanes} f32 but file has {len(raw) - 8} payload bytes")
    a = np.frombuffer(raw, dtype="<f4", offset=8).astype(np.float32)
    return a.reshape(count, lanes) if lanes > 1 else a


def write_vec_f32(path: str, data: np.ndarray, lanes: int = 1):
    a = np.ascontiguousarray(data, dtype="<f4").reshape(-1)
    if a.size % lanes:
        raise ValueError("payload is not a whole number of elements")
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", a.size // lanes))
        f.write(a.tobytes())


def load_spectrum(spectrum_path: str, omega_path: str):
    """-> (h0 complex64 [N,N], omega float32 [N,N])"""
    spec = read_vec_f32(spectrum_path, 2)
    omega = read_vec_f32(omega_path, 1)
    n = int(round(np.sqrt(omega.size)))
    if n * n != omega.size or spec.shape[0] != omega.size:
        raise ValueError("spectrum/omega are not matching square grids")
    h0 = np.ascontiguousarray(spec, dtype=np.float32).view(np.complex64).reshape(n, n)   # bit-exact (keeps -0.0)
    return h0, omega.reshape(n, n)


def save_spectrum(spectrum_path: str, omega_path: str, h0: np.ndarray, omega: np.ndarray):
    h0 = np.ascontiguousarray(h0, dtype=np.complex64)
    write_vec_f32(spectrum_path, h0.view(np.float32), 2)
    write_vec_f32(omega_path, omega, 1)
