#!/usr/bin/env python3
"""Numerics of a 16-bit intermediate between the two passes (BASELINE config 5 "fp16 inter-pass", SURVEY 8d B_frame16).

Evaluates, with numpy in fp64 except for the quantisation itself, what happens to the final displacement map when the
half-spectrum intermediate (the column-transformed, symmetrised spectra of kx < N/2 that pass 1 hands to pass 2) is
stored in 16 bits per component:
   fp16      -- IEEE half with one power-of-two scale per 4 x 4 chunk (the "block-scaled fp16" of VERDICT r01 #3c)
   bfp16     -- int16 mantissas with one shared exponent per chunk row piece (4 complex = 8 values, 16 bits each)
   bfp15     -- the same with 15-bit mantissas (the 16th bit of each value carrying one bit of the exponent, so that a
                16-byte row piece is self-describing)
Parity metric as everywhere: normalised max and relative L2 per channel against the unquantised result.
usage: python tools/inter16_numerics.py [N] [t]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfx_ocean_amd as g  # noqa: E402
from oracle import ocean_oracle as oc  # noqa: E402


def symmetrised_column_transforms(fields):
    """For each complex field F[y, x]: S(F) = (F + conj(F[-y, -x])) / 2, transformed along y, columns kx < N/2 + Nyquist."""
    out = []
    for F in fields:
        n = F.shape[0]
        idx = (-np.arange(n)) % n
        S = 0.5 * (F + np.conj(F[np.ix_(idx, idx)]))
        out.append(n * np.fft.ifft(S, axis=0))          # unnormalised inverse transform along y
    return out


def finish(cols, n):
    """Row transform of the (Hermitian along x) column transforms, real part, correction sign."""
    gx = np.arange(n)
    sign = np.where(((gx[None, :] + gx[:, None]) % 2) == 0, -1.0, 1.0)
    return [np.real(n * np.fft.ifft(c, axis=1)) * sign for c in cols]


def requantise(c, mode):
    """Quantise the kx < N/2 half (what is stored); the other half is rebuilt from it by Hermitian symmetry."""
    n = c.shape[0]
    half = c[:, : n // 2].copy()
    v = np.stack([half.real, half.imag], -1)                                  # [y, kx, 2]
    if mode == "fp16":
        blk = v.reshape(n // 4, 4, n // 8, 4, 2)
        mx = np.abs(blk).max(axis=(1, 3, 4), keepdims=True)
        e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))), 0)
        scale = 2.0 ** (14 - e)                                               # max lands in [2^14, 2^15)
        q = (blk * scale).astype(np.float16).astype(np.float64) / scale
        v = q.reshape(n, n // 2, 2)
    else:
        bits = 16 if mode == "bfp16" else 15
        blk = v.reshape(n, n // 8, 4, 2)                                      # row piece: 4 complex
        mx = np.abs(blk).max(axis=(2, 3), keepdims=True)
        e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))) + 1, 0)   # |v| < 2^e
        step = 2.0 ** (e - (bits - 1))
        q = np.clip(np.rint(blk / step), -(2 ** (bits - 1)), 2 ** (bits - 1) - 1) * step
        v = q.reshape(n, n // 2, 2)
    half = v[..., 0] + 1j * v[..., 1]
    full = np.empty_like(c)
    full[:, : n // 2] = half
    # Hermitian along x after the y transform: C[y, N - kx] = conj(C[y, kx]); the Nyquist column is real (kept exact:
    # it rides in column 0's imaginary part in the kernels and is quantised with it -- second-order here)
    full[:, n // 2] = c[:, n // 2]
    full[:, n // 2 + 1:] = np.conj(half[:, 1:][:, ::-1])
    return full


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    t = float(sys.argv[2]) if len(sys.argv) > 2 else 1.25
    h0, om = g.synth.make_inputs(n)
    H, DX, DZ = oc.propagate_f64(h0, om, t)
    cols = symmetrised_column_transforms([DX, H, DZ])
    ref = np.stack(finish(cols, n), -1)
    chk = oc.frame_f64(h0, om, t)[..., :3]
    base = oc.parity_errors(ref, chk)
    res = {"n": n, "t": t, "self_check_vs_frame_f64": [float(base[0].max()), float(base[1].max())]}
    for mode in ("fp16", "bfp16", "bfp15"):
        out = np.stack(finish([requantise(c, mode) for c in cols], n), -1)
        nmax, rl2 = oc.parity_errors(out, ref)
        res[mode] = {"normalised_max": [float(x) for x in nmax], "rel_l2": [float(x) for x in rl2]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
