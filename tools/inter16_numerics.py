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
import re
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
    elif mode.startswith("bfp16_r"):
        # int16 mantissas, ONE exponent per block of R rows x C columns kept in a side array: the blocks a
        # pass-1 wave stores with one instruction -- 64 lanes x (2 columns of one row) = 64 rows x 2 columns at N = 8192
        # (P = 2), 32 rows x 4 columns at N = 4096 (P = 4) -- so that the exponent is a wave max-reduction
        r_, c_ = (int(x) for x in re.match(r"bfp16_r(\d+)c(\d+)", mode).groups())
        blk = v.reshape(n // r_, r_, (n // 2) // c_, c_, 2)
        mx = np.abs(blk).max(axis=(1, 3, 4), keepdims=True)
        e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))) + 1, 0)
        step = 2.0 ** (e - 15)
        q = np.clip(np.rint(blk / step), -(2 ** 15), 2 ** 15 - 1) * step
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
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(pos[0]) if len(pos) > 0 else 2048
    t = float(pos[1]) if len(pos) > 1 else 1.25
    h0, om = g.synth.make_inputs(n)
    if "--f16-spectrum" in sys.argv:                  # config 5: the spectrum the kernels use is the fp16-quantised one
        s_ = 14 - int(np.floor(np.log2(np.abs(h0.view(np.float32)).max())))
        q = (h0.view(np.float32) * np.float32(2.0 ** s_)).astype(np.float16).astype(np.float32) * np.float32(2.0 ** -s_)
        h0 = q.view(np.complex64).reshape(h0.shape)
    H, DX, DZ = oc.propagate_f64(h0, om, t)
    cols = symmetrised_column_transforms([DX, H, DZ])
    ref = np.stack(finish(cols, n), -1)
    res = {"n": n, "t": t, "f16_spectrum": "--f16-spectrum" in sys.argv}
    if n <= 4096:
        chk = oc.frame_f64(h0, om, t)[..., :3]
        base = oc.parity_errors(ref, chk)
        res["self_check_vs_frame_f64"] = [float(base[0].max()), float(base[1].max())]
    modes = ["fp16", "bfp16", "bfp15", "bfp16_r4c4", "bfp16_r32c4", "bfp16_r64c2", "bfp16_r16c2",
             f"bfp16_r1024c2", f"bfp16_r{n}c2", f"bfp16_r{n}c1", f"bfp16_r{n}c4"]     # one exponent per (field, column [pair]): a whole line
    if "--modes" in sys.argv:
        modes = sys.argv[sys.argv.index("--modes") + 1].replace("N", str(n)).split(",")
    for mode in modes:
        out = np.stack(finish([requantise(c, mode) for c in cols], n), -1)
        nmax, rl2 = oc.parity_errors(out, ref)
        res[mode] = {"normalised_max": [float(x) for x in nmax], "rel_l2": [float(x) for x in rl2]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
