#!/usr/bin/env python3
"""Average duration of the normal-field kernel (ocean_normals) behind a finished frame: K back-to-back launches, one sync.
usage: [OCEAN_HIP_LIB=...] python tools/normals_time.py [N ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gfx_ocean_amd as g
from gfx_ocean_amd._lib import load_library
lib = load_library()
for n in [int(a) for a in sys.argv[1:]] or [2048]:
    h0, om = g.synth.make_inputs(n, seed=2)
    d = g.OceanDevice(n); d.upload_spectrum(h0, om); d.frame(1.0)
    for ch in (0, 1):
        for _ in range(50): lib.ocean_normals(d._ctx, ch, None)
        lib.ocean_sync(d._ctx)
        K = 2000 if n <= 4096 else 300
        t0 = time.perf_counter()
        for _ in range(K): lib.ocean_normals(d._ctx, ch, None)
        lib.ocean_sync(d._ctx)
        us = (time.perf_counter() - t0) / K * 1e6
        print(f"N={n} channel {ch}: {us:.2f} us per ocean_normals ({32.0 * n * n / us / 1e6:.2f} TB/s on 32 B/texel)")
    d.destroy()
