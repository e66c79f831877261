#!/usr/bin/env python3
"""Soak test on the GPU box: create/destroy loops at several sizes (fp32 and fp16-stored spectrum), bit-reproducible frames,
a 20000-frame loop at N = 4096.  python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gfx_ocean_amd as g
# create/destroy loop at several sizes (leaks, handle registry), long frame loop, result stability
for rep in range(3):
    for n in (256, 512, 1024, 2048, 4096):
        d = g.OceanDevice(n)
        h0, om = g.synth.make_inputs(n, seed=rep)
        d.upload_spectrum(h0, om, spectrum_fp16=(rep == 1))
        d.frame(1.0); a = d.read_displacement()
        ms = d.time_frames(300) / 300
        d.frame(1.0); b = d.read_displacement()
        assert np.array_equal(a, b), ("frame not reproducible", n)
        d.destroy()
        print(rep, n, round(ms * 1000, 1), "us/frame", flush=True)
d = g.OceanDevice(4096); h0, om = g.synth.make_inputs(4096); d.upload_spectrum(h0, om)
t0 = time.time(); ms = d.time_frames(20000); print("20000 frames", round(ms / 20000 * 1000, 2), "us/frame", round(time.time() - t0, 1), "s")
d.destroy()
print("SOAK_OK")
