#!/usr/bin/env python3
"""Per-frame durations of the two fused kernels over 80 back-to-back frames (events bound to the dispatches):  frame_series.py N"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gfx_ocean_amd as g
n = int(sys.argv[1])
h0, om = g.synth.make_inputs(n, seed=3)
d = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
d.upload_spectrum(h0, om)
d.time_frames(200)
p1, p2, _, _ = d.frame_times_ex(80)
print(n, "pass1", " ".join(f"{v*1000:.0f}" for v in p1))
print(n, "pass2", " ".join(f"{v*1000:.0f}" for v in p2))
