#!/usr/bin/env python3
"""Runs K frames of the staged 8-dispatch path (the reference's dispatch structure, src/render.rs:1122-1310) so that
rocprofv3 can collect kernel stats / HBM counters for the staged kernels:  python tools/staged_frames.py [N] [K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfx_ocean_amd as g  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h0, om = g.synth.make_inputs(n)
r = g.OceanRenderer(n)
r.upload(h0, om)
for i in range(k):
    r.render(i / 60.0)
r.device.sync()
r.dispose()
print("staged frames done", n, k)
