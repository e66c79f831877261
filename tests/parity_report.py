"""Test infrastructure: prints the fused frame's parity against the C oracle, every texel, at the full sizes (python tests/parity_report.py)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gfx_ocean_amd as g
from oracle import ocean_oracle as oc
from oracle import c_oracle as cc
for n in (2048, 4096, 8192):
    h0, om = g.synth.make_inputs(n)
    cc.set_threads(min(32, cc.max_threads()))
    refc = cc.FrameRunner(h0, om).frame(0.75)
    r = g.OceanRenderer(n); r.upload(h0, om); r.render_fused(0.75)
    fused = r.displacement()
    nmax, rl2 = oc.parity_errors(fused[..., :3], refc[..., :3])
    print(n, "fused vs C oracle: normalised max", nmax.max(), "rel L2", rl2.max())
    r.dispose()
