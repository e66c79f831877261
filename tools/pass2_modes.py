#!/usr/bin/env python3
"""Is the two-valued duration of pass 2 at N = 8192 (347 / 376 us in r05_run32) a property of the allocation or of the box's state?
One process: several contexts in a row (with dummy allocations of different sizes between them, so that the buffers move), each
measured a few times over a few seconds; prints the map's device address and the per-kernel averages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gfx_ocean_amd as g
from hipmem import DeviceBuffer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
h0, om = g.synth.make_inputs(n, seed=3)
dummies = []
for ctx in range(6):
    d = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
    d.upload_spectrum(h0, om)
    d.time_frames(100)
    for rep in range(4):
        p1, p2, _, _ = d.frame_times_ex(60)
        fps = 60 / d.time_frames(60) * 1000.0
        print(f"ctx {ctx} out=0x{d.displacement_device_ptr():x} rep {rep}: pass1 {np.mean(p1)*1000:7.1f} us  pass2 {np.mean(p2)*1000:7.1f} us (min {min(p2)*1000:6.1f} max {max(p2)*1000:6.1f})  {fps:7.1f} frames/s", flush=True)
        time.sleep(0.5)
    d.destroy()
    dummies.append(DeviceBuffer((ctx + 1) * 3 * 1024 * 1024 + 4096 * ctx))    # shifts what the next context gets
