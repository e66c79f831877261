#!/usr/bin/env python3
"""ocean_frame_batch: frames/s of K time steps per launch pair against the plain frame loop.  usage: python tools/batch_time.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gfx_ocean_amd as g
for n in [int(a) for a in sys.argv[1:]] or [512]:
    h0, om = g.synth.make_inputs(n, seed=2)
    d = g.OceanDevice(n); d.upload_spectrum(h0, om)
    d.time_frames(400)
    F = 4000 if n <= 1024 else 400
    rec = {"lib": os.path.basename(os.environ.get("OCEAN_HIP_LIB", "libocean_hip.so")), "n": n, "plain_fps": round(1000.0 * F / d.time_frames(F), 1)}
    for k in [int(v) for v in os.environ.get('OCEAN_BATCH_KS', '1,2,4,8,16,32,64').split(',')]:
        d.time_frame_batch(20, k)
        L = max(10, F // k)
        rec[f"batch{k}_fps"] = round(1000.0 * L * k / d.time_frame_batch(L, k), 1)
    print(json.dumps(rec), flush=True)
    d.destroy()
