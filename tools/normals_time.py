#!/usr/bin/env python3
"""The normal field's cost, three ways, behind a finished frame (K back-to-back launches, one sync):
  rgba   -- ocean_normals: k_normals from the RGBA map (16 + 16 B/texel);
  frame  -- the frame with the normal field (ocean_set_frame_normals): per-kernel begin/end events of pass 1, pass 2 (+ plane)
            and k_normals_plane (4 + 16 B/texel), and the frame rate of the plain loop with and without the field.
usage: [OCEAN_HIP_LIB=...] python tools/normals_time.py [N ...]"""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gfx_ocean_amd as g
from gfx_ocean_amd._lib import load_library
lib = load_library()
for n in [int(a) for a in sys.argv[1:]] or [2048]:
    h0, om = g.synth.make_inputs(n, seed=2)
    d = g.OceanDevice(n); d.upload_spectrum(h0, om); d.frame(1.0)
    rec = {"lib": os.path.basename(os.environ.get("OCEAN_HIP_LIB", "libocean_hip.so")), "n": n}
    K = 2000 if n <= 4096 else 300
    for ch in (0, 1):
        for _ in range(50): lib.ocean_normals(d._ctx, ch, None)
        lib.ocean_sync(d._ctx)
        t0 = time.perf_counter()
        for _ in range(K): lib.ocean_normals(d._ctx, ch, None)
        lib.ocean_sync(d._ctx)
        rec[f"rgba_ch{ch}_us"] = round((time.perf_counter() - t0) / K * 1e6, 2)
    F = 400 if n <= 4096 else 60
    d.time_frames(100)
    rec["frame_ms"] = round(d.time_frames(F) / F, 5)
    if hasattr(d, "set_frame_normals"):
        d.set_frame_normals(0)
        d.time_frames(50)
        rec["frame_with_normals_ms"] = round(d.time_frames(F) / F, 5)
        p1, p2, nr, _ = d.frame_times_ex(min(F, 200))
        med = lambda v: sorted(v)[len(v) // 2]
        rec.update({"pass1_us": round(med(p1) * 1e3, 2), "pass2_plane_us": round(med(p2) * 1e3, 2), "normals_plane_us": round(med(nr) * 1e3, 2)})
        d.set_frame_normals(None)
        p1, p2, _, _ = d.frame_times_ex(min(F, 200))
        rec.update({"pass2_us": round(med(p2) * 1e3, 2)})
        rec["fps_with_normals"] = round(1000.0 / rec["frame_with_normals_ms"], 1)
    print(json.dumps(rec), flush=True)
    d.destroy()
