#!/usr/bin/env python3
"""Per-kernel timing sweep over N for DESIGN.md / profiles: fused and staged paths, HIP events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (moved_bytes_per_texel: the one accounting table)
import gfx_ocean_amd as g  # noqa: E402


def main():
    fused_only = "--fused-only" in sys.argv
    ns = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [256, 512, 1024, 2048, 4096, 8192]
    for n in ns:
        h0, om = g.synth.make_inputs(n)
        d = g.OceanDevice(n)
        d.upload_spectrum(h0, om)
        for i in range(200 if n <= 4096 else 40):      # enough untimed work for the GPU to reach its running clocks
            d.frame(i / 60.0)
        d.sync()
        frames = 200 if n <= 4096 else 50
        ms = d.time_frames(frames) / frames
        moved = bench.moved_bytes_per_texel(n)
        rec = {"n": n, "fused_ms": ms, "fused_fps": 1000.0 / ms, "frame_GBps": sum(moved.values()) * n * n / ms / 1e6,
               "frame_GBps_alg": 76.0 * n * n / ms / 1e6}
        reps = 10
        for kind, fn in (("fused", d.profile_frame),) + (() if fused_only else (("staged", d.profile_staged),)):
            acc = {}
            fn(0.0)
            for i in range(reps):
                for name, t in fn(i / 60.0):
                    acc[name] = acc.get(name, 0.0) + t / reps
            rec[kind] = acc
        # bytes the half-spectrum kernels move (bench.moved_bytes_per_texel) and the three-complex-transform accounting (36 / 40)
        rec["fused_GBps"] = {k: moved[bench.pass_of(k)] * n * n / v / 1e6 for k, v in rec["fused"].items()}
        rec["fused_GBps_contract"] = {k: (36.0 if "pass1" in k else 40.0) * n * n / v / 1e6 for k, v in rec["fused"].items()}
        if not fused_only:
            st = rec["staged"]
            # the eight stages in order: propagate 12 R + 24 W (paired kernel), 3 row and 3 column passes 8 R + 8 W each (N >= 8192: the
            # column pass's first step), correction 24 R + 16 W (N >= 8192: with the column passes' second step, the same bytes)
            per_stage = [36.0] + [16.0] * 6 + [40.0]
            rec["staged_GBps"] = {k: b * n * n / v / 1e6 for (k, v), b in zip(st.items(), per_stage)}
            rec["staged_ms_total"] = sum(st.values())
        print(json.dumps(rec), flush=True)
        d.destroy()


if __name__ == "__main__":
    main()
