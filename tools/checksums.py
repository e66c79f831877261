#!/usr/bin/env python3
"""Device checksums of one fused (and staged) frame per size and spectrum format -- the fingerprint used to show that a
source clean-up left every shipped kernel's arithmetic untouched:   python tools/checksums.py > before.json; ...; diff.
(OCEAN_HIP_LIB selects another build of the same ABI.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfx_ocean_amd as g  # noqa: E402


def main():
    res = {}
    for n in (256, 512, 1024, 2048, 4096, 8192):
        for f16 in (False, True):
            h0, om = g.synth.make_inputs(n, seed=n + 1)
            r = g.OceanRenderer(n)
            r.upload(h0, om, spectrum_fp16=f16)
            for t in (0.0, 2.75, 1000.0):
                r.render_fused(t)
                res[f"{n}:{'f16' if f16 else 'f32'}:fused:t={t:g}"] = r.device.checksum()
            if n == 8192:
                r.device.set_intermediate(g.INTER_BFP16)
                r.render_fused(2.75)
                res[f"{n}:{'f16' if f16 else 'f32'}:fused_bfp16:t=2.75"] = r.device.checksum()
                r.device.set_intermediate(g.INTER_F32)
            if not f16 and n <= 4096:
                r.render(2.75)
                res[f"{n}:f32:staged:t=2.75"] = r.device.checksum()
            r.dispose()
    print(json.dumps(res, indent=0, sort_keys=True))


if __name__ == "__main__":
    main()
