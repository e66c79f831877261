#!/usr/bin/env python3
"""Build A/B variants of the library (same ABI) into gfx_ocean_amd/variants/ for tools/ab_variants.sh:
    python tools/build_variants.py "v_name:-DSOME_FLAG -DOTHER=1" ["PRODUCT:" rebuilds gfx_ocean_amd/libocean_hip.so itself]
(csrc/ carries no A/B switches: a variant is a temporary source edit behind a -D, measured, recorded in EXPERIMENTS.md and
removed again; round 4's are in git history.)"""
import sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gfx_ocean_amd as g
from concurrent.futures import ThreadPoolExecutor
V = dict(a.split(':',1) for a in sys.argv[1:])
def build(item):
    name, flags = item
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gfx_ocean_amd', 'variants', name + '.so') if name != 'PRODUCT' else g.library_path()
    cmd = g._lib.hipcc_command(out=out, extra=flags.split())
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-1500:] if r.returncode else ''
with ThreadPoolExecutor(4) as ex:
    for name, rc, err in ex.map(build, V.items()):
        print(name, rc, err)
