#!/bin/bash
set -u
exec < /dev/null
TAG=${1:-r5m}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/checksums.txt
import os, subprocess, sys
code = r'''
import sys; sys.path.insert(0, ".")
import gfx_ocean_amd as g
out = []
for n in (8192, 16384):
    h0, om = g.synth.make_inputs(n, seed=5)
    for f16 in (False, True):
        d = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY); d.upload_spectrum(h0, om, spectrum_fp16=f16); d.frame(2.5); out.append(d.checksum())
        if n == 8192 and not f16:
            d.set_frame_normals(0); d.frame(2.5); out.append(d.checksum()); import zlib; out.append(zlib.crc32(d.read_normals().tobytes()))
        d.destroy()
print(out)
'''
for lib in ("gfx_ocean_amd/libocean_hip.so", "gfx_ocean_amd/variants/real2rows.so"):
    env = dict(os.environ, OCEAN_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print(lib, r.stdout.strip(), r.stderr[-500:])
PY
for rep in 1 2; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/real2rows.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/sweep.py --fused-only 8192 16384 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'lib': '$so'.split('/')[-1], 'n': r['n'], 'fused_ms': round(r['fused_ms'], 5), 'fps': round(r['fused_fps'], 1), 'fused': {k: round(v * 1000, 2) for k, v in r['fused'].items()}}))
" | tee -a $O/ab.jsonl
  done
done
