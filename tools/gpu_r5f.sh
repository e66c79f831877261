#!/bin/bash
# round 5, run F: (a) late start of a CU's second pass-1 workgroup at N = 2048 by hardware wave slot; (b) 2 x 8 chunks at N = 8192
set -u
exec < /dev/null
TAG=${1:-r5f}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=gfx_ocean_amd/variants
run() {  # lib, sizes...
  local so=$1; shift
  OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/sweep.py --fused-only "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'lib': '$so'.split('/')[-1], 'n': r['n'], 'fused_ms': round(r['fused_ms'], 5), 'fps': round(r['fused_fps'], 1), 'fused': {k: round(v * 1000, 2) for k, v in r['fused'].items()}}))
" | tee -a $O/ab.jsonl
}
echo "== checksum of the 2x8 variant against the product at 8192"
python - <<'PY' | tee $O/chunk2x8_checksums.txt
import os, subprocess, sys, json
code = r'''
import sys; sys.path.insert(0, ".")
import gfx_ocean_amd as g
h0, om = g.synth.make_inputs(8192, seed=5)
out = []
for f16 in (False, True):
    d = g.OceanDevice(8192); d.upload_spectrum(h0, om, spectrum_fp16=f16); d.frame(2.5); out.append(d.checksum()); d.destroy()
print(out)
'''
for lib in ("gfx_ocean_amd/libocean_hip.so", "gfx_ocean_amd/variants/chunk2x8.so"):
    env = dict(os.environ, OCEAN_HIP_LIB=os.path.abspath(lib))
    print(lib, subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip())
PY
for rep in 1 2 3; do
  for so in gfx_ocean_amd/libocean_hip.so $V/late75_slot.so $V/late127_slot.so $V/late75_order.so $V/late75_inverse.so; do run $so 2048; done
done
for rep in 1 2; do
  for so in gfx_ocean_amd/libocean_hip.so $V/chunk2x8.so; do run $so 8192; done
done
