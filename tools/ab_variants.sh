#!/bin/bash
# A/B of alternative builds of the same ABI: tools/ab_variants.sh <tag> [N ...]; runs tools/sweep.py per library
# (gfx_ocean_amd/libocean_hip.so and every gfx_ocean_amd/variants/*.so), two interleaved repetitions.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-abv}; mkdir -p $O; shift
NS=${@:-4096}
for rep in 1 2; do
for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
  OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/sweep.py --fused-only $NS 2>&1 < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'lib': '$so'.split('/')[-1], 'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'fused': {k: round(v * 1000, 1) for k, v in r['fused'].items()}}))
" | tee -a $O/ab.jsonl
done; done
