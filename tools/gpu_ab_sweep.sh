#!/bin/bash
# A/B of the library against every build in gfx_ocean_amd/variants/ (tools/build_variants.py): per-kernel sweep, three interleaved
# repetitions.   SIZES="2048 8192" tools/gpu_ab_sweep.sh <tag>
set -u
exec < /dev/null
TAG=${1:-r5l}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/sweep.py --fused-only ${SIZES:-1024} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'lib': '$so'.split('/')[-1], 'n': r['n'], 'fused_ms': round(r['fused_ms'], 5), 'fps': round(r['fused_fps'], 1), 'fused': {k: round(v * 1000, 2) for k, v in r['fused'].items()}}))
" | tee -a $O/ab.jsonl
  done
done
