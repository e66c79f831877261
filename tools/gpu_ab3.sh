#!/bin/bash
# A/B: OCEAN_NT = 0,1,2,3 (non-temporal stores) for both algorithms at N=4096 (+2048, 8192 for the default)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab3}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for a in half c2c; do for nt in 0 1 2 3; do
  echo "== ALGO=$a NT=$nt"
  OCEAN_ALGO=$a OCEAN_NT=$nt timeout 600 python tools/sweep.py 4096 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'frame_GBps': round(r['frame_GBps_alg']), 'fused': {k: round(v, 4) for k, v in r['fused'].items()}}))
" | tee -a $O/sweep_nt.jsonl
done; done
echo "== pytest gpu (defaults)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tee $O/bench.json | cut -c1-330
