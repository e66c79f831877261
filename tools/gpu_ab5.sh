#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab5}; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest gpu (defaults)"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== pytest gpu OCEAN_P=2"; OCEAN_P=2 timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do for p in 4 2; do
  echo "== rep=$rep OCEAN_P=$p"
  OCEAN_P=$p timeout 600 python tools/sweep.py 2048 4096 8192 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'frame_GBps': round(r['frame_GBps_alg']), 'fused': {k: round(v, 4) for k, v in r['fused'].items()}}))
" | tee -a $O/sweep_p.jsonl
done; done
