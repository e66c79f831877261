#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab4}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lay in p2 p1; do for nt in 3 0; do
  echo "== rep=$rep layout=$lay NT=$nt"
  OCEAN_INTER_LAYOUT=$lay OCEAN_NT=$nt timeout 600 python tools/sweep.py 4096 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'frame_GBps': round(r['frame_GBps_alg']), 'fused': {k: round(v, 4) for k, v in r['fused'].items()}}))
" | tee -a $O/sweep_layout_nt.jsonl
done; done; done
