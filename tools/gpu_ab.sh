#!/bin/bash
# A/B over the runtime knobs (OCEAN_PASS2 x OCEAN_INTER_LAYOUT) + GPU tests with the shipped defaults.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
for lay in p2 p1; do for v in thin fat; do
  echo "== sweep layout=$lay pass2=$v"
  OCEAN_INTER_LAYOUT=$lay OCEAN_PASS2=$v timeout 600 python tools/sweep.py ${SWEEP_NS:-2048 4096 8192} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'frame_GBps': round(r['frame_GBps_alg']), 'fused': {k: round(v, 4) for k, v in r['fused'].items()}}))
" | tee $O/sweep_${lay}_$v.jsonl
done; done
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tee $O/bench.json | cut -c1-600
