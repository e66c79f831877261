#!/bin/bash
# A/B: OCEAN_ALGO=half|c2c at several N, plus GPU tests under both.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab2}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for a in half c2c; do
  echo "== pytest gpu ALGO=$a"; OCEAN_ALGO=$a timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu_$a.txt
  echo "== sweep ALGO=$a"
  OCEAN_ALGO=$a timeout 600 python tools/sweep.py ${SWEEP_NS:-512 1024 2048 4096 8192} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_ms': round(r['fused_ms'], 4), 'fps': round(r['fused_fps'], 1), 'frame_GBps': round(r['frame_GBps_alg']), 'fused': {k: round(v, 4) for k, v in r['fused'].items()}}))
" | tee $O/sweep_$a.jsonl
done
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tee $O/bench.json | cut -c1-500
