#!/bin/bash
# latency path sizes: sweep + per-workgroup timelines:  tools/gpu_small.sh <tag>
set -u
exec < /dev/null
TAG=${1:-small}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout 600 python tools/sweep.py --fused-only 256 512 1024 2048 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'n': r['n'], 'fused_us': round(r['fused_ms']*1000, 2), 'fps': round(r['fused_fps']), 'fused': {k: round(v * 1000, 2) for k, v in r['fused'].items()}}))
" | tee -a $O/sweep_small.jsonl; done
for n in 512 1024; do timeout 120 ./tools/timeline $n > $O/timeline_n$n.txt 2>&1; grep -v "^  *[0-9.]* *[0-9]* *[0-9]* *[0-9]*$" $O/timeline_n$n.txt | head -40; done
