#!/bin/bash
# after the source clean-up: fingerprints (must equal r04_run3's), the whole GPU tier, the driver's bench line, sizes.
set -u
exec < /dev/null
TAG=${1:-run4}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== checksums after the clean-up"
timeout 900 python tools/checksums.py > $O/checksums_after.json 2>$O/checksums_after.err
if cmp -s $O/checksums_after.json profiles/r04_run3_checksums_before_cleanup.json; then echo "IDENTICAL to profiles/r04_run3_checksums_before_cleanup.json"; else echo "DIFFERENT"; diff $O/checksums_after.json profiles/r04_run3_checksums_before_cleanup.json | head; fi
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-900
echo "== sweep"; timeout 900 python tools/sweep.py --fused-only 256 512 1024 2048 4096 8192 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(r['n'], 'fused %.4f ms  %.0f fps' % (r['fused_ms'], r['fused_fps']), {k: round(v * 1000, 1) for k, v in r['fused'].items()})
" | tee $O/sweep.txt
