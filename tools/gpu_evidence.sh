#!/bin/bash
# Evidence run for profiles/: GPU tests, smoke, bench (with cpu baseline), rocprof kernel stats, HBM counters.
set -u
TAG=${1:-ev}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
echo "== sweep"; timeout 900 python tools/sweep.py 512 1024 2048 4096 8192 > $O/sweep.jsonl 2>&1; python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r["n"], "fused %.4f ms  %.0f fps  %.0f GB/s alg |"%(r["fused_ms"], r["fused_fps"], r["frame_GBps_alg"]), {k: round(v,4) for k,v in r["fused"].items()}, "| staged %.4f ms"%r["staged_ms_total"])
PY
echo "== bench"; timeout 900 python bench.py 2>&1 | tee $O/bench.json | cut -c1-400
cd /tmp
echo "== rocprof kernel stats (same command as bench, no cpu baseline)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/rocprof_stats_stdout.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --profile-frames 2 > $O/pmc_${c}_stdout.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | head -40
