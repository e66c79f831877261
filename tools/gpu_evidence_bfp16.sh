#!/bin/bash
# Evidence for the opt-in 16-bit intermediate (config 5): bench lines with and without it on ONE box, rocprofv3 stats and
# HBM counters of the bfp16 run:  tools/gpu_evidence_bfp16.sh <tag>
set -u
exec < /dev/null
TAG=${1:-bfp16}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --n 8192 --spectrum f16 --steps 50 --warmup 5 2>/dev/null > $O/bench_n8192_f16_rep$rep.json
  timeout 600 python bench.py --no-cpu-baseline --n 8192 --spectrum f16 --intermediate bfp16 --steps 50 --warmup 5 2>/dev/null > $O/bench_n8192_f16_bfp16_rep$rep.json
  timeout 600 python bench.py --no-cpu-baseline --n 8192 --steps 50 --warmup 5 2>/dev/null > $O/bench_n8192_rep$rep.json
  timeout 600 python bench.py --no-cpu-baseline --n 8192 --intermediate bfp16 --steps 50 --warmup 5 2>/dev/null > $O/bench_n8192_bfp16_rep$rep.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    r = json.load(open(f)); print(f.split("/")[-1], round(r["value"], 1), [(k["name"], round(k["avg_ms"] * 1000, 1)) for k in r["roofline"]["kernels"]])
PY
cd /tmp
name=fused_n8192_f16_bfp16
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 8192 --spectrum f16 --intermediate bfp16 --steps 60 --warmup 2 --profile-frames 2"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/stats -o run -- $CMD > $O/$name.stats_stdout.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$name/pmc_$c -o run -- $CMD > $O/$name.pmc_${c}_stdout.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/$name > $O/$name.summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/make_hbm_traffic.py $O/$name 8192 $TAG f16 bfp16 > $O/$name.hbm_traffic.txt 2>&1
cp $GRAFT_REPO_ROOT/profiles/hbm_traffic_n8192_f16_bfp16.json $O/ 2>/dev/null
grep -v "^$" $O/$name.summary.txt | cut -c1-170 | head -30
find $O -name "*.csv" -size +2M -delete
