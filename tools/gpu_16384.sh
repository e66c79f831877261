#!/bin/bash
# N = 16384 on one GPU: per-kernel times of the fused frame, rocprofv3 stats and HBM counters.  tools/gpu_16384.sh <tag>
set -u
exec < /dev/null
TAG=${1:-n16384}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/sweep.py --fused-only 16384 2>&1 | tee $O/sweep_n16384.jsonl | cut -c1-400
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fused_n16384/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 16384 --steps 20 --warmup 2 --profile-frames 2 --ramp-frames 10 --distribution-frames 20 > $O/pmc_${c}_stdout.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/fused_n16384 > $O/fused_n16384.summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/make_hbm_traffic.py $O/fused_n16384 16384 $TAG - > $O/fused_n16384.hbm_traffic.txt 2>&1
cp $GRAFT_REPO_ROOT/profiles/hbm_traffic_n16384.json $O/ 2>/dev/null
grep -v "^$" $O/fused_n16384.summary.txt | cut -c1-200 | head; cat $O/fused_n16384.hbm_traffic.txt
find $O -name "*.csv" -size +2M -delete
