#!/bin/bash
# HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel stats of the fused frame at other sizes than the
# headline: tools/gpu_pmc_sizes.sh <tag> N...   -> gpurun_out/<tag>/n<N>/summary.txt
set -u
TAG=${1:-pmc}; shift
export TMPDIR=/tmp
for N in "$@"; do
  O=$GRAFT_REPO_ROOT/gpurun_out/$TAG/n$N; mkdir -p $O
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --n $N --no-cpu-baseline --steps 100 --warmup 10 > $O/bench.json 2>$O/stderr.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --n $N --no-cpu-baseline --steps 10 --warmup 2 --profile-frames 2 > $O/pmc_${c}_stdout.txt 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python tools/rocprof_summary.py $O > $O/summary.txt 2>&1
  echo "== N=$N"; grep -v "^$" $O/summary.txt | cut -c1-150 | head -14
done
