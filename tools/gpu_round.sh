#!/bin/bash
# One gpurun call: GPU tests, smoke, membench, sweep, bench, rocprof summary.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== nproc: $(nproc) ; $(grep -m1 'model name' /proc/cpuinfo)" | tee gpurun_out/host.txt
rocm-smi --showproductname 2>/dev/null | head -20 >> gpurun_out/host.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "== membench"; timeout 300 ./tools/membench 2>&1 | tee gpurun_out/membench.jsonl
echo "== sweep"; timeout 900 python tools/sweep.py 2>&1 | tee gpurun_out/sweep.jsonl
echo "== bench"; timeout 900 python bench.py 2>&1 | tee gpurun_out/bench.json
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -12 $f; done
