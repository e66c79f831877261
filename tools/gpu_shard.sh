#!/bin/bash
# sharded-tile GPU tests + world-1 timings of both schemes:  tools/gpu_shard.sh <tag>
set -u
exec < /dev/null
TAG=${1:-shard}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest"; timeout 1800 python -m pytest tests/test_sharded.py -m gpu -x -q --durations=5 2>&1 | tail -25 | tee $O/pytest.txt
for n in 2048 4096 8192; do for sch in fused rows; do timeout 300 python tools/shard_bench.py --n $n --scheme $sch --steps 100 --warmup 20 2>/dev/null | tee -a $O/shard_bench_world1.jsonl | cut -c1-220; done; done
timeout 300 python tools/shard_bench.py --n 16384 --scheme rows --steps 20 2>/dev/null | tee -a $O/shard_bench_world1.jsonl | cut -c1-220
OCEAN_SHARD_FORCE_DIST=1 timeout 300 python tools/shard_bench.py --n 4096 --scheme fused --steps 100 --warmup 20 2>/dev/null | tee -a $O/shard_bench_world1.jsonl | cut -c1-260
OCEAN_SHARD_FORCE_DIST=1 timeout 300 python tools/shard_bench.py --n 4096 --scheme fused --parts 4 --steps 100 --warmup 20 2>/dev/null | tee -a $O/shard_bench_world1.jsonl | cut -c1-260
