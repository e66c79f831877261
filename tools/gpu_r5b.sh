#!/bin/bash
# round 5, run B: ocean_frame_batch (tests, rates), 16384 in the race / shard suites
set -u
exec < /dev/null
TAG=${1:-r5b}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest (batch, normals)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or normal or abi" 2>&1 | tail -8 | tee $O/pytest_batch.txt
echo "== batch rates"; for rep in 1 2; do timeout 600 python tools/batch_time.py 256 512 1024 2048 2>&1 | tee -a $O/batch_rates.jsonl; done
echo "== bench --batch 8 at 512"; timeout 300 python bench.py --no-cpu-baseline --n 512 --batch 8 --steps 4000 --warmup 80 2>$O/bench_b8.err | tee $O/bench_n512_batch8.json | cut -c1-300; tail -3 $O/bench_b8.err
echo "== pytest 16384 race + shard"; timeout 2400 python -m pytest tests/test_gpu_race.py tests/test_sharded.py -m gpu -x -q -k "16384 or jitter" --durations=5 2>&1 | tail -15 | tee $O/pytest_16384.txt
