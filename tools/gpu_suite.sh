#!/bin/bash
# GPU suite of a round: pytest -m gpu (durations), smoke(), the bench line with the driver's flags.   tools/gpu_suite.sh <tag>
set -u
exec < /dev/null
TAG=${1:-r5d}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 | tee $O/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-300
tail -3 $O/bench.err
