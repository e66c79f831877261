#!/bin/bash
# parity at the small sizes, then A/B of gfx_ocean_amd/variants/*.so there:  tools/gpu_small_ab.sh <tag>
set -u
exec < /dev/null
TAG=${1:-sab}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== parity (shipped build)"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_race.py tests/test_gpu_native.py -m gpu -x -q -k "not 4096 and not 8192 and not full_size" 2>&1 | tail -5 | tee $O/pytest.txt
bash tools/ab_variants.sh $TAG 256 512 1024 2>&1 | tail -40
