#!/bin/bash
# parity of the shipped build at the full sizes, then an A/B of gfx_ocean_amd/variants/*.so:  tools/gpu_ab.sh <tag> N...
set -u
exec < /dev/null
TAG=${1:-ab}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== parity (shipped build)"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_race.py -m gpu -x -q -k "full_size or config5 or fused_frame or bit_identical or checksum or packed" 2>&1 | tail -5 | tee $O/pytest.txt
bash tools/ab_variants.sh $TAG "$@" 2>&1 | tail -40
