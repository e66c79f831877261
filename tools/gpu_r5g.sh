#!/bin/bash
set -u
exec < /dev/null
TAG=${1:-r5g}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/no_late.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/normals_time.py 2048 2>&1 | tee -a $O/late_ab.jsonl
  done
done
timeout 900 python -m pytest tests/test_gpu_race.py tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "2048" 2>&1 | tail -4 | tee $O/pytest_2048.txt
