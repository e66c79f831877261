#!/bin/bash
# round 5, run C: batched-mode A/B (variants/*.so) + the batch test
set -u
exec < /dev/null
TAG=${1:-r5c}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch" 2>&1 | tail -4 | tee $O/pytest_batch.txt
for rep in 1 2; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 300 python tools/batch_time.py ${SIZES:-256 512 1024} 2>&1 | tee -a $O/batch_ab.jsonl
  done
done
