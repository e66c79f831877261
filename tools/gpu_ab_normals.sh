#!/bin/bash
# The same A/B for the frame with the normal field (tools/normals_time.py).   SIZES="2048 4096" tools/gpu_ab_normals.sh <tag>
set -u
exec < /dev/null
TAG=${1:-r5e}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/normals_time.py ${SIZES:-2048 4096} 2>&1 | tee -a $O/normals_ab.jsonl
  done
done
