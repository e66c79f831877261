#!/bin/bash
# plain A/B of gfx_ocean_amd/variants/*.so against the product:  SIZES="8192 16384" tools/gpu_ab.sh <tag>
set -u
exec < /dev/null
cd $GRAFT_REPO_ROOT
bash tools/ab_variants.sh ${1:-ab} ${SIZES:-4096} 2>&1 | tail -60
