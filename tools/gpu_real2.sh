#!/bin/bash
# real-output pass 2 (k_half_pass2_real): parity of the variant builds, then A/B against the product.   tools/gpu_real2.sh <tag>
set -u
exec < /dev/null
TAG=${1:-real2}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-v_real2_w4}; do
  echo "== parity with $v"
  OCEAN_HIP_LIB=$PWD/gfx_ocean_amd/variants/$v.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q \
    -k "${PARITY_K:-full_size_against_c_oracle or synthetic_against_c_oracle or bfp16 or config5 or properties or sharded or tile}" 2>&1 | tail -8 | tee $O/parity_$v.txt
done
echo "== A/B"; bash tools/ab_variants.sh $TAG ${SIZES:-2048 4096 8192 16384} 2>&1 | tail -80
