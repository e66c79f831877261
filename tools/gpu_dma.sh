#!/bin/bash
# First measurement of the LDS-DMA pass-1 loader: parity of the shipped build at the full sizes, A/B of
# gfx_ocean_amd/variants/*.so (two interleaved repetitions), per-workgroup timelines with and without the loader.
#   tools/gpu_dma.sh <tag>
set -u
exec < /dev/null
TAG=${1:-dma}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== parity (shipped build)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_race.py tests/test_sharded.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
echo "== A/B"; bash tools/ab_variants.sh $TAG 4096 8192 2>&1 | tail -60
for n in 4096 8192; do
  for t in timeline timeline_nodma; do
    [ -x tools/$t ] && timeout 300 tools/$t $n > $O/${t}_n$n.txt 2>&1
    echo "== $t $n"; grep -A9 "^pass 1" $O/${t}_n$n.txt | cut -c1-150
  done
done
