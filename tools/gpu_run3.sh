#!/bin/bash
# full GPU test tier + fingerprints before the source clean-up + a driver-flag bench line.   tools/gpu_run3.sh <tag>
set -u
exec < /dev/null
TAG=${1:-run3}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== checksums (DMA loader at N >= 2048: the state the clean-up starts from)"
OCEAN_HIP_LIB=$PWD/gfx_ocean_amd/variants/v_dma2048.so timeout 900 python tools/checksums.py > $O/checksums_before.json 2>$O/checksums_before.err; tail -3 $O/checksums_before.json
timeout 900 python tools/checksums.py > $O/checksums_product_dma4096.json 2>/dev/null; cmp $O/checksums_before.json $O/checksums_product_dma4096.json && echo "identical to the DMA >= 4096 build"
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | tee $O/pytest_gpu.txt
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-1500
