#!/bin/bash
# round 5, run A: the frame with the normal field -- GPU tests of the new path, A/B of the normal arithmetic, bench line.
set -u
exec < /dev/null
TAG=${1:-r5a}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pytest (normals)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "normal" 2>&1 | tail -8 | tee $O/pytest_normals.txt
echo "== normals A/B"
for rep in 1 2; do
  for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
    OCEAN_HIP_LIB=$PWD/$so timeout 600 python tools/normals_time.py 512 2048 4096 8192 2>&1 | tee -a $O/normals_ab.jsonl
  done
done
echo "== bench config 3 with normals"; timeout 600 python bench.py --no-cpu-baseline --n 2048 --normals disp_x --steps 200 --warmup 20 2>$O/bench_n2048_normals.err | tee $O/bench_n2048_normals.json | cut -c1-400
echo "== bench config 3 without"; timeout 600 python bench.py --no-cpu-baseline --n 2048 --steps 200 --warmup 20 2>/dev/null | tee $O/bench_n2048.json | cut -c1-300
tail -5 $O/bench_n2048_normals.err
