#!/bin/bash
# The same A/B for the staged 8-dispatch frame (ocean_profile_staged, column passes listed).   N=8192 tools/gpu_ab_staged.sh <tag>
set -u
exec < /dev/null
TAG=${1:-r5k}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for so in gfx_ocean_amd/libocean_hip.so gfx_ocean_amd/variants/*.so; do
OCEAN_HIP_LIB=$PWD/$so python - <<'PY' | tee -a $O/staged_ab.txt
import os, sys; sys.path.insert(0, ".")
import numpy as np
import gfx_ocean_amd as g
n = int(os.environ.get('N', '8192'))
h0, om = g.synth.make_inputs(n, seed=3)
d = g.OceanDevice(n); d.upload_spectrum(h0, om)
d.profile_staged(0.0)
acc = {}
reps = 5
for i in range(reps):
    for k, (name, ms) in enumerate(d.profile_staged(i / 60.0)):
        acc[k] = acc.get(k, 0.0) + ms / reps
d.frame(1.0); want = d.checksum()
print(os.path.basename(os.environ["OCEAN_HIP_LIB"]), "staged frame %.3f ms" % sum(acc.values()), "stages", [round(acc[k], 3) for k in range(8)])
d.destroy()
PY
done; done
