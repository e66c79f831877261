#!/bin/bash
# Samples rocm-smi clocks and package power while a frame loop runs at N = 4096 and 8192:  tools/gpu_clocks.sh
cd $GRAFT_REPO_ROOT
for n in 4096 8192; do
  python - <<PY &
import gfx_ocean_amd as g, time
n=$n
h0, om = g.synth.make_inputs(n, seed=1)
d = g.OceanDevice(n); d.upload_spectrum(h0, om)
t=time.time()
while time.time()-t < 14: d.time_frames(400 if n==4096 else 100, t0=0.0, dt=1/60)
PY
  pid=$!
  sleep 7
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power|socclk' | tr -s ' ' | head -8 | sed "s/^/N=$n: /"; sleep 1.5; done
  wait $pid
done
