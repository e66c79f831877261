#!/bin/bash
# One attempt at an AddressSanitizer run of the device code (SURVEY 5 "race detection"): the library built
# with  hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g -O1  (gfx_ocean_amd/variants/libocean_hip_asan.so),
# HSA_XNACK=1, frames at N = 256 .. 2048 against the oracle.  The log goes to profiles/ whatever the outcome.
set -u
exec < /dev/null
TAG=${1:-asan}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
{
echo "# asan runtime: $RT"; echo "# instrumented ROCm runtime (/opt/rocm/lib/asan): $(ls /opt/rocm/lib/asan 2>/dev/null | head -3 | tr '\n' ' ')"
echo "# xnack: $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m2 'xnack' | tr '\n' ' ')"
HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:abort_on_error=0 LD_PRELOAD=$RT \
OCEAN_HIP_LIB=$PWD/gfx_ocean_amd/variants/libocean_hip_asan.so timeout 420 python - <<'PY'
import numpy as np, gfx_ocean_amd as g
print("library:", g.load_library()._name)
for n in (256, 512, 1024, 2048):                 # (a sanitizer run checks memory accesses, not values: fused against staged is enough)
    h0, om = g.synth.make_inputs(n, seed=n)
    r = g.OceanRenderer(n)
    r.upload(h0, om)
    r.render_fused(1.5); a = r.displacement()
    r.render(1.5); b = r.displacement()
    print(f"N={n}: fused vs staged max abs {np.abs(a - b).max():.2e}", flush=True)
    r.dispose()
print("ASAN_RUN_COMPLETE")
PY
echo "# exit status: $?"
} > $O/asan_log.txt 2>&1
tail -40 $O/asan_log.txt
