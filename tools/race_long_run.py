#!/usr/bin/env python3
"""Long form of tests/test_gpu_race.py::test_barrier_jitter_build_is_bit_identical: 1000 frames per case in the barrier-jitter
build (every workgroup barrier wrapped in pseudo-random sleeps), every size, fp32 and fp16-stored spectrum, fused and staged;
all checksums must equal the product build's.   python tools/race_long_run.py [reps [sizes, e.g. 16384]]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_race as t
from gfx_ocean_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
if len(sys.argv) > 2:
    os.environ["OCEAN_RACE_SIZES"] = sys.argv[2]
so = os.path.join(tempfile.mkdtemp(), "libocean_hip_jitter.so")
subprocess.check_call(_lib.hipcc_command(out=so, extra=("-DOCEAN_RACE_JITTER",)))
def sums(lib, n):
    env = dict(os.environ)
    if lib: env["OCEAN_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, "-c", t._JITTER_WORKER, ROOT, str(n)], capture_output=True, text=True, timeout=3000, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("SUMS ")][0][5:])
want = sums(None, 2)
got = sums(so, reps)
bad = {k: len(v) for k, v in got.items() if v != want[k]}
print(f"barrier-jitter build, {reps} fused frames ({max(4, reps // 8)} staged) per case, {len(got)} cases: "
      f"{'ALL BIT-IDENTICAL to the product build' if not bad else 'DIFFERENCES ' + str(bad)}")
sys.exit(1 if bad else 0)
