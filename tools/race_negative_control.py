import sys, os, json, subprocess
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
import test_gpu_race as t
def sums(lib, reps):
    env = dict(os.environ)
    if lib: env["OCEAN_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, "-c", t._JITTER_WORKER, root, str(reps)], capture_output=True, text=True, timeout=1500, env=env)
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("SUMS ")][0][5:])
want = sums(None, 2)
for name in ("nowar.so", "jitter_nowar.so"):
    got = sums(root + "/gfx_ocean_amd/variants/" + name, 64)
    bad = {k: len(v) for k, v in got.items() if v != want[k]}
    print(name, "cases that differ from the product build:", len(bad), "of", len(got), bad)
