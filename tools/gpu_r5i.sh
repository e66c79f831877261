#!/bin/bash
# round 5, run I: the two-step staged column pass at N >= 8192 -- tests, per-dispatch profile of the staged frame
set -u
exec < /dev/null
TAG=${1:-r5i}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_race.py -m gpu -x -q -k "full_size_properties or staged_column_pass_16384 or (consecutive and 8192) or quirk" --durations=5 2>&1 | tail -12 | tee $O/pytest.txt
python - <<'PY' | tee $O/staged_profile.txt
import sys; sys.path.insert(0, ".")
import gfx_ocean_amd as g
for n in (8192, 16384):
    h0, om = g.synth.make_inputs(n, seed=3)
    d = g.OceanDevice(n); d.upload_spectrum(h0, om)
    d.profile_staged(0.0)
    acc = {}
    reps = 5
    for i in range(reps):
        for name, ms in d.profile_staged(i / 60.0):
            acc[name] = acc.get(name, 0.0) + ms / reps
    print(n, "staged frame %.3f ms" % sum(acc.values()), {k: round(v, 3) for k, v in acc.items()})
    d.destroy()
PY
