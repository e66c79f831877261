"""GPU tier: the torch.distributed / RCCL legs of bench.py at world size 1 (the pool gives one GPU per call).

VERDICT r03 #3b: before the driver's first multi-GPU SCALE run, everything of the N > 1 path that CAN run on one device
does -- `OCEAN_BENCH_FORCE_DIST=1 python bench.py --gather ...` initialises the RCCL process group, runs the barriers and
the MAX reduction on device tensors and the final gather of BASELINE config 4 (`gather_leg`: ordered and double-buffered
schedules) through RCCL for the RGBA image and for both packed formats; what arrives at the root is checked against a
frame computed and read back here (ocean_read_displacement).  The multi-rank launcher itself is covered on gloo
(tests/test_dist.py: 2, 4 and 8 self-launched ranks)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
import gfx_ocean_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,floats,channel", [("rgba32f", 4, 0), ("rgb32f", 3, 0), ("height32f", 1, 1)])
def test_rccl_gather_at_world_1(fmt, floats, channel):
    n, gsteps = 1024, 6
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
               OCEAN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(bench.free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, BENCH, "--n", str(n), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--gather",
                        "--gather-format", fmt, "--gather-steps", str(gsteps), "--distribution-frames", "50"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                        # RCCL's banner must not reach stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and not r.get("gather_abandoned")
    gth = r["gather"]
    assert "error" not in gth, gth
    assert gth["format"] == fmt and gth["bytes_per_peer_per_frame"] == n * n * 4 * floats
    for leg in ("ordered", "overlapped"):
        assert gth[leg]["ms_per_step"] > 0 and gth[leg]["frames_per_s"] > 0
        assert gth[leg]["root_ingest_GBps"] == 0.0                   # one rank: nothing crosses a link
    # what the root holds after the last frame of the overlapped run (time (gsteps - 1) / 60, tile seed N + 0)
    h0, om = g.synth.make_inputs(n, seed=bench.tile_seed(n, 0))
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om)
        d.frame((gsteps - 1) / 60.0)
        rgba = d.read_displacement()
    finally:
        d.destroy()
    assert gth["peer_tile_first_texel"] == [float(np.float32(rgba[0, 0, channel]))]
    # the line carries SURVEY 8d's distribution
    c = r["config"]
    assert 0 < c["frame_ms_p10"] <= c["frame_ms_median"] <= c["frame_ms_p90"]
