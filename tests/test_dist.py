"""Multi-GPU path on CPU: world-size-2 gloo run of the bench's rank logic (tile seeds, barrier,
MAX-over-ranks timing, whole-job aggregation).  The data path has no collective (tiles are
independent), so this is all the N > 1 logic there is."""
import json
import os
import subprocess
import sys

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import bench, gfx_ocean_amd as g
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
n = 256
h0, om = g.synth.make_inputs(n, seed=bench.tile_seed(n, rank))
wall_ms = 10.0 * (rank + 1)                       # pretend rank r took 10(r+1) ms for 5 steps
t = torch.tensor([wall_ms], dtype=torch.float64)
dist.barrier()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
agg = bench.aggregate([float(t.item())], world, 5)
sums = [None] * world
dist.all_gather_object(sums, float(abs(h0).sum()))
if rank == 0:
    print("RESULT " + json.dumps({"agg": agg, "sums": sums}))
dist.destroy_process_group()
"""


def test_aggregate_is_whole_job_over_slowest_rank():
    a = bench.aggregate([12.0, 20.0, 16.0], n_gpus=3, steps=4)
    assert a["ms_per_step"] == 5.0 and a["value"] == 3 * 1000.0 / 5.0


def test_tile_seeds_differ_per_rank():
    assert bench.tile_seed(4096, 0) == 4096 and bench.tile_seed(4096, 3) == 4099


def test_two_rank_gloo_run(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    worker = tmp_path / "worker.py"
    worker.write_text(_WORKER)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(worker), ROOT],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0]
    r = json.loads(line[7:])
    assert r["agg"]["ms_per_step"] == 4.0                 # slowest rank: 20 ms / 5 steps
    assert r["agg"]["value"] == 2 * 1000.0 / 4.0           # two tiles per step
    assert not np.isclose(r["sums"][0], r["sums"][1])      # different tiles on different ranks
