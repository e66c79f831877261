#!/usr/bin/env python3
"""Frames/s of ONE N x N tile sharded by row blocks over the ranks of this job (gfx_ocean_amd/sharded.py).

    python tools/shard_bench.py --n 16384                                   # one GPU, no collective
    OCEAN_SHARD_FORCE_DIST=1 python tools/shard_bench.py --n 8192           # one GPU, the RCCL all-to-all with itself
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/shard_bench.py --n 16384

One JSON line on rank 0: frames/s, ms/frame, all-to-all payload per rank and the rate it implies."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scheme", choices=("rows", "fused"), default="fused",
                    help="fused = the half-spectrum frame sharded by column blocks then row blocks (ocean_tile_*, N <= 8192, "
                         "12 B/texel exchanged); rows = the staged row-block scheme (ocean_shard_*, N <= 16384, 24 B/texel)")
    ap.add_argument("--parts", type=int, default=1, help="fused scheme: pieces of the pipelined exchange (one all-to-all each)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1 or os.environ.get("OCEAN_SHARD_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import gfx_ocean_amd as g
    from gfx_ocean_amd import sharded
    n = args.n
    h0, om = g.synth.make_inputs(n, seed=n)
    fused = args.scheme == "fused" and n <= 8192
    if fused:
        tile = sharded.FusedShardedTile(sharded.HipTileBackend(n, rank, world, local, parts=args.parts), dist)
    else:
        tile = sharded.ShardedTile(sharded.HipShardBackend(n, rank, world, local), dist)
    tile.upload(h0, om)

    frame = tile.frame

    for i in range(args.warmup):
        frame(i / 60.0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i / 60.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1000.0
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    if rank == 0:
        per = ms / args.steps
        payload = sharded.fused_exchange_bytes_per_rank(n, world) if fused else sharded.exchange_bytes_per_rank(n, world)
        print(json.dumps({"metric": "frames/s of one sharded N x N tile", "scheme": "fused" if fused else "rows", "parts": args.parts if fused else None, "n": n, "world": world, "value": 1000.0 / per,
                          "ms_per_frame": per, "all_to_all_bytes_per_rank": payload,
                          "collective": "torch.distributed.all_to_all_single (RCCL)" if dist is not None else "none (one rank)",
                          "hbm_bytes_per_texel": "26 + 28 = 54 (the fused frame's)" if fused else
                                                 "36 propagate + 3 x (16 rows + 16 transpose + 16 columns) + 40 correction = 220"}),
              flush=True)
    tile.b.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
