#!/usr/bin/env python3
"""bench.py -- ocean frames/sec + achieved HBM GB/s on N x N tiles, one tile per MI355X.

A "step" is one frame of the hot path on one tile per GPU: spectrum propagate -> 2-D inverse FFT of
the three fields -> sign correction + RGBA pack (the fused 2-launch path, `ocean_frame`), with h0 and
omega already resident in HBM.  Tiles are independent, so N GPUs = N tiles per step
and no data-path collective ("weak" scaling).  One JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# Algorithmic bytes per texel of each fused kernel (DESIGN.md "Kernels"):
#   pass1: read h0 8 + omega 4, write 3 complex fields 24            = 36
#   pass2: read 3 complex fields 24, write RGBA32F 16                = 40   (frame total 76)
KERNEL_BYTES_PER_TEXEL = {"pass1": 36.0, "pass2": 40.0}
# Bytes the shipped half-spectrum algorithm itself has to move: only columns
# kx < N/2 of the three fields cross between the passes (12 B/texel instead of 24).
#   pass1: read h0 10 + omega 4 (lines x, x-1, N-x, N-1-x), write 12     = 26
#   pass2: read 12, write RGBA32F 16                                      = 28   (frame total 54)
HALF_BYTES_PER_TEXEL = {"pass1": 26.0, "pass2": 28.0}


def pass_of(kernel_name):
    return "pass1" if "pass1" in kernel_name else "pass2"


def aggregate(values_ms, n_gpus, steps):
    """Whole-job throughput from per-rank wall times: tiles processed / slowest rank's time."""
    worst_ms = max(values_ms)
    ms_per_step = worst_ms / steps
    return {"ms_per_step": ms_per_step, "value": n_gpus * 1000.0 / ms_per_step}


def measured_traffic(n, kernel_name):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command, never inside the timed bench;
    gfx950 correction 2*FETCH_SIZE + WRITE_SIZE -- DESIGN.md 7).  None if no pass exists for this N."""
    path = os.path.join(ROOT, "profiles", f"hbm_traffic_n{n}.json")
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    v = rec.get(kernel_name)
    return v["hbm_bytes"] if v else None


def tile_seed(n, rank):
    return n + rank     # tile r of a multi-GPU run uses seed N + r


def cpu_baseline(n, h0, omega, budget_s=10.0, max_frames=12):
    """The C restatement of the reference shaders (oracle/, kind "port") timed on this box's host
    cores on a bounded sample of the same workload: whole frames of the same N until ~budget_s.
    The thread count is the best of {all hardware threads, half, 64, 32, 16} on one probe frame each
    (the strided column pass does not scale to 256 SMT threads on a 2-socket box)."""
    from oracle import c_oracle as cc          # cpu_baseline leg: the oracle as the measured CPU path
    cc.build()
    runner = cc.FrameRunner(h0, omega)
    hw = cc.max_threads()
    runner.frame(0.0)                          # first-touch of the scratch buffers, not timed
    probes = {}
    for th in sorted({hw, max(1, hw // 2), min(hw, 64), min(hw, 32), min(hw, 16)}, reverse=True):
        cc.set_threads(th)
        t0 = time.perf_counter()
        runner.frame(0.5)
        probes[th] = time.perf_counter() - t0
    best = min(probes, key=probes.get)
    cc.set_threads(best)
    frames, t0 = 0, time.perf_counter()
    while frames < max_frames and (time.perf_counter() - t0) < budget_s:
        runner.frame(frames / 60.0)
        frames += 1
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": best, "kind": "port",
            "sample": f"{frames} whole frames at N={n} ({dt:.1f} s), OpenMP over lines with {best} threads "
                      f"(best of probes {{{', '.join(f'{k}: {v:.2f} s' for k, v in probes.items())}}}, "
                      f"{hw} hardware threads), radix-2 Stockham with sincosf per butterfly as in the shaders"}


def gather_leg(dev, dist, torch, n, n_gpus, rank, steps):
    """BASELINE config 4 / SURVEY 8e: every tile's RGBA map gathered to rank 0 with one RCCL collective per frame
    (root ingest N*N*16 B per peer over xGMI).  Two schedules, both reported, neither part of `value`:
    `ordered`   -- frame and collective on one stream;
    `overlapped` -- two output buffers; the collective of frame f runs on a second stream while frame f+1 is
                    computed (the frame is ~0.2 ms, the root's ingest of 7 x 256 MiB ~1.8 ms: the pipeline is
                    gather-bound and the overlap hides the compute, not the other way round)."""
    outs = [torch.empty((n, n, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    dsts = [[torch.empty_like(outs[0]) for _ in range(n_gpus)] if rank == 0 else None for _ in range(2)]
    cs, gs = torch.cuda.Stream(), torch.cuda.Stream()
    frame_done = [torch.cuda.Event() for _ in range(2)]
    gather_done = [torch.cuda.Event() for _ in range(2)]

    def run(count, overlapped):
        for i in range(count):
            b = i % 2 if overlapped else 0
            if overlapped:
                cs.wait_event(gather_done[b])                       # buffer b is free again (frame i-2 gathered)
            dev.bind_displacement(outs[b].data_ptr())
            dev.frame(i / 60.0, stream=cs.cuda_stream)
            if overlapped:
                frame_done[b].record(cs)
                gs.wait_event(frame_done[b])
                with torch.cuda.stream(gs):
                    dist.gather(outs[b], dsts[b], dst=0)
                    gather_done[b].record(gs)
            else:
                with torch.cuda.stream(cs):
                    dist.gather(outs[b], dsts[b], dst=0)
        torch.cuda.synchronize()

    res = {"steps": steps, "bytes_per_peer_per_frame": n * n * 16,
           "collective": "torch.distributed.gather (RCCL send/recv group), one per frame"}
    for name, overlapped in (("ordered", False), ("overlapped", True)):
        for e in gather_done:
            e.record(gs)
        run(3, overlapped)
        dist.barrier()
        t0 = time.perf_counter()
        run(steps, overlapped)
        ms = (time.perf_counter() - t0) * 1000.0
        dist.barrier()
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        res[name] = {"ms_per_step": ms / steps, "frames_per_s": n_gpus * 1000.0 * steps / ms,
                     "root_ingest_GBps": (n_gpus - 1) * n * n * 16 / (ms / steps) / 1e6}
    dev.bind_displacement(None)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=4096, help="tile edge (power of two, 256..8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="also time frames followed by an RCCL gather of every tile's RGBA map to rank 0 "
                         "(BASELINE config 4; reported separately, never part of `value`)")
    ap.add_argument("--gather-timeout", type=float, default=180.0, help="seconds before a stuck --gather leg is abandoned")
    ap.add_argument("--profile-frames", type=int, default=20, help="frames averaged for the per-kernel durations")
    args = ap.parse_args()

    # The JSON line must be the only thing on stdout.  Native libraries (RCCL's banner and WARN lines,
    # written from its own threads) print to fd 1, so keep a private handle on the real stdout and
    # point fd 1 at stderr for the rest of the process.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # OCEAN_BENCH_FORCE_DIST=1 exercises the torch.distributed / RCCL plumbing at world size 1 (1-GPU boxes)
    if world > 1 or os.environ.get("OCEAN_BENCH_FORCE_DIST") == "1":
        # torch first: libocean_hip.so then binds to the HIP runtime torch loaded (same soname)
        import torch
        import torch.distributed as dist
        os.environ["NCCL_DEBUG"] = os.environ.get("OCEAN_NCCL_DEBUG", "WARN")   # keep RCCL's banner off stdout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)

    import gfx_ocean_amd as g
    n = args.n
    h0, omega = g.synth.make_inputs(n, seed=tile_seed(n, rank))
    dev = g.OceanDevice(n, device_ordinal=local_rank)
    dev.upload_spectrum(h0, omega)

    def barrier():
        dev.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for i in range(args.warmup):
        dev.frame(i / 60.0)
    barrier()
    t0 = time.perf_counter()
    event_ms = dev.time_frames(args.steps, t0=0.0, dt=1.0 / 60.0)   # K frames between two HIP events + sync
    dev.sync()
    wall_ms = (time.perf_counter() - t0) * 1000.0
    barrier()

    if dist is not None:
        import torch
        t = torch.tensor([wall_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        all_ms = [float(t.item())]
    else:
        all_ms = [wall_ms]
    agg = aggregate(all_ms, n_gpus, args.steps)

    # per-kernel durations, live, HIP events on the stream the kernels run on
    acc = {}
    for i in range(args.profile_frames):
        for name, ms in dev.profile_frame(i / 60.0):
            acc[name] = acc.get(name, 0.0) + ms
    kernels = []
    for name, total in acc.items():
        avg_ms = total / args.profile_frames
        b = KERNEL_BYTES_PER_TEXEL[pass_of(name)] * n * n
        rec = {"name": name, "avg_ms": avg_ms, "algorithmic_bytes": b, "GBps": b / avg_ms / 1e6}
        if "half" in name:
            hb = HALF_BYTES_PER_TEXEL[pass_of(name)] * n * n
            rec.update({"half_spectrum_bytes": hb, "half_spectrum_GBps": hb / avg_ms / 1e6})
        kernels.append(rec)
    dom = max(kernels, key=lambda k: k["avg_ms"])
    roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": measured_traffic(n, dom["name"]),
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_launch_ms": dom["avg_ms"],
                "kernels": kernels,
                "frame": {"algorithmic_bytes": 76.0 * n * n, "GBps": 76.0 * n * n / (event_ms / args.steps) / 1e6,
                          "frac": 76.0 * n * n / (event_ms / args.steps) / 1e6 / HBM_PEAK_GBS}}

    gather = None
    if args.gather and dist is not None:
        import threading
        import torch
        # a collective that never completes must not take the measured line with it
        watchdog = threading.Timer(args.gather_timeout, lambda: os._exit(3))
        watchdog.daemon = True
        watchdog.start()
        try:
            gather = gather_leg(dev, dist, torch, n, n_gpus, rank, max(1, min(args.steps, 50)))
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the main metric
            gather = {"error": f"{type(e).__name__}: {e}"}
        watchdog.cancel()

    if rank == 0:
        line = {
            "metric": "ocean frames/sec (propagate + 3x 2-D iFFT + correction, one NxN tile per GPU)",
            "value": agg["value"], "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": agg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"N={n} tile per GPU, fused frame (2 launches: propagate + column pass, row pass + "
                                   f"correction), height+disp_x+disp_z; algorithmic bytes counted as for 3 complex "
                                   f"iFFTs/frame (76 B/texel, SURVEY 8d), computed with the half-spectrum real-output "
                                   f"algorithm (54 B/texel actually moved); seed N+rank",
                       "n": n, "tiles": n_gpus, "parallelism": f"tile-parallel x{n_gpus}, no data-path collective",
                       "gpu_event_ms_per_step": event_ms / args.steps},
            "roofline": roofline,
        }
        if gather is not None:
            line["gather"] = gather
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, h0, omega)
        print(json.dumps(line), file=json_out, flush=True)
    dev.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
