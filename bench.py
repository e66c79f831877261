#!/usr/bin/env python3
"""bench.py -- ocean frames/sec + achieved HBM GB/s on N x N tiles, one tile per MI355X.

A "step" is one frame of the hot path on one tile per GPU: spectrum propagate -> 2-D inverse FFT of
the three fields -> sign correction + RGBA pack (the fused 2-launch path, `ocean_frame`), with h0 and
omega already resident in HBM.  Tiles are independent, so N GPUs = N tiles per step
and no data-path collective ("weak" scaling).  One JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 4096] [--spectrum f32|f16]

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (one process
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for each, 127.0.0.1 rendezvous); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it uses the ranks it is given.
Either way rank 0 prints exactly one line with "n_gpus": N.

For N > 1 the line also carries BASELINE config 4's final gather ("gather": tile maps collected on rank 0 with
one RCCL collective per frame, ordered and double-buffered/overlapped; never part of `value`).

`--plumbing` (tests only): the same launcher, rendezvous, barrier, MAX-reduce and gather bookkeeping with the
gloo backend on CPU tensors and NO device work -- `value` is null and the line says "plumbing": true.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# Bytes per texel (of the N^2 frame) each fused kernel of the shipped half-spectrum algorithm has to move
#  -- what `roofline.achieved` is computed from:
#   pass1: read h0 + omega 4, write the half-spectrum intermediate 12
#          h0: 8 (4 when the spectrum is stored as fp16 pairs) where every spectrum line is requested once -- N >= 4096,
#          the LDS-DMA loader (round 4; round 3: N = 8192, the LDS hand-over); 10 (5) below, where a workgroup of P columns
#          asks for lines x, x-1, N-x, N-1-x: 2P + 2 distinct lines per 2P (cache-resident sizes)
#   pass2: read 12, write RGBA32F 16
# (Round 3 priced every N at 10 B of h0: at 8192 the "algorithmic" bytes then EXCEEDED the counted ones, VERDICT r03 weak #2;
#  tests/test_dist.py checks algorithmic <= counted against every committed PMC file.)
#   with the normal field (--normals, BASELINE config 3): pass 2 also writes the source channel as a dense fp32 plane (+4), and
#   the normal-field kernel reads that plane (4) and writes float4 normals (16): 20 moved; its ALGORITHMIC minimum -- what
#   `frac` of that kernel is priced on -- is 4 read + 12 written (NORMALS_ALGORITHMIC_BYTES_PER_TEXEL)
def moved_bytes_per_texel(n, spectrum="f32", intermediate="f32", normals=False):
    h0 = (8.0 if n >= 4096 else 10.0) * (0.5 if spectrum == "f16" else 1.0)
    inter = 6.0 if intermediate == "bfp16" else 12.0     # bfp16 (opt-in, N = 8192): int16 pairs (+ 3 MB of block scales, not counted)
    moved = {"pass1": h0 + 4.0 + inter, "pass2": inter + 16.0}
    if normals:
        moved["pass2"] += 4.0
        moved["normals"] = 20.0
    return moved


NORMALS_ALGORITHMIC_BYTES_PER_TEXEL = 16.0   # 4 R (one channel) + 12 W (x, y, z)
NORMALS_CONTRACT_BYTES_PER_TEXEL = 32.0      # "a cheap 5th streaming kernel: +16 B read, +12/16 B write per texel"
NORMALS_CHANNELS = {"disp_x": 0, "height": 1, "disp_z": 2}   # disp_x = what the reference differentiates (quirk Q5)


# The contract accounting of SURVEY.md 8d (three complex 2-D transforms per frame, B_frame = 76 N^2; 72 N^2 with
# an fp16-stored spectrum): reported next to the real bytes as `contract_*`, never as `achieved`.
#   pass1: read h0 8 (4) + omega 4, write 3 complex fields 24; pass2: read 24, write RGBA32F 16
CONTRACT_BYTES_PER_TEXEL = {"f32": {"pass1": 36.0, "pass2": 40.0}, "f16": {"pass1": 32.0, "pass2": 40.0}}
# ... and with a 16-bit intermediate of three complex fields (SURVEY 8d "B_frame16": 4 + 4 | 12 | 12 | 16 with fp16 h0)
CONTRACT16_BYTES_PER_TEXEL = {"f32": {"pass1": 24.0, "pass2": 28.0}, "f16": {"pass1": 20.0, "pass2": 28.0}}


def pass_of(kernel_name):
    return "pass1" if "pass1" in kernel_name else ("normals" if "normals" in kernel_name else "pass2")


def aggregate(values_ms, n_gpus, steps):
    """Whole-job throughput from per-rank wall times: tiles processed / slowest rank's time."""
    worst_ms = max(values_ms)
    ms_per_step = worst_ms / steps
    return {"ms_per_step": ms_per_step, "value": n_gpus * 1000.0 / ms_per_step}


def traffic_suffix(spectrum="f32", intermediate="f32", normals=False):
    return ("" if spectrum == "f32" else "_f16") + ("" if intermediate == "f32" else "_bfp16") + ("_normals" if normals else "")


def measured_traffic(n, kernel_name, spectrum="f32", intermediate="f32", normals=False):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command, never inside the timed bench;
    gfx950 correction 2*FETCH_SIZE + WRITE_SIZE -- DESIGN.md 7).  None if no pass exists for this N."""
    suffix = traffic_suffix(spectrum, intermediate, normals)
    path = os.path.join(ROOT, "profiles", f"hbm_traffic_n{n}{suffix}.json")
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    v = rec.get(kernel_name)
    return v["hbm_bytes"] if v else None


def traffic_source(n, spectrum="f32", intermediate="f32", normals=False):
    """Where `roofline.traffic` comes from: never from the timed run (counters perturb timing and need rocprofv3
    around the process) but from a committed PMC pass of this same command."""
    suffix = traffic_suffix(spectrum, intermediate, normals)
    rel = os.path.join("profiles", f"hbm_traffic_n{n}{suffix}.json")
    try:
        with open(os.path.join(ROOT, rel)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return {"file": rel, "run": rec.get("run"), "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, "
            "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, DESIGN.md 7); NOT measured in this run"}


def percentiles(values_ms):
    """median / p10 / p90 / min of a list of per-frame times (nearest-rank on the sorted list)."""
    v = sorted(values_ms)
    pick = lambda q: v[min(len(v) - 1, int(q * len(v)))]
    return {"median_ms": pick(0.5), "p10_ms": pick(0.1), "p90_ms": pick(0.9), "min_ms": v[0]}


def rank_device_index(local_rank, visible, env):
    """The HIP ordinal rank `local_rank` of this node drives, or None when it has no GPU.  Normally the launcher leaves every GPU
    visible to every rank and rank r takes device r; a launcher that isolates each rank to ONE device (HIP_ / ROCR_ /
    CUDA_VISIBLE_DEVICES set per process) leaves it a single device, ordinal 0, whatever its LOCAL_RANK."""
    if local_rank < visible:
        return local_rank
    if visible == 1 and any(env.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return 0
    return None


def tile_seed(n, rank):
    return n + rank     # tile r of a multi-GPU run uses seed N + r


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n, h0, omega, budget_s=10.0, max_frames=12):
    """The C restatement of the reference shaders (oracle/, kind "port") timed on this box's host
    cores on a bounded sample of the same workload: whole frames of the same N until ~budget_s.
    The thread count is the best of {all hardware threads, half, 64, 32, 16} on one probe frame each
    (the strided column pass does not scale to 256 SMT threads on a 2-socket box); the 1-thread time of one
    frame is reported beside it when a frame fits the budget (N <= 4096)."""
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
    one = None
    if n <= 4096:
        cc.set_threads(1)
        t1 = time.perf_counter()
        runner.frame(0.25)
        one = time.perf_counter() - t1
        cc.set_threads(best)
    return {"value": frames / dt, "unit": "frames/s", "cores": best, "kind": "port",
            "one_thread_s_per_frame": one, "cpu_model": cpu_model(), "hardware_threads": hw,
            "sample": f"{frames} whole frames at N={n} ({dt:.1f} s), OpenMP over lines with {best} threads "
                      f"(best of probes {{{', '.join(f'{k}: {v:.2f} s' for k, v in probes.items())}}}, "
                      f"{hw} hardware threads, {cpu_model()}), radix-2 Stockham with sincosf per butterfly as in "
                      f"the shaders" + (f"; 1 thread: {one:.2f} s per frame" if one is not None else "")}


# ------------------------------------------------------------------------------------------------------
# Final gather (BASELINE config 4 / SURVEY 8e).  The schedule is written against a tiny runtime interface so
# that the CPU plumbing mode runs the SAME bookkeeping (buffer rotation, event order, collective calls) on
# gloo tensors: GpuRuntime = torch.cuda streams/events + the HIP frame; PlumbingRuntime = in-order no-ops
# that log what was asked of them.
# ------------------------------------------------------------------------------------------------------
class GpuRuntime:
    def __init__(self, torch, dev):
        self.torch, self.dev = torch, dev

    def empty_tile(self, n, channels=4):
        return self.torch.empty((n, n, channels), dtype=self.torch.float32, device="cuda")

    def stream(self, name):
        return self.torch.cuda.Stream()

    def event(self, name):
        return self.torch.cuda.Event()

    def wait(self, stream, event):
        stream.wait_event(event)

    def record(self, event, stream):
        event.record(stream)

    def on(self, stream):
        return self.torch.cuda.stream(stream)

    def frame(self, out, t, stream):
        self.dev.bind_displacement(out.data_ptr())
        self.dev.frame(t, stream=stream.cuda_stream)

    def frame_packed(self, packed, fmt, t, stream):
        """Frame into the context's own RGBA map, then the packed copy the collective ships (12 or 4 B/texel)."""
        self.dev.bind_displacement(None)
        self.dev.frame(t, stream=stream.cuda_stream)
        self.dev.pack_displacement(fmt, packed.data_ptr(), stream=stream.cuda_stream)

    def unbind(self):
        self.dev.bind_displacement(None)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def scalar(self, v):
        return self.torch.tensor([v], dtype=self.torch.float64, device="cuda")


class PlumbingRuntime:
    """CPU stand-in used by `--plumbing` (tests/test_dist.py): no device, no frame; every call is logged."""

    class _Named:
        def __init__(self, name):
            self.name = name

    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def __init__(self, torch, rank):
        self.torch, self.rank, self.log = torch, rank, []

    def empty_tile(self, n, channels=4):
        return self.torch.zeros((n, n, channels), dtype=self.torch.float32)

    def stream(self, name):
        return self._Named(name)

    def event(self, name):
        return self._Named(name)

    def wait(self, stream, event):
        self.log.append(("wait", stream.name, event.name))

    def record(self, event, stream):
        self.log.append(("record", event.name, stream.name))

    def on(self, stream):
        self.log.append(("on", stream.name))
        return self._Ctx()

    def frame(self, out, t, stream):
        out.fill_(float(self.rank + 1))            # a recognisable tile: rank r writes r + 1
        self.log.append(("frame", stream.name, id(out)))

    def frame_packed(self, packed, fmt, t, stream):
        packed.fill_(float(self.rank + 1))
        self.log.append(("frame", stream.name, None))
        self.log.append(("pack", stream.name, fmt, id(packed)))

    def unbind(self):
        pass

    def synchronize(self):
        pass

    def scalar(self, v):
        return self.torch.tensor([v], dtype=self.torch.float64)


GATHER_FORMATS = {"rgba32f": (0, 4), "rgb32f": (1, 3), "height32f": (2, 1)}   # name -> (OCEAN_PACK_*, floats per texel)


def dominant_kernel(kernels):
    """The kernel `roofline` is quoted on: the longest one.  The two passes of a frame run within a few per cent of each other and
    trade places from box to box, so among kernels within 3 % of the longest the one FURTHEST from its roofline is reported (the
    conservative line, and the same kernel from run to run)."""
    longest = max(k["avg_ms"] for k in kernels)
    return min((k for k in kernels if k["avg_ms"] >= 0.97 * longest), key=lambda k: k["frac"])


def gather_leg(rt, dist, n, n_gpus, rank, steps, warm=3, fmt="rgba32f"):
    """Every tile's RGBA map gathered to rank 0 with one collective per frame (root ingest N*N*16 B per peer
    over xGMI).  Two schedules, both reported, neither part of `value`:
    `ordered`    -- frame and collective on one stream;
    `overlapped` -- two output buffers; the collective of frame f runs on a second stream while frame f+1 is
                    computed (the frame is ~0.2 ms, the root's ingest of 7 x 256 MiB ~1.8 ms: the pipeline is
                    gather-bound and the overlap hides the compute, not the other way round)."""
    pack_id, channels = GATHER_FORMATS[fmt]
    # rgba32f: the frame writes straight into the buffer the collective ships (ocean_bind_displacement);
    # rgb32f / height32f: the frame writes the context's own map and ocean_pack_displacement fills the shipped buffer
    # (12 or 4 instead of 16 B/texel over xGMI; the root's ingest is what bounds the with-gather rate)
    outs = [rt.empty_tile(n, channels) for _ in range(2)]
    dsts = [[rt.empty_tile(n, channels) for _ in range(n_gpus)] if rank == 0 else None for _ in range(2)]
    cs, gs = rt.stream("compute"), rt.stream("gather")
    frame_done = [rt.event(f"frame_done{b}") for b in range(2)]
    gather_done = [rt.event(f"gather_done{b}") for b in range(2)]

    def run(count, overlapped):
        for i in range(count):
            b = i % 2 if overlapped else 0
            if overlapped:
                rt.wait(cs, gather_done[b])                         # buffer b is free again (frame i-2 gathered)
            if pack_id == 0:
                rt.frame(outs[b], i / 60.0, cs)
            else:
                rt.frame_packed(outs[b], pack_id, i / 60.0, cs)
            if overlapped:
                rt.record(frame_done[b], cs)
                rt.wait(gs, frame_done[b])
                with rt.on(gs):
                    dist.gather(outs[b], dsts[b], dst=0)
                    rt.record(gather_done[b], gs)
            else:
                with rt.on(cs):
                    dist.gather(outs[b], dsts[b], dst=0)
        rt.synchronize()

    res = {"steps": steps, "format": fmt, "bytes_per_peer_per_frame": n * n * 4 * channels,
           "collective": "torch.distributed.gather (RCCL send/recv group), one per frame"}
    for name, overlapped in (("ordered", False), ("overlapped", True)):
        for e in gather_done:
            rt.record(e, gs)
        run(warm, overlapped)
        dist.barrier()
        t0 = time.perf_counter()
        run(steps, overlapped)
        ms = (time.perf_counter() - t0) * 1000.0
        dist.barrier()
        t = rt.scalar(ms)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        res[name] = {"ms_per_step": ms / steps, "frames_per_s": n_gpus * 1000.0 * steps / ms,
                     "root_ingest_GBps": (n_gpus - 1) * n * n * 4 * channels / (ms / steps) / 1e6}
    rt.unbind()
    if rank == 0:       # what arrived: one scalar per peer tile (checked by the plumbing test; cheap on the GPU)
        res["peer_tile_first_texel"] = [float(d.reshape(-1)[0].item()) for d in dsts[(steps - 1) % 2]]
    return res


# ------------------------------------------------------------------------------------------------------
# Self-launch: `python bench.py --gpus N` without a launcher spawns N ranks of this same script.
# ------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks, argv, timeout_s):
    """One child process per rank (rank r on GPU r); rank 0's stdout is ours, the others only have stderr.
    Returns the worst exit code.  Children are addressed by PID only (never by pattern)."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OCEAN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts (RCCL needs it)
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    deadline = time.time() + timeout_s
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is not None:
                    pending.remove(p)
                    worst = max(worst, abs(rc))
                    if rc != 0:                       # one rank failed: the others would wait on it forever
                        for q in pending:
                            q.terminate()
            if time.time() > deadline:
                for q in pending:
                    q.kill()
                print(f"# bench.py: ranks still running after {timeout_s:.0f} s were killed", file=sys.stderr)
                return 124
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed frames K (default: 200 at N >= 4096, 1000 at 2048, 4000 below: a timed region of >= ~40 ms, against which "
                         "the launch latency of its first frame and the wake-up behind its last -- 30-130 us together -- are nothing)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=4096, help="tile edge (power of two, 256..16384)")
    ap.add_argument("--spectrum", choices=("f32", "f16"), default="f32",
                    help="storage of the initial spectrum in HBM (f16 = BASELINE config 5: scaled fp16 pairs, fp32 arithmetic)")
    ap.add_argument("--intermediate", choices=("f32", "bfp16"), default="f32",
                    help="precision of the intermediate between the two launches: f32 (default, what every parity figure refers to) or "
                         "bfp16 (opt-in, N = 8192: int16 mantissas + one power-of-two scale per 64 x 2 block; ~3e-5 normalised max)")
    ap.add_argument("--normals", choices=("off",) + tuple(NORMALS_CHANNELS), default="off",
                    help="BASELINE config 3 (\"height + displacement + normal\"): every frame is followed, inside the timed region, by the "
                         "normal field of the finished map (shader/ocean.frag:50-66 at texel centres) differentiated from this channel; "
                         "disp_x is what the reference differentiates (quirk Q5)")
    ap.add_argument("--batch", type=int, default=1,
                    help="NOT the headline: K time steps of the tile per launch pair (ocean_frame_batch; one launch pair at N <= 1024, where a "
                         "frame's two launches fill an eighth of the chip).  `value` is then frames/s of ceil(steps / K) batched launches, "
                         "the line says \"batched\": K, and carries the frame-level roofline only")
    ap.add_argument("--batch-tiles", action="store_true",
                    help="with --batch K: the K frames of a launch pair are K DIFFERENT tiles (seeds N + rank + 1000 k; ocean_frame_tiles, "
                         "N <= 1024) instead of K time steps of one tile")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="time frames followed by an RCCL gather of every tile's RGBA map to rank 0 (BASELINE config 4; "
                         "reported separately, never part of `value`).  Default: on when more than one rank runs")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--gather-format", choices=sorted(GATHER_FORMATS), default="rgba32f",
                    help="payload of the gather: the RGBA32F map as the reference's image (16 B/texel), or packed by "
                         "ocean_pack_displacement to (disp_x, height, disp_z) (12) or the height alone (4)")
    ap.add_argument("--gather-steps", type=int, default=30)
    ap.add_argument("--gather-timeout", type=float, default=120.0, help="seconds before a stuck gather leg is abandoned")
    ap.add_argument("--profile-frames", type=int, default=20, help="minimum number of frames of the per-dispatch-event loop behind the timed "
                    "region that yields the per-kernel durations (it runs max(steps, distribution-frames, this) frames)")
    ap.add_argument("--distribution-frames", type=int, default=200,
                    help="frames of the untimed loop behind the timed region that yields config.frame_ms_{median,p10,p90} (SURVEY 8d)")
    ap.add_argument("--ramp-frames", type=int, default=None,
                    help="untimed frames before anything is measured (GPU clock ramp); default: at least 100 and at least --ramp-ms of frames")
    ap.add_argument("--ramp-ms", type=float, default=50.0,
                    help="the clock ramp as a duration: the part needs tens of milliseconds of work to reach its running clocks, which 100 "
                         "frames are at N = 4096 (18 ms) but not at 2048 (6 ms: the 200 timed steps behind them measured 61.1 us per frame "
                         "against 59.9 behind 60 ms of ramp, r05_run27)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launch: seconds before the ranks are killed")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank to the cpus local to its GPU's NUMA node")
    ap.add_argument("--plumbing", action="store_true",
                    help="tests only: launcher + rendezvous + reductions + gather bookkeeping on gloo/CPU, no device work")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.n >= 4096 else (1000 if args.n >= 2048 else 4000)
    if args.batch_tiles and (args.batch < 2 or args.spectrum != "f32" or args.intermediate != "f32"):
        ap.error("--batch-tiles goes with --batch K >= 2 and the fp32 spectrum (a context of several tiles stores fp32 spectra)")

    # More ranks than visible devices: one JSON error line within seconds instead of a rendezvous that times out.  In the
    # process that only launches the ranks, and in a single-GPU run, the count comes from the library's own HIP runtime (no
    # torch, no context); a rank of a distributed run asks torch after importing it (below) -- the library must not bring up
    # its runtime in a process before torch has loaded its own copy.
    def too_few_devices(have, want):
        print(json.dumps({"metric": "ocean frames/sec (propagate + 3x 2-D iFFT + correction, one NxN tile per GPU)", "value": None,
                          "unit": "frames/s", "n_gpus": want, "steps": args.steps, "warmup": args.warmup,
                          "error": f"{want} ranks asked for, {max(have, 0)} HIP device(s) visible"}), flush=True)
        sys.exit(2)

    in_dist_rank = "WORLD_SIZE" in os.environ or os.environ.get("OCEAN_BENCH_FORCE_DIST") == "1"
    if not args.plumbing and not in_dist_rank and os.environ.get("OCEAN_BENCH_SKIP_DEVICE_CHECK") != "1":   # (the switch: tests)
        import gfx_ocean_amd as g0
        try:
            have = g0._lib.device_count()
        except g0.OceanError as e:
            have = -1
            print(f"# bench.py: {e}", file=sys.stderr)
        if have < args.gpus:
            too_few_devices(have, args.gpus)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))

    # The JSON line must be the only thing on stdout.  Native libraries (RCCL's banner and WARN lines,
    # written from its own threads) print to fd 1, so keep a private handle on the real stdout and
    # point fd 1 at stderr for the rest of the process.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.n
    dist = None
    torch = None
    # OCEAN_BENCH_FORCE_DIST=1 exercises the torch.distributed / RCCL plumbing at world size 1 (1-GPU boxes)
    if world > 1 or os.environ.get("OCEAN_BENCH_FORCE_DIST") == "1" or args.plumbing:
        # torch first: libocean_hip.so then binds to the HIP runtime torch loaded (same soname)
        import torch
        import torch.distributed as dist
        os.environ["NCCL_DEBUG"] = os.environ.get("OCEAN_NCCL_DEBUG", "WARN")   # keep RCCL's banner off stdout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.plumbing:
            dist.init_process_group(backend="gloo")
        else:
            device_index = rank_device_index(local_rank, torch.cuda.device_count(), os.environ)
            if device_index is None and os.environ.get("OCEAN_BENCH_SKIP_DEVICE_CHECK") != "1":   # this rank has no GPU
                if rank == 0:
                    sys.stdout = json_out
                    too_few_devices(torch.cuda.device_count(), world)
                sys.exit(2)
            local_rank = device_index if device_index is not None else local_rank     # the HIP ordinal this rank drives from here on
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; the line reports n_gpus={n_gpus}", file=sys.stderr)
    want_gather = (n_gpus > 1) if args.gather is None else args.gather

    import gfx_ocean_amd as g
    seed = tile_seed(n, rank)
    line = None
    # Rank r's host thread(s) on the NUMA node of GPU r (a 2-socket host; at N <= 1024 the frame rate is the host's submit rate,
    # 5-7 us per frame): the cpus sysfs lists as local to the device's PCI function.  --no-pin leaves the placement to the OS.
    affinity = {"pinned": False}
    if not args.plumbing and not args.no_pin:
        bus, node, cpus = g._lib.device_numa(local_rank)
        affinity.update({"pci_bus_id": bus, "numa_node": node})
        if cpus:
            try:
                allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
                if allowed:
                    os.sched_setaffinity(0, allowed)
                    affinity.update({"pinned": True, "cpus": len(allowed)})
            except OSError as e:
                affinity["error"] = str(e)

    if args.plumbing:
        # ---- launcher / rendezvous / reduction / gather bookkeeping only; nothing is measured -----------------
        rt = PlumbingRuntime(torch, rank)
        dist.barrier()
        wall_ms = 1.0 + rank                                   # rank r "took" r + 1 ms: the MAX must pick the last rank
        t = rt.scalar(wall_ms)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seeds = [None] * world
        dist.all_gather_object(seeds, seed)
        gather = gather_leg(rt, dist, min(n, 64), n_gpus, rank, max(1, min(args.gather_steps, 4)), warm=1,
                            fmt=args.gather_format) if want_gather else None
        if rank == 0:
            line = {"metric": "plumbing only (no device work)", "value": None, "unit": "frames/s", "n_gpus": n_gpus,
                    "steps": args.steps, "warmup": args.warmup, "plumbing": True, "max_rank_ms": float(t.item()),
                    "seeds": seeds, "gather": gather, "gather_log": rt.log}
            print(json.dumps(line), file=json_out, flush=True)
        dist.destroy_process_group()
        return

    h0, omega = g.synth.make_inputs(n, seed=seed)
    if args.batch_tiles and args.batch > 1:                          # K independent tiles per launch pair (labelled batched mode)
        dev = g.OceanDevice(n, device_ordinal=local_rank, tiles=args.batch)
        dev.upload_spectrum(h0, omega, tile=0)
        for k in range(1, args.batch):
            dev.upload_spectrum(*g.synth.make_inputs(n, seed=seed + 1000 * k), tile=k)
    else:
        dev = g.OceanDevice(n, device_ordinal=local_rank, flags=g.CTX_FUSED_ONLY)   # the fused frame's buffers only (40 instead of 76-100 B/texel)
        dev.upload_spectrum(h0, omega, spectrum_fp16=(args.spectrum == "f16"))
    if args.intermediate == "bfp16":
        dev.set_intermediate(g.INTER_BFP16)
    with_normals = args.normals != "off"
    if with_normals:
        dev.set_frame_normals(NORMALS_CHANNELS[args.normals])       # from here on a frame is three launches: pass 1, pass 2 (+ plane), normals

    def barrier():
        dev.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    # A cold GPU needs tens of milliseconds of work to reach its running clocks (measured on MI355X: the first ~25
    # frames of a run are ~10 % slower, and with W = 5 the timed K = 20 steps would be measured on the ramp).  So,
    # untimed and BEFORE the timed region: `--ramp-frames` frames, then the W warmup steps.
    if args.ramp_frames is None:                                    # ... as a duration (--ramp-ms): 20 probe frames size it
        probe_ms = dev.time_frames(20, t0=0.0, dt=1.0 / 60.0) / 20.0
        args.ramp_frames = max(100, min(20000, int(args.ramp_ms / max(probe_ms, 1e-4))))
    dev.time_frames(args.ramp_frames, t0=0.0, dt=1.0 / 60.0)        # clock ramp, untimed (see above)
    batched = max(1, args.batch)
    if batched > 1 and with_normals and n > 1024:
        ap.error("--batch carries the normal field at N <= 1024 (above, a batch is K ordinary frames)")
    timed_frames = args.steps
    if batched > 1:
        launches = -(-args.steps // batched)
        timed_frames = launches * batched                           # whole batches: what `value` counts
        dev.time_frame_batch(max(1, -(-args.warmup // batched)), batched)
        barrier()
        t0 = time.perf_counter()
        event_ms = dev.time_frame_batch(launches, batched, t0=0.0, dt=1.0 / 60.0)
        dev.sync()
        wall_ms = (time.perf_counter() - t0) * 1000.0
        barrier()
    else:
        for i in range(args.warmup):
            dev.frame(i / 60.0)
        barrier()
        t0 = time.perf_counter()
        event_ms = dev.time_frames(args.steps, t0=0.0, dt=1.0 / 60.0)   # K frames between two HIP events + sync
        dev.sync()
        wall_ms = (time.perf_counter() - t0) * 1000.0
        barrier()

    # Distribution ("median + p10/p90"), AFTER the timed region so that `value` is untouched: a plain back-to-back
    # loop with one stream event every 10 frames (ocean_time_frame_batches: the frame), and a loop whose dispatches carry their
    # own begin/end events (ocean_frame_times: the two kernels; those launches leave ~5 % more gaps, so not the frame).
    # The two kernels' durations come from the second loop as well: back-to-back frames, begin/end events bound to every
    # dispatch on the stream it runs on.  [Rounds 1-3 profiled single frames with a synchronisation behind each: the clocks
    # drop between such frames and pass 1 read 94-100 us where the frame loop -- and rocprofv3's steady-state average of the
    # same command -- has 87-90 (r04_run19/20); the same synchronisations also slowed the timed region that followed.]
    per_batch = 10
    dist_frames = min(4096, max(args.steps, args.distribution_frames, args.profile_frames))   # (the C API's bound on both loops)
    batch_ms = dev.time_frame_batches(min(4096, max(2, dist_frames // per_batch)), per_batch)
    p1_ms, p2_ms, nrm_ms, _ = dev.frame_times_ex(dist_frames)
    acc = {"k_half_pass1": sum(p1_ms) / len(p1_ms), "k_half_pass2": sum(p2_ms) / len(p2_ms)}
    spread = {"frames": len(batch_ms) * per_batch, "frames_per_batch": per_batch, "frame": percentiles([b / per_batch for b in batch_ms]),
              "pass1": percentiles(p1_ms), "pass2": percentiles(p2_ms)}
    if with_normals:
        acc["k_normals_plane"] = sum(nrm_ms) / len(nrm_ms)
        spread["normals"] = percentiles(nrm_ms)

    if dist is not None:
        t = torch.tensor([wall_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        all_ms = [float(t.item())]
    else:
        all_ms = [wall_ms]
    agg = aggregate(all_ms, n_gpus, timed_frames)

    moved, contract = moved_bytes_per_texel(n, args.spectrum, args.intermediate, with_normals), dict(CONTRACT_BYTES_PER_TEXEL[args.spectrum])
    if args.intermediate == "bfp16":
        contract = dict(CONTRACT16_BYTES_PER_TEXEL[args.spectrum])
    if with_normals:
        contract["normals"] = NORMALS_CONTRACT_BYTES_PER_TEXEL
    kernels = []
    for name, avg_ms in acc.items():
        b = moved[pass_of(name)] * n * n
        cb = contract[pass_of(name)] * n * n
        # `name` is the pass; the template rocprofv3 lists for it at this size (Launch<N> in csrc/ocean_api.hip)
        device_kernel = name if "normals" in name else name + (("_split" if name.endswith("1") else "_real") if n > 4096 else "")
        rec = {"name": name, "device_kernel": device_kernel, "avg_ms": avg_ms, "algorithmic_bytes": b, "GBps": b / avg_ms / 1e6,
               "frac": b / avg_ms / 1e6 / HBM_PEAK_GBS, "contract_bytes": cb, "contract_GBps": cb / avg_ms / 1e6,
               "contract_frac": cb / avg_ms / 1e6 / HBM_PEAK_GBS,
               "traffic": measured_traffic(n, name, args.spectrum, args.intermediate, with_normals)}
        if "normals" in name:      # priced on the pass's algorithmic minimum (4 R + 12 W); what the kernel moves (4 R + 16 W) beside it
            ab = NORMALS_ALGORITHMIC_BYTES_PER_TEXEL * n * n
            rec.update({"moved_bytes": b, "moved_GBps": rec["GBps"], "algorithmic_bytes": ab, "GBps": ab / avg_ms / 1e6,
                        "frac": ab / avg_ms / 1e6 / HBM_PEAK_GBS})
        kernels.append(rec)
    dom = dominant_kernel(kernels)
    frame_ms = event_ms / timed_frames
    fb, fcb = sum(moved.values()) * n * n, sum(contract.values()) * n * n
    roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"], "traffic": dom["traffic"], "traffic_source": traffic_source(n, args.spectrum, args.intermediate, with_normals),
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_launch_ms": dom["avg_ms"],
                "accounting": f"achieved = bytes the shipped half-spectrum algorithm must move ({moved['pass1']:.0f} + "
                              f"{moved['pass2']:.0f} B/texel" + (f" + {moved['normals']:.0f} for the normal field, whose kernel is priced "
                              f"on its algorithmic minimum of {NORMALS_ALGORITHMIC_BYTES_PER_TEXEL:.0f}" if with_normals else "") +
                              f") / kernel time; contract_* = SURVEY 8d's three-complex-"
                              f"transform accounting ({contract['pass1']:.0f} + {contract['pass2']:.0f} B/texel" +
                              (f" + {contract['normals']:.0f}, SURVEY 8f #1" if with_normals else "") + ") / the same time",
                "contract_achieved": dom["contract_GBps"], "contract_frac": dom["contract_frac"],
                "kernels": kernels,
                "frame": {"algorithmic_bytes": fb, "GBps": fb / frame_ms / 1e6, "frac": fb / frame_ms / 1e6 / HBM_PEAK_GBS,
                          "contract_bytes": fcb, "contract_GBps": fcb / frame_ms / 1e6,
                          "contract_frac": fcb / frame_ms / 1e6 / HBM_PEAK_GBS}}

    if rank == 0:
        spec_txt = "fp32 spectrum" if args.spectrum == "f32" else "fp16-stored spectrum (scaled pairs), fp32 arithmetic and intermediate"
        if args.intermediate == "bfp16":
            spec_txt = spec_txt.replace(" and intermediate", "") + ("; OPT-IN PRECISION MODE: 16-bit block-floating intermediate (int16 mantissas, "
                                                                    "one power-of-two scale per 64 x 2 block; ~3e-5 normalised max against the fp32 intermediate)")
        line = {
            "metric": "ocean frames/sec (propagate + 3x 2-D iFFT + correction, one NxN tile per GPU)",
            "value": agg["value"], "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": agg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"N={n} tile per GPU, {spec_txt}, fused frame (" +
                                   (f"3 launches: propagate + column pass, row pass + correction, normal field from {args.normals}), "
                                    f"height+disp_x+disp_z+normal" if with_normals else
                                    "2 launches: propagate + column pass, row pass + correction), height+disp_x+disp_z") +
                                   f"; half-spectrum real-output algorithm: "
                                   f"{sum(moved.values()):.0f} B/texel moved ({sum(contract.values()):.0f} B/texel on the "
                                   f"three-complex-transform accounting of SURVEY 8d); seed N+rank",
                       "n": n, "spectrum": args.spectrum, "intermediate": args.intermediate, "normals": args.normals, "tiles": n_gpus,
                       "parallelism": f"tile-parallel x{n_gpus}, no data-path collective",
                       "host_affinity": affinity,
                       "gpu_event_ms_per_step": frame_ms,
                       "frame_ms_median": spread["frame"]["median_ms"], "frame_ms_p10": spread["frame"]["p10_ms"],
                       "frame_ms_p90": spread["frame"]["p90_ms"],
                       "frame_time_distribution": dict(spread, method="behind the timed region: `frame` = per-frame time of consecutive 10-frame batches of a "
                                                       "plain back-to-back loop (one stream event per batch, one sync at the end); pass1 / pass2 = per-dispatch "
                                                       "begin/end events of a second loop (ocean_frame_times)"),
                       "effective_warmup_frames": args.ramp_frames + args.warmup,
                       "untimed_before_timed_region": f"{args.ramp_frames} clock-ramp frames + {args.warmup} warmup (`warmup` above is W as "
                                                      f"given; effective_warmup_frames counts everything untimed); the per-kernel durations and "
                                                      f"the frame-time distribution are measured BEHIND the timed region"},
            "roofline": roofline,
        }
        if batched > 1:
            what = "independent tiles per launch pair (ocean_frame_tiles)" if args.batch_tiles else "time steps per launch pair (ocean_frame_batch)"
            line["batched"] = batched
            line["batched_kind"] = "tiles" if args.batch_tiles else "time steps"
            line["metric"] += f" -- BATCHED MODE, not the headline: {batched} {what}"
            line["steps_timed"] = timed_frames
            line["config"]["workload"] += (f"; BATCHED: {batched} time steps of the tile per launch pair, {timed_frames} frames timed as "
                                           f"{timed_frames // batched} batched launches (the per-kernel figures and the frame-time "
                                           f"distribution are those of the unbatched frame)")

    emitted = threading.Event()
    emit_lock = threading.Lock()

    def emit(extra=None):
        """Print the one line (rank 0) exactly once (the watchdog thread may race the main thread here)."""
        with emit_lock:
            if emitted.is_set():
                return
            emitted.set()
        if rank == 0:
            if extra:
                line.update(extra)
            print(json.dumps(line), file=json_out, flush=True)

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(n, h0, omega)

    if want_gather and dist is not None:
        # A collective that never completes must not take the measured line with it: on timeout rank 0 prints the
        # line without the leg and every rank leaves.
        def abandon():
            # the main metric was measured and is printed; the exit status still says that a collective hung
            emit({"gather": {"error": f"abandoned after {args.gather_timeout:.0f} s"}, "gather_abandoned": True})
            json_out.flush()
            os._exit(3)
        watchdog = threading.Timer(args.gather_timeout, abandon)
        watchdog.daemon = True
        watchdog.start()
        try:
            gather = gather_leg(GpuRuntime(torch, dev), dist, n, n_gpus, rank, max(1, args.gather_steps), fmt=args.gather_format)
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the main metric
            gather = {"error": f"{type(e).__name__}: {e}"}
        watchdog.cancel()
        emit({"gather": gather})
    else:
        emit()
    dev.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
