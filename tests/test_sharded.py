"""One N x N tile sharded by row blocks over several ranks (gfx_ocean_amd/sharded.py, csrc/ocean_shard.hip):
propagate + row pass on the rank's rows, ONE all-to-all, column pass + correction on the rank's columns.

CPU tier: the shard kernels run in the host emulation, the all-to-all is gloo with world size 2 (and 4), and the
assembled tile must match the fp64 oracle of the whole frame -- i.e. the decomposition, the send/receive layouts and
the partner-block indexing of propagate are right.  GPU tier: the same through the C ABI on one GPU (world = 1 and,
with OCEAN_BENCH_FORCE_DIST-style single-rank RCCL, the collective call itself)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gfx_ocean_amd as g
from gfx_ocean_amd import sharded
from oracle import ocean_oracle as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import emu, gfx_ocean_amd as g
from gfx_ocean_amd import sharded
from oracle import ocean_oracle as oc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
n, t = int(sys.argv[2]), 1.5
h0, om = g.synth.make_inputs(n, seed=77)            # every rank builds the same tile and keeps its slices
tile = sharded.ShardedTile(emu.EmuShardBackend(n, rank, world), dist)
tile.upload(h0, om)
tile.frame(t)
full = tile.gather_tile()
if rank == 0:
    ref = oc.frame_f64(h0, om, t)
    nmax, rl2 = oc.parity_errors(full[..., :3], ref[..., :3])
    print("RESULT " + json.dumps({"nmax": float(nmax.max()), "rl2": float(rl2.max()), "alpha0": bool(np.all(full[..., 3] == 0)),
                                  "bytes": sharded.exchange_bytes_per_rank(n, world)}))
dist.destroy_process_group()
"""


def test_split_inputs_partner_block_is_the_mirror():
    n, world = 64, 4
    h0 = (np.arange(n * n).reshape(n, n) * (1 + 0.5j)).astype(np.complex64)
    om = np.arange(n * n, dtype=np.float32).reshape(n, n)
    for r in range(world):
        own, partner, o = sharded.split_inputs(h0, om, r, world)
        rows = n // world
        assert np.array_equal(own, h0[r * rows:(r + 1) * rows]) and np.array_equal(o, om[r * rows:(r + 1) * rows])
        # texel (ly, x) of the block pairs with (N-1-gy, N-1-x) = the partner block read backwards (propagate.comp:48)
        gy = r * rows + 3
        assert partner[::-1, ::-1][3, 5] == h0[n - 1 - gy, n - 1 - 5]
    assert sharded.exchange_bytes_per_rank(16384, 8) == 3 * 2048 * 16384 * 8


@pytest.mark.parametrize("world,n", [(2, 512), (4, 512)])
def test_sharded_tile_matches_the_whole_frame_oracle(tmp_path, world, n):
    worker = tmp_path / "worker.py"
    worker.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), str(worker), ROOT, str(n)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["nmax"] <= 1e-4 and r["rl2"] <= 1e-4 and r["alpha0"], r       # north_star tolerance; expect ~1e-6
    assert r["nmax"] < 1e-5
    assert r["bytes"] == 3 * (n // world) * n * 8


def test_single_rank_shard_is_the_whole_tile():
    """world = 1 degenerates to rows pass + column pass on one rank, no collective."""
    import emu
    n, t = 512, 0.75
    h0, om = g.synth.make_inputs(n, seed=5)
    tile = sharded.ShardedTile(emu.EmuShardBackend(n, 0, 1))
    tile.upload(h0, om)
    tile.frame(t)
    nmax, rl2 = oc.parity_errors(tile.gather_tile()[..., :3], oc.frame_f64(h0, om, t)[..., :3])
    assert nmax.max() < 1e-5 and rl2.max() < 1e-5


# The GPU checks run in their own process with torch imported FIRST: torch bundles a HIP runtime with the same soname
# as /opt/rocm's, and device pointers are only meaningful inside the runtime that made them.
_GPU_WORLD1 = r"""
import sys
import torch
sys.path.insert(0, sys.argv[1])
import numpy as np
import gfx_ocean_amd as g
from gfx_ocean_amd import sharded
from oracle import ocean_oracle as oc
n = int(sys.argv[2])
h0, om = g.synth.make_inputs(n, seed=9)
tile = sharded.ShardedTile(sharded.HipShardBackend(n, 0, 1))
tile.upload(h0, om)
tile.frame(2.25)
if n <= 8192:                                   # every texel against the ORACLE (not against ocean_frame: HIP vs HIP proves nothing)
    got = tile.gather_tile()
    if n <= 1024:
        want = oc.frame_f64(h0, om, 2.25)
    else:
        from oracle import c_oracle as cc
        cc.build(); cc.set_threads(min(32, cc.max_threads()))
        want = cc.FrameRunner(h0, om).frame(2.25)
    nmax, rl2 = oc.parity_errors(got[..., :3], want[..., :3])
    assert nmax.max() < 2e-5 and rl2.max() < 2e-5 and np.all(got[..., 3] == 0.0), (nmax, rl2)
else:                                           # 16384 exists only sharded: sampled texels, direct fp64 2-D sums
    out = tile.result()                         # [x, y, 4]
    H, DX, DZ = oc.propagate_f64(h0, om, 2.25)
    k = np.arange(n)
    scale = np.abs(out[..., :3]).max((0, 1))
    for (x, y) in [(0, 0), (n // 2 + 3, n // 3), (n - 1, n - 1)]:
        ey, ex = np.exp(2j * np.pi * k * y / n), np.exp(2j * np.pi * k * x / n)
        sgn = -1.0 if (x + y) % 2 == 0 else 1.0
        ref = np.array([(ey @ (F @ ex)).real for F in (DX, H, DZ)]) * sgn
        assert np.all(np.abs(out[x, y, :3] - ref) <= 1e-4 * scale), (x, y, out[x, y, :3], ref)
tile.b.destroy()
print("SHARD_GPU_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 4096, 16384])
def test_gpu_shard_abi_on_one_gpu(n):
    """The C ABI of the sharded tile on one GPU (world = 1) through torch memory and streams: every texel against the
    oracle (fp64 at 512, the C restatement of the shaders at 4096);
    at N = 16384 -- a size only the sharded path supports: one 16384-point line is a whole 1024-thread workgroup --
    sampled texels against a direct fp64 evaluation of the 2-D sum."""
    p = subprocess.run([sys.executable, "-c", _GPU_WORLD1, ROOT, str(n)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "SHARD_GPU_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def _loopback_frame(n, world, t, h0, om):
    """Ranks 0 .. world-1 of ONE tile, all on device 0: the real HIP kernels with the real partner-block indexing and
    send / receive layouts; the all-to-all is done by hand with device-to-device copies
    (recv of rank r, slot src  <-  send of rank src, slot r).  Returns the assembled tile [y, x, 4]."""
    import ctypes
    from hipmem import DeviceBuffer
    from gfx_ocean_amd._lib import PropagateLocalsC, load_library
    lib = load_library()
    rows = n // world
    slot = 3 * rows * rows * 8                                   # [field][row][column] complex fp32 of one (src, dest) pair
    shards, sends, recvs, outs = [], [], [], []

    def check(st, h=None):
        assert st == 0, (st, (lib.ocean_shard_last_error(h) or b"").decode())
    try:
        for r in range(world):
            h = ctypes.c_void_p()
            check(lib.ocean_shard_create(0, n, r, world, ctypes.byref(h)))
            shards.append(h)
            own, partner, o = sharded.split_inputs(h0, om, r, world)
            check(lib.ocean_shard_upload(h, own.ctypes.data, partner.ctypes.data, o.ctypes.data), h)
            sends.append(DeviceBuffer(world * slot))
            recvs.append(DeviceBuffer(world * slot))
            outs.append(DeviceBuffer(rows * n * 16))
        loc = PropagateLocalsC(float(t), int(n), 1000.0)
        for r in range(world):
            check(lib.ocean_shard_rows(shards[r], ctypes.byref(loc), sends[r].ptr, None), shards[r])
        for r in range(world):
            check(lib.ocean_shard_sync(shards[r]), shards[r])
        for r in range(world):                                   # the all-to-all, by hand
            for src in range(world):
                recvs[r].copy_from_device(sends[src].ptr + r * slot, slot, offset=src * slot)
        for r in range(world):
            check(lib.ocean_shard_cols(shards[r], recvs[r].ptr, outs[r].ptr, None), shards[r])
        parts = []
        for r in range(world):
            check(lib.ocean_shard_sync(shards[r]), shards[r])
            parts.append(outs[r].to_host(np.float32).reshape(rows, n, 4))      # [x - x0, y, 4]
        return np.ascontiguousarray(np.concatenate(parts, axis=0).transpose(1, 0, 2))
    finally:
        for h in shards:
            lib.ocean_shard_destroy(h)
        for b in sends + recvs + outs:
            b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n,world", [(512, 2), (512, 4), (2048, 8), (4096, 2), (4096, 4)])
def test_gpu_shard_multi_rank_on_one_device(n, world):
    """nothing the HIP shard kernels do had run with world > 1.  One GPU is enough: every rank of
    a world-2 / 4 / 8 tile lives on device 0 and the exchange is permuted by hand; the assembled tile is checked
    against the oracle (fp64 closed form at N <= 2048, every texel of the C restatement of the shaders at 4096)."""
    t = 1.5
    h0, om = g.synth.make_inputs(n, seed=31 + world)
    got = _loopback_frame(n, world, t, h0, om)
    if n <= 2048:
        want = oc.frame_f64(h0, om, t)
    else:
        from oracle import c_oracle as cc
        cc.build()
        cc.set_threads(min(32, cc.max_threads()))
        want = cc.FrameRunner(h0, om).frame(t)
    nmax, rl2 = oc.parity_errors(got[..., :3], want[..., :3])
    assert nmax.max() <= 1e-4 and rl2.max() <= 1e-4, (nmax, rl2)          # north_star tolerance
    assert nmax.max() < 2e-5 and np.all(got[..., 3] == 0.0)                # two fp32 paths: a few 1e-6


@pytest.mark.gpu
def test_gpu_shard_state_errors():
    """ocean_shard_cols before any upload is a state error, not a transform of uninitialised rows."""
    import ctypes
    from hipmem import DeviceBuffer
    from gfx_ocean_amd._lib import load_library
    lib = load_library()
    h = ctypes.c_void_p()
    assert lib.ocean_shard_create(0, 512, 0, 1, ctypes.byref(h)) == 0
    buf, out = DeviceBuffer(3 * 512 * 512 * 8), DeviceBuffer(512 * 512 * 16)
    try:
        assert lib.ocean_shard_cols(h, buf.ptr, out.ptr, None) == -5       # OCEAN_E_STATE
    finally:
        lib.ocean_shard_destroy(h)
        buf.free(); out.free()


# ----------------------------------------------------------------------------------------------------------------------
# Second generation: the fused half-spectrum frame, sharded (ocean_tile_pass1 / ocean_tile_pass2, FusedShardedTile)
# ----------------------------------------------------------------------------------------------------------------------
_WORKER2 = _WORKER.replace("sharded.ShardedTile(emu.EmuShardBackend(n, rank, world), dist)",
                          "sharded.FusedShardedTile(emu.EmuTileBackend(n, rank, world, parts=int(sys.argv[3])), dist)") \
                  .replace("sharded.exchange_bytes_per_rank(n, world)", "sharded.fused_exchange_bytes_per_rank(n, world)")


@pytest.mark.parametrize("world,n,parts", [(2, 512, 1), (4, 256, 1), (2, 256, 2)])
def test_fused_sharded_tile_matches_the_whole_frame_oracle(tmp_path, world, n, parts):
    """World 2 and 4 over gloo with the fused kernels in the host emulation: column blocks of the half spectrum, ONE
    all-to-all of half the volume, row blocks out -- against the fp64 oracle of the whole frame."""
    assert "FusedShardedTile" in _WORKER2
    worker = tmp_path / "worker2.py"
    worker.write_text(_WORKER2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29560 + world + 8 * parts), str(worker), ROOT, str(n), str(parts)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["nmax"] <= 1e-4 and r["rl2"] <= 1e-4 and r["alpha0"], r
    assert r["nmax"] < 1e-5
    assert r["bytes"] == 3 * (n // 2 // world) * n * 8 == sharded.exchange_bytes_per_rank(n, world) // 2


def test_fused_sharded_single_rank_and_loopback_in_the_emulation():
    """world = 1 is the fused frame itself; world = 2 with both ranks in one process and the exchange permuted by hand
    (the shape of the GPU test below)."""
    import emu
    n, t = 256, 0.75
    h0, om = g.synth.make_inputs(n, seed=5)
    ref = oc.frame_f64(h0, om, t)
    tile = sharded.FusedShardedTile(emu.EmuTileBackend(n, 0, 1))
    tile.upload(h0, om)
    tile.frame(t)
    nmax, rl2 = oc.parity_errors(tile.gather_tile()[..., :3], ref[..., :3])
    assert nmax.max() < 1e-5 and rl2.max() < 1e-5
    world = 2
    for parts in (1, 2, 4):                                       # the exchange in 1, 2, 4 pipelined pieces
        backs = [emu.EmuTileBackend(n, r, world, parts=parts) for r in range(world)]
        sends = [b.alloc_exchange() for b in backs]               # [part][dest][message]
        recvs = [b.alloc_exchange() for b in backs]               # [part][src][message]
        outs = [b.alloc_out() for b in backs]
        for b, s_ in zip(backs, sends):
            b.upload(h0, om)
            for k in range(parts):
                b.pass1(t, 1000.0, s_[k], part=k)
        for r in range(world):
            for src in range(world):
                for k in range(parts):
                    recvs[r][k, src] = sends[src][k, r]
        for b, r_, o in zip(backs, recvs, outs):
            b.pass2(r_, o)
        full = np.concatenate([o.numpy() for o in outs], axis=0)
        nmax, rl2 = oc.parity_errors(full[..., :3], ref[..., :3])
        assert nmax.max() < 1e-5 and rl2.max() < 1e-5 and np.all(full[..., 3] == 0.0), parts
    with pytest.raises(g.OceanError):
        sharded.FusedShardedTile(backs[0])                       # world 2 without a process group


def test_fused_sharded_real_output_rows_in_the_emulation():
    """The kernels ocean_tile_pass1 / ocean_tile_pass2 launch at N >= 8192 (k_half_pass1_split with a column-group offset,
    k_half_pass2_real<SHARD>; column-major chunks) at a size the emulation can run: both ranks of a world-2 tile in one
    process, the exchange in 1 and 2 pieces."""
    import emu
    n, t, world = 512, 0.75, 2
    h0, om = g.synth.make_inputs(n, seed=6)
    ref = oc.frame_f64(h0, om, t)
    for parts in (1, 2):
        backs = [emu.EmuTileBackend(n, r, world, psel=22, parts=parts) for r in range(world)]
        sends = [b.alloc_exchange() for b in backs]
        recvs = [b.alloc_exchange() for b in backs]
        outs = [b.alloc_out() for b in backs]
        for b, s_ in zip(backs, sends):
            b.upload(h0, om)
            for k in range(parts):
                b.pass1(t, 1000.0, s_[k], part=k)
        for r in range(world):
            for src in range(world):
                for k in range(parts):
                    recvs[r][k, src] = sends[src][k, r]
        for b, r_, o in zip(backs, recvs, outs):
            b.pass2(r_, o)
        full = np.concatenate([o.numpy() for o in outs], axis=0)
        nmax, rl2 = oc.parity_errors(full[..., :3], ref[..., :3])
        assert nmax.max() < 1e-5 and rl2.max() < 1e-5 and np.all(full[..., 3] == 0.0), parts


def _fused_loopback_frame(n, world, t, h0, om, f16=False, parts=1):
    """All ranks of ONE fused sharded tile on device 0, one context (the tile entry points keep no per-rank state); the
    all-to-all by hand with device-to-device copies.  Returns the assembled tile [y, x, 4]."""
    import ctypes
    from hipmem import DeviceBuffer
    from gfx_ocean_amd._lib import PropagateLocalsC, load_library
    lib = load_library()
    d = g.OceanDevice(n, flags=0 if f16 else g.CTX_TILE_RANK)     # (the f16 case reads the dequantised spectrum back: a full context)
    bufs = []
    try:
        d.upload_spectrum(h0, om, spectrum_fp16=f16)
        nbytes = int(lib.ocean_tile_exchange_bytes(d._ctx, world))
        assert nbytes == sharded.fused_exchange_bytes_per_rank(n, world)
        slot = nbytes // world // parts                           # one (src, dest, part) message
        rows = n // world
        sends = [DeviceBuffer(nbytes) for _ in range(world)]
        recvs = [DeviceBuffer(nbytes) for _ in range(world)]
        outs = [DeviceBuffer(rows * n * 16) for _ in range(world)]
        bufs = sends + recvs + outs
        for b in sends + recvs:
            b.fill(0xFF)                                          # NaN patterns: an element nobody wrote shows up in the result
        loc = PropagateLocalsC(float(t), int(n), 1000.0)
        for r in range(world):
            for k in range(parts):                                # send buffer of rank r: [part][dest][message]
                d._check(lib.ocean_tile_pass1(d._ctx, ctypes.byref(loc), r, world, k, parts, sends[r].ptr + k * world * slot, None))
        d.sync()
        for r in range(world):                                    # receive buffer of rank r: [part][src][message]
            for src in range(world):
                for k in range(parts):
                    recvs[r].copy_from_device(sends[src].ptr + (k * world + r) * slot, slot, offset=(k * world + src) * slot)
        for r in range(world):
            d._check(lib.ocean_tile_pass2(d._ctx, r, world, parts, recvs[r].ptr, outs[r].ptr, None))
        d.sync()
        return np.concatenate([o.to_host(np.float32).reshape(rows, n, 4) for o in outs], axis=0), (d.read_spectrum() if f16 else None)
    finally:
        for b in bufs:
            b.free()
        d.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,world,f16", [(256, 2, False), (512, 2, False), (512, 4, False), (1024, 8, False), (2048, 8, False),
                                         (4096, 1, False), (4096, 2, False), (4096, 8, False), (8192, 2, True), (8192, 4, False)])
def test_gpu_fused_shard_multi_rank_on_one_device(n, world, f16):
    """The fused sharded tile with the REAL kernels at world 1 / 2 / 4 / 8 on one GPU (every size the fused frame
    supports, fp16-stored spectrum included): assembled tile against the oracle -- fp64 closed form at N <= 2048, every
    texel of the C restatement of the shaders above."""
    t = 1.5
    h0, om = g.synth.make_inputs(n, seed=51 + world)
    got, deq = _fused_loopback_frame(n, world, t, h0, om, f16)
    src = deq if f16 else h0
    if n <= 2048:
        want = oc.frame_f64(src, om, t)
    else:
        from oracle import c_oracle as cc
        cc.build()
        cc.set_threads(min(32, cc.max_threads()))
        want = cc.FrameRunner(src, om).frame(t)
    nmax, rl2 = oc.parity_errors(got[..., :3], want[..., :3])
    assert nmax.max() <= 1e-4 and rl2.max() <= 1e-4, (nmax, rl2)          # north_star tolerance
    assert nmax.max() < 2e-5 and np.all(got[..., 3] == 0.0)


@pytest.mark.gpu
def test_gpu_fused_shard_16384_every_rank_against_the_c_oracle():
    """SURVEY 8f #4 / VERDICT r03 #4: N = 16384 -- the size that motivates sharding one transform -- on the FUSED
    half-spectrum path (rounds 2-3 ran it on the staged row-block kernels at 220 B/texel): a 16384-point line as two
    interleaved 8192-point transforms with the last radix-2 step at read-out, one column per pass-1 workgroup (two
    sub-lines fill the LDS).  EVERY texel of the tile assembled from every rank of world 1, 2, 4 and 8 (all run on this
    one device, the exchange by device copies) against the C restatement of the shaders, computed once (~1 min of host
    time); the unsharded ocean_frame must be the same bits, and its time is printed (target <= 4 ms; 11.8 ms on the staged
    row-block path, profiles/r03_run14_shard_bench_world1.jsonl)."""
    from oracle import c_oracle as cc
    n, t = 16384, 1.5
    h0, om = g.synth.make_inputs(n, seed=16384)
    cc.build()
    cc.set_threads(min(32, cc.max_threads()))
    want = cc.FrameRunner(h0, om).frame(t)[..., :3].copy()
    first = None
    for world in (1, 2, 4, 8):
        got, _ = _fused_loopback_frame(n, world, t, h0, om)
        nmax, rl2 = oc.parity_errors(got[..., :3], want)
        print(f"N = 16384 fused sharded tile, world {world}: normalised max {nmax.max():.2e}, rel-L2 {rl2.max():.2e}")
        assert nmax.max() <= 1e-4 and rl2.max() <= 1e-4, (world, nmax, rl2)      # north_star tolerance
        assert nmax.max() < 2e-5 and np.all(got[..., 3] == 0.0)
        if first is None:
            first = got
        else:
            assert np.array_equal(got, first), world                          # sharding does not change a bit
        del got
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om)
        d.frame(t)
        assert np.array_equal(d.read_displacement(), first)
        d.time_frames(10)
        ms = d.time_frames(20) / 20
        print(f"N = 16384 ocean_frame on one GPU: {ms:.3f} ms per frame ({54 * n * n / ms / 1e6:.0f} GB/s on 54 B/texel)")
    finally:
        d.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,world", [(256, 2), (512, 4), (1024, 2), (2048, 8), (4096, 4), (8192, 8), (16384, 4)])
def test_gpu_a_rank_reads_only_its_two_bands_of_the_static_inputs(n, world):
    """What sharded.tile_rank_lines says a rank's pass 1 reads of h0T / omegaT is ALL it reads: with every other line of the
    inputs poisoned (NaN), every rank's send buffer is bit-identical to the one computed from the clean inputs -- every loader
    (registers at N <= 1024, the LDS-DMA ring above, the split kernels at 8192), the Nyquist column included."""
    import ctypes
    from hipmem import DeviceBuffer
    from gfx_ocean_amd._lib import PropagateLocalsC, load_library
    lib = load_library()
    h0, om = g.synth.make_inputs(n, seed=61)
    loc = PropagateLocalsC(1.75, int(n), 1000.0)
    nbytes = sharded.fused_exchange_bytes_per_rank(n, world)

    def send_buffers(poison):
        out = []
        for r in range(world):
            hp, op = h0, om
            if poison:
                keep = np.zeros(n, bool)
                keep[sharded.tile_rank_lines(n, r, world)] = True
                hp, op = h0.copy(), om.copy()
                hp[:, ~keep] = np.nan + 1j * np.nan               # line x of the transposed inputs = column x of the natural ones
                op[:, ~keep] = np.nan
            d = g.OceanDevice(n, flags=g.CTX_TILE_RANK)
            buf = DeviceBuffer(nbytes)
            try:
                d.upload_spectrum(hp, op)
                buf.fill(0)
                d._check(lib.ocean_tile_pass1(d._ctx, ctypes.byref(loc), r, world, 0, 1, buf.ptr, None))
                d.sync()
                out.append(buf.to_host(np.uint32))
            finally:
                buf.free()
                d.destroy()
        return out

    clean, poisoned = send_buffers(False), send_buffers(True)
    for r in range(world):
        assert not np.isnan(clean[r].view(np.float32)).any()
        assert np.array_equal(clean[r], poisoned[r]), (n, world, r)
    assert len(sharded.tile_rank_lines(n, 1 % world, world)) <= 2 * (n // 2 // world + 1)
    # ... and a band-limited rank context (ocean_context_create_tile_rank) backs just those lines -- rounded to blocks of 32 -- with
    # memory: the same send buffer, a footprint of ~12 / world B/texel, other ranks refused
    import hipmem
    for r in sorted({0, world - 1, world // 2}):
        before = hipmem.free_bytes()
        d = g.OceanDevice.for_tile_rank(n, r, world)
        buf = DeviceBuffer(nbytes)
        try:
            assert lib.ocean_context_flags(d._ctx) == (g.CTX_FUSED_ONLY | g.CTX_TILE_RANK | g.CTX_TILE_BANDS)
            used = before - hipmem.free_bytes() - nbytes
            blocks = len(set(int(x) // 32 for x in sharded.tile_rank_lines(n, r, world)))
            assert used <= blocks * 32 * n * 12 + (24 << 20), (used, blocks)      # (+ rounding of every band to 2 MiB mappings)
            if n >= 2048 and world >= 4:
                assert used < 0.6 * 12 * n * n                    # far from the whole tile's inputs
            d.upload_spectrum(h0, om)
            buf.fill(0)
            d._check(lib.ocean_tile_pass1(d._ctx, ctypes.byref(loc), r, world, 0, 1, buf.ptr, None))
            d.sync()
            assert np.array_equal(buf.to_host(np.uint32), clean[r]), (n, world, r)
            assert lib.ocean_tile_pass1(d._ctx, ctypes.byref(loc), (r + 1) % world, world, 0, 1, buf.ptr, None) == -1
            with pytest.raises(g.OceanError):
                d.upload_spectrum(h0, om, spectrum_fp16=True)
        finally:
            buf.free()
            d.destroy()
        assert abs(hipmem.free_bytes() - before) <= (64 << 20)    # and everything is unmapped and released again


@pytest.mark.gpu
@pytest.mark.parametrize("n,world,parts", [(2048, 4, 1), (2048, 2, 4), (4096, 8, 2), (512, 2, 2), (16384, 2, 2)])
def test_gpu_fused_shard_equals_the_fused_frame_bit_for_bit(n, world, parts):
    """Sharding -- and cutting the exchange into pipelined parts -- must not change a single bit: the same kernels on
    the same columns and rows, only the addresses of the intermediate differ."""
    t = 2.0
    h0, om = g.synth.make_inputs(n, seed=3)
    got, _ = _fused_loopback_frame(n, world, t, h0, om, parts=parts)
    d = g.OceanDevice(n)
    try:
        d.upload_spectrum(h0, om)
        d.frame(t)
        assert np.array_equal(got, d.read_displacement())
    finally:
        d.destroy()


_GPU_FUSED_WORLD1 = r"""
import sys
import torch, torch.distributed as dist, os
sys.path.insert(0, sys.argv[1])
import numpy as np
import gfx_ocean_amd as g
from gfx_ocean_amd import sharded
from oracle import ocean_oracle as oc
n = int(sys.argv[2])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # RCCL, world 1: the collective call itself
h0, om = g.synth.make_inputs(n, seed=9)
for parts in (1, 2, 4):                    # 2, 4: the pipelined exchange -- that many all-to-alls on the communication stream
    be = sharded.HipTileBackend(n, 0, 1, parts=parts)
    assert g.load_library().ocean_context_flags(be.dev._ctx) == (g.CTX_FUSED_ONLY | g.CTX_TILE_RANK | g.CTX_TILE_BANDS)   # this rank's input lines only
    assert be.context_kind == "bands", be.context_kind
    tile = sharded.FusedShardedTile(be, dist)
    tile.upload(h0, om)
    for rep in range(3):                   # consecutive frames reuse the exchange buffers behind the right events
        be.trace = [] if rep == 2 else None
        tile.frame(2.25)
    # the stream order of a frame (what the first multi-GPU run must not have to debug): pass 1 of part k on the compute
    # stream, the communication stream waits for it, the all-to-all of part k (its own slices of the buffers), ...; the
    # compute stream waits for the communication stream exactly once, right before pass 2
    want = []
    for k in range(parts):
        want += [("pass1", "compute", k), ("wait", "comm", "compute"), ("all_to_all", "comm", tile.recv[k].data_ptr(), tile.send[k].data_ptr())]
    want += [("wait", "compute", "comm"), ("pass2", "compute")]
    assert be.trace == want, (parts, be.trace, want)
    assert len({tile.send[k].data_ptr() for k in range(parts)}) == parts
    got = tile.gather_tile()
    nmax, rl2 = oc.parity_errors(got[..., :3], oc.frame_f64(h0, om, 2.25)[..., :3])
    assert nmax.max() < 2e-5 and rl2.max() < 2e-5 and np.all(got[..., 3] == 0.0), (parts, nmax, rl2)
    tile.b.destroy()
try:                                       # a wrong argument is raised, not papered over with a full-size context (ADVICE r05)
    sharded.HipTileBackend(n, 3, 2)
    raise SystemExit("rank 3 of world 2 was accepted")
except g.OceanError as e:
    assert e.status == -1, e
dist.destroy_process_group()
print("FUSED_SHARD_GPU_OK")
"""


@pytest.mark.gpu
def test_gpu_fused_shard_through_torch_and_rccl():
    """HipTileBackend + FusedShardedTile on torch memory and a torch stream, the all-to-all through RCCL (world 1)."""
    p = subprocess.run([sys.executable, "-c", _GPU_FUSED_WORLD1, ROOT, "1024"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "FUSED_SHARD_GPU_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
