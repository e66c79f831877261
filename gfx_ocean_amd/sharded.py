"""ONE N x N tile whose 2-D transform is sharded over the GPUs of a node (include/ocean_hip.h).
Independent tiles need no exchange (bench.py); this module is for a single tile that is too large or too slow
for one GPU.  Two schemes, both with ONE all-to-all where the reference has its barrier between the row and the column
dispatches (src/render.rs:1181-1208):

`FusedShardedTile` (N <= 16384, the one to use) -- the fused half-spectrum frame of `ocean_frame`, sharded:

    rank r owns half-spectrum columns [r N/2R, ..)   fused pass 1 on them                  (ocean_tile_pass1)
    all_to_all_single(recv, send)                     3 (N/2) (N/R) 8 bytes per rank: 12 B/texel in total; optionally cut
                                                      into `parts` all-to-alls overlapped with pass 1 of the next part
    rank r owns rows [r N/R, (r+1) N/R)               fused pass 2 on them                  (ocean_tile_pass2)

  Result: the rank's ROW block in the natural orientation, ``out[y - r N/R, x] = (disp_x, height, disp_z, 0)``; 54 B/texel
  of HBM traffic, bit-identical to `ocean_frame`.

`ShardedTile` (N <= 16384; the first generation, kept as the 1:1 restatement) -- the reference's own dispatch order (src/render.rs:1122-1310):

    rank r owns rows [r N/R, (r+1) N/R)           propagate + row pass on them        (ocean_shard_rows)
    all_to_all_single(recv, send)                 3 N^2 8 / R bytes per rank and frame, (R-1)/R of it over xGMI
    rank r owns columns [r N/R, (r+1) N/R)        column pass + correction on them    (ocean_shard_cols)

  Result: the rank's COLUMN block, transposed: ``out[x - r N/R, y]``; staged kernels, 220 B/texel of HBM traffic.

The drivers work on a backend with a handful of calls.  The product backends are `HipTileBackend` / `HipShardBackend` (the
C ABI on device memory, RCCL through torch.distributed); the tests plug in backends that execute the same kernels on the
CPU (tests/emu.py) under gloo.  There is no CPU fallback in the product: the Hip backends raise OceanError without the
HIP library or a GPU.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import OCEAN_OK, OceanError, PropagateLocalsC, load_library

E_UNSUPPORTED = -6      # include/ocean_hip.h OCEAN_E_UNSUPPORTED
from .ocean import DOMAIN_SIZE


def split_inputs(h0: np.ndarray, omega: np.ndarray, rank: int, world: int):
    """The static inputs rank `rank` of `world` needs: its own rows of h0 and omega, and the rows of h0 in which the
    "-k" partners of its texels live -- texel (gx, gy) pairs with (N-1-gx, N-1-gy) (shader/propagate.comp:48), i.e.
    with the opposite row block read backwards."""
    n = h0.shape[0]
    rows = n // world
    own = slice(rank * rows, (rank + 1) * rows)
    partner = slice(n - (rank + 1) * rows, n - rank * rows)
    return (np.ascontiguousarray(h0[own], np.complex64), np.ascontiguousarray(h0[partner], np.complex64),
            np.ascontiguousarray(omega[own], np.float32))


def exchange_bytes_per_rank(n: int, world: int) -> int:
    """All-to-all payload per rank and frame: three complex fp32 fields of N/world rows (its own part stays local)."""
    return 3 * (n // world) * n * 8


class HipShardBackend:
    """The C ABI (ocean_shard_*) on torch CUDA tensors; torch supplies device memory and the collective only."""

    def __init__(self, n: int, rank: int, world: int, device_ordinal: int = 0):
        import torch
        self.torch = torch
        self.lib = load_library()
        h = ctypes.c_void_p()
        st = self.lib.ocean_shard_create(int(device_ordinal), int(n), int(rank), int(world), ctypes.byref(h))
        if st != OCEAN_OK:
            raise OceanError(st, (self.lib.ocean_shard_last_error(None) or b"").decode())
        self._h = h
        self.n, self.rank, self.world, self.rows = n, rank, world, n // world
        self.device = torch.device("cuda", device_ordinal)
        # Kernels and the collective are ordered on ONE explicit stream.  (torch's default stream has the handle 0,
        # which the C ABI reads as "the shard's own stream": work launched there would not be ordered with torch's.)
        self.stream = torch.cuda.Stream(self.device)

    def _check(self, st):
        if st != OCEAN_OK:
            raise OceanError(st, (self.lib.ocean_shard_last_error(self._h) or b"").decode())

    def upload(self, h0_own, h0_partner, omega_own):
        self._check(self.lib.ocean_shard_upload(self._h, h0_own.ctypes.data, h0_partner.ctypes.data, omega_own.ctypes.data))

    def alloc_exchange(self):
        """[dest or src][field][row][column] complex fp32, as float32 pairs (RCCL moves plain floats)."""
        return self.torch.empty((self.world, 3, self.rows, self.rows, 2), dtype=self.torch.float32, device=self.device)

    def alloc_out(self):
        return self.torch.empty((self.rows, self.n, 4), dtype=self.torch.float32, device=self.device)

    def on_stream(self):
        """Context manager: torch work issued inside (the all-to-all) is ordered with the shard's kernels."""
        return self.torch.cuda.stream(self.stream)

    def rows_pass(self, time, domain_size, send):
        loc = PropagateLocalsC(float(time), int(self.n), float(domain_size))
        self._check(self.lib.ocean_shard_rows(self._h, ctypes.byref(loc), send.data_ptr(), self.stream.cuda_stream))

    def cols_pass(self, recv, out):
        self._check(self.lib.ocean_shard_cols(self._h, recv.data_ptr(), out.data_ptr(), self.stream.cuda_stream))

    def synchronize(self):
        self.stream.synchronize()

    def to_numpy(self, out):
        self.stream.synchronize()
        return out.cpu().numpy()

    def destroy(self):
        if self._h:
            self.lib.ocean_shard_destroy(self._h)
            self._h = None


class ShardedTile:
    """frame(t) = rows pass -> one all-to-all -> column pass.  `dist` is an initialised torch.distributed (RCCL on the
    GPUs, gloo in the CPU tests); None (world == 1 only) skips the collective."""

    def __init__(self, backend, dist=None, domain_size: float = DOMAIN_SIZE):
        if dist is None and backend.world != 1:
            # without a process group the send buffer would be consumed as if it were the column block: a wrong tile,
            # silently.  (Multi-rank RCCL runs of the shard are unmeasured until a box with >= 2 GPUs runs them; the
            # multi-rank kernels themselves are checked on one device, tests/test_sharded.py.)
            raise OceanError(-1, f"ShardedTile: world = {backend.world} needs an initialised torch.distributed (dist=None is world 1 only)")
        self.b, self.dist, self.domain_size = backend, dist, float(domain_size)
        self.send = backend.alloc_exchange()
        self.recv = backend.alloc_exchange()
        self.out = backend.alloc_out()

    def upload(self, h0: np.ndarray, omega: np.ndarray):
        """Every rank passes the same full arrays (or at least its own slices of them) and keeps only what it needs."""
        self.b.upload(*split_inputs(h0, omega, self.b.rank, self.b.world))

    def frame(self, time: float):
        self.b.rows_pass(time, self.domain_size, self.send)
        if self.dist is not None:
            with self.b.on_stream():
                self.dist.all_to_all_single(self.recv, self.send)  # the only collective of the frame
            recv = self.recv
        else:
            recv = self.send                                       # one rank, no group: the send buffer IS the column block
        self.b.cols_pass(recv, self.out)
        return self.out

    def result(self) -> np.ndarray:
        """The rank's column block, transposed: [N/world, N, 4] with [x - x0, y] = (disp_x, height, disp_z, 0)."""
        return self.b.to_numpy(self.out)

    def gather_tile(self) -> np.ndarray | None:
        """Whole tile in the natural orientation [y, x, 4] on rank 0 (tests / small N only)."""
        mine = self.result()
        if self.b.world == 1:
            return np.ascontiguousarray(mine.transpose(1, 0, 2))
        parts = [None] * self.b.world if self.b.rank == 0 else None
        self.dist.gather_object(mine, parts, dst=0)
        if self.b.rank != 0:
            return None
        return np.ascontiguousarray(np.concatenate(parts, axis=0).transpose(1, 0, 2))


# ----------------------------------------------------------------------------------------------------------------------
# Second generation: the FUSED half-spectrum frame, sharded (include/ocean_hip.h ocean_tile_pass1 / ocean_tile_pass2)
# ----------------------------------------------------------------------------------------------------------------------
def fused_exchange_bytes_per_rank(n: int, world: int) -> int:
    """All-to-all payload per rank and frame of the fused sharded tile: the three symmetrised spectra of the rank's
    N/(2 world) half-spectrum columns, complex fp32 -- half of `exchange_bytes_per_rank` (the Hermitian half suffices)."""
    return 3 * (n // 2 // world) * n * 8


def tile_rank_lines(n: int, rank: int, world: int) -> np.ndarray:
    """The lines of the transposed inputs (h0T[x][:], omegaT[x][:] = columns x of the natural arrays) that pass 1 of rank `rank`
    of a fused sharded tile reads: its half-spectrum columns [a, b) need the "own-type" lines a-1 .. b-1 and the "mirror-type"
    lines N-b .. N-a (mod N), and the workgroup that owns column 0 (rank 0) the Nyquist column's lines N/2-1 and N/2.
    Sorted, unique.  2 (N / 2 world + 1) lines (+ 2 on rank 0) of N."""
    a, b = rank * (n // 2 // world), (rank + 1) * (n // 2 // world)
    lines = set(x % n for x in range(a - 1, b)) | set(x % n for x in range(n - b, n - a + 1))
    if rank == 0:
        lines |= {n // 2 - 1, n // 2}
    return np.array(sorted(lines), dtype=np.int64)


class HipTileBackend:
    """ocean_tile_pass1 / ocean_tile_pass2 on an ordinary OceanDevice that holds the whole tile's static inputs; torch
    supplies the exchange buffers, the streams and the collective.

    UNMEASURED on more than one GPU: the pool this was developed on gives one GPU per call, so the RCCL all-to-all, the
    two-stream pipelining (`exchange` / `join_exchanges`) and the `[part][peer][message]` buffers have run through
    `all_to_all_single` only at world = 1; every rank's KERNELS of world 2 / 4 / 8 are checked on one device against the
    oracle (tests/test_sharded.py), the multi-rank driver under gloo on the CPU emulation backend.

    The context is a band-limited rank context (ocean_context_create_tile_rank): of the tile's static inputs only the lines
    pass 1 of THIS rank reads are backed by memory -- lines x, x-1, N-x, N-1-x for its columns x: two bands of N/(2 world) + 1
    lines (tile_rank_lines; + the Nyquist column's two on rank 0), 12/world B/texel, 0.4 GiB instead of 3 at N = 16384 and
    world 8 -- and nothing else: no staged buffers, no intermediate, no map; the exchange buffers and the rank's rows are
    torch's.  The two address ranges keep their full size (HIP virtual-memory API), so the kernels index absolute lines as in
    every other context.  [Round 4 used a full context: 76-100 B/texel, 20 GiB per rank at 16384.]

    `context_kind`: "bands", or "full (...)" when the runtime cannot map memory sparsely (OCEAN_E_UNSUPPORTED; any other failure of
    the context is raised, not papered over).  `trace` (a list, or None): when set, every stream-ordering step of a frame is appended to it -- what
    tests/test_sharded.py checks before the first multi-GPU run has to debug RCCL rather than bookkeeping."""

    def __init__(self, n: int, rank: int, world: int, device_ordinal: int = 0, parts: int = 1):
        import torch
        from .render import OceanDevice
        self.torch = torch
        self.lib = load_library()
        from .render import CTX_TILE_RANK
        # Only the two bands of input lines this rank reads (12 / world B/texel).  The whole tile's inputs (12 B/texel) ONLY if the
        # runtime cannot map memory sparsely (OCEAN_E_UNSUPPORTED): a wrong argument or an out-of-memory context must not turn,
        # silently, into a context of `world` times the footprint.  `context_kind` says which one this backend runs on.
        try:
            self.dev = OceanDevice.for_tile_rank(n, rank, world, device_ordinal)
            self.context_kind = "bands"
        except OceanError as e:
            if e.status != E_UNSUPPORTED:
                raise
            self.dev = OceanDevice(n, device_ordinal, flags=CTX_TILE_RANK)
            self.context_kind = "full (sparse mapping unsupported: " + str(e) + ")"
        self.trace = None
        self.n, self.rank, self.world, self.rows, self.parts = n, rank, world, n // world, parts
        self.device = torch.device("cuda", device_ordinal)
        self.stream = torch.cuda.Stream(self.device)          # see HipShardBackend: explicit streams for kernels and collectives
        self.comm = torch.cuda.Stream(self.device)            # the all-to-alls of a pipelined exchange (parts > 1)
        nbytes = int(self.lib.ocean_tile_exchange_bytes(self.dev._ctx, world))
        if nbytes < 0 or nbytes % (4 * world * parts):
            raise OceanError(-1, f"sharded tile: N = {n} cannot be split over {world} ranks x {parts} parts")
        self.exchange_floats = nbytes // 4

    def upload(self, h0, omega):
        # every rank is handed the whole tile's (static) inputs and keeps -- and copies to the device -- only the columns that
        # become the lines it reads (ocean_upload_spectrum on a band-limited context: 1 / world of the tile crosses PCIe)
        self.dev.upload_spectrum(h0, omega)

    def alloc_exchange(self):
        """[part][peer][message]: one contiguous all-to-all buffer per part."""
        return self.torch.empty((self.parts, self.world, self.exchange_floats // self.world // self.parts), dtype=self.torch.float32,
                                device=self.device)

    def alloc_out(self):
        return self.torch.empty((self.rows, self.n, 4), dtype=self.torch.float32, device=self.device)

    def pass1(self, time, domain_size, send_part, part=0):
        loc = PropagateLocalsC(float(time), int(self.n), float(domain_size))
        self._t("pass1", "compute", int(part))
        self.dev._check(self.lib.ocean_tile_pass1(self.dev._ctx, ctypes.byref(loc), self.rank, self.world, int(part), self.parts,
                                                  send_part.data_ptr(), self.stream.cuda_stream))

    def _t(self, *what):
        if self.trace is not None:
            self.trace.append(what)

    def pass2(self, recv, out):
        self._t("pass2", "compute")
        self.dev._check(self.lib.ocean_tile_pass2(self.dev._ctx, self.rank, self.world, self.parts, recv.data_ptr(), out.data_ptr(),
                                                  self.stream.cuda_stream))

    def exchange(self, dist, recv_part, send_part):
        """The all-to-all of one part on the communication stream, behind everything the compute stream holds so far."""
        self._t("wait", "comm", "compute")
        self.comm.wait_stream(self.stream)
        with self.torch.cuda.stream(self.comm):
            self._t("all_to_all", "comm", int(recv_part.data_ptr()), int(send_part.data_ptr()))
            dist.all_to_all_single(recv_part, send_part)

    def join_exchanges(self):
        self._t("wait", "compute", "comm")
        self.stream.wait_stream(self.comm)                    # pass 2 (and the next frame's pass 1) behind every all-to-all

    def synchronize(self):
        self.stream.synchronize()

    def to_numpy(self, out):
        self.stream.synchronize()
        return out.cpu().numpy()

    def destroy(self):
        self.dev.destroy()


class FusedShardedTile:
    """frame(t) = pass 1 on the rank's half-spectrum columns -> all-to-all (12 B/texel in total) -> pass 2 on the rank's
    rows.  With backend.parts = K > 1 the column block is cut into K pieces and the exchange into K all-to-alls on a second
    stream: the exchange of piece k runs under pass 1 of piece k + 1 (how much of the exchange that hides is UNMEASURED:
    no run of this class has had more than one GPU, see HipTileBackend).
    The result is distributed by ROW blocks in the natural orientation: out[y - r N/R, x] = (dx, h, dz, 0).
    `dist`: an initialised torch.distributed (RCCL on GPUs, gloo in the CPU tests); None only for world == 1."""

    def __init__(self, backend, dist=None, domain_size: float = DOMAIN_SIZE):
        if dist is None and backend.world != 1:
            raise OceanError(-1, f"FusedShardedTile: world = {backend.world} needs an initialised torch.distributed")
        self.b, self.dist, self.domain_size = backend, dist, float(domain_size)
        self.parts = int(getattr(backend, "parts", 1))
        self.send = backend.alloc_exchange()                       # [part][dest][message]
        self.recv = backend.alloc_exchange()                       # [part][src][message]
        self.out = backend.alloc_out()

    def upload(self, h0: np.ndarray, omega: np.ndarray):
        self.b.upload(np.ascontiguousarray(h0, np.complex64), np.ascontiguousarray(omega, np.float32))

    def frame(self, time: float):
        for k in range(self.parts):
            self.b.pass1(time, self.domain_size, self.send[k], part=k)
            if self.dist is not None:
                self.b.exchange(self.dist, self.recv[k], self.send[k])     # overlaps pass 1 of part k + 1
        if self.dist is not None:
            self.b.join_exchanges()
            recv = self.recv
        else:
            recv = self.send                                       # one rank, no group: what it sends is what it receives
        self.b.pass2(recv, self.out)
        return self.out

    def result(self) -> np.ndarray:
        """The rank's block of rows: [N/world, N, 4]."""
        return self.b.to_numpy(self.out)

    def gather_tile(self) -> np.ndarray | None:
        """Whole tile [y, x, 4] on rank 0 (tests / small N only)."""
        mine = self.result()
        if self.b.world == 1:
            return mine
        parts = [None] * self.b.world if self.b.rank == 0 else None
        self.dist.gather_object(mine, parts, dst=0)
        if self.b.rank != 0:
            return None
        return np.ascontiguousarray(np.concatenate(parts, axis=0))
