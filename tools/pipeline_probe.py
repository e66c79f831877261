#!/usr/bin/env python3
"""Experiment: pass 1 of frame f + 1 and pass 2 of frame f SIDE BY SIDE on disjoint sets of CUs (streams created with
hipExtStreamCreateWithCUMask; tools/cumask_probe.hip: bits [0, c) = c/8 CUs of every XCD).  Pass 1 is VALU-bound for 60 % of its
time and leaves HBM idle then; pass 2 is HBM-bound with little arithmetic.  Uses the two halves of the fused frame as the C ABI
exposes them separately (ocean_tile_pass1 / ocean_tile_pass2 with world = 1), two intermediates, two maps, events between the
streams.  Prints frames/s of: one stream (serial), two unmasked streams, and partitions c1 + c2 = 256.
    pipeline_probe.py [N] [frames]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gfx_ocean_amd as g  # noqa: E402
from gfx_ocean_amd._lib import PropagateLocalsC  # noqa: E402
from hipmem import DeviceBuffer, hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H = hip()
H.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
H.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
H.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
H.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
H.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
H.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
H.hipStreamDestroy.argtypes = [ctypes.c_void_p]


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    st = H.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert st == 0, f"hipExtStreamCreateWithCUMask -> {st}"
    return s


def plain_stream():
    s = ctypes.c_void_p()
    assert H.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0          # hipStreamNonBlocking
    return s


def event():
    e = ctypes.c_void_p()
    assert H.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0           # hipEventDisableTiming
    return e


lib = g.load_library()
h0, om = g.synth.make_inputs(n, seed=3)
dev = g.OceanDevice(n, flags=g.CTX_FUSED_ONLY)
dev.upload_spectrum(h0, om)
xbytes = int(lib.ocean_tile_exchange_bytes(dev._ctx, 1))
X = [DeviceBuffer(xbytes), DeviceBuffer(xbytes)]
OUT = [DeviceBuffer(n * n * 16), DeviceBuffer(n * n * 16)]
ev1, ev2 = [event(), event()], [event(), event()]


def run(count, sA, sB, pipelined):
    for f in range(count):
        b = f & 1
        loc = PropagateLocalsC(f / 60.0, n, 1000.0)
        if pipelined and f >= 2:
            assert H.hipStreamWaitEvent(sA, ev2[b], 0) == 0             # pass 2 of frame f - 2 has read X[b]
        dev._check(lib.ocean_tile_pass1(dev._ctx, ctypes.byref(loc), 0, 1, 0, 1, ctypes.c_void_p(X[b].ptr), sA))
        if pipelined:
            assert H.hipEventRecord(ev1[b], sA) == 0
            assert H.hipStreamWaitEvent(sB, ev1[b], 0) == 0
        dev._check(lib.ocean_tile_pass2(dev._ctx, 0, 1, 1, ctypes.c_void_p(X[b].ptr), ctypes.c_void_p(OUT[b].ptr), sB))
        if pipelined:
            assert H.hipEventRecord(ev2[b], sB) == 0
    assert H.hipStreamSynchronize(sA) == 0 and H.hipStreamSynchronize(sB) == 0


def measure(name, sA, sB, pipelined):
    run(100, sA, sB, pipelined)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        run(frames, sA, sB, pipelined)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"N={n} {name:34s} {frames / best:9.1f} frames/s   {best / frames * 1e6:7.1f} us per frame", flush=True)


# the maps of the last two frames against ocean_frame at the same times (checked once, on the first partition)
def check(sA, sB):
    run(6, sA, sB, True)
    for f in (4, 5):
        dev.frame(f / 60.0)
        want = dev.read_displacement()
        got = OUT[f & 1].to_host().reshape(n, n, 4)
        assert np.array_equal(got, want), f"frame {f}: the pipelined map differs"
    print("maps of the pipelined frames == ocean_frame, bit for bit", flush=True)


one = plain_stream()
measure("one stream (serial)", one, one, False)
dev.time_frames(100)
print(f"N={n} {'ocean_frame loop (the product)':34s} {frames / dev.time_frames(frames) * 1000.0:9.1f} frames/s", flush=True)
a, b = plain_stream(), plain_stream()
check(a, b)
measure("two streams, no masks", a, b, True)
for c1 in (232, 224, 208, 192, 176, 160, 128):
    sA, sB = masked_stream(0, c1), masked_stream(c1, 256)
    measure(f"pass 1 on {c1} CUs | pass 2 on {256 - c1}", sA, sB, True)
    H.hipStreamDestroy(sA)
    H.hipStreamDestroy(sB)
measure("one stream (serial), again", one, one, False)
