"""Race / reproducibility evidence on the GPU (the reference's discipline is the barrier chain of
shader/fft_row.comp:48-59, ours is ~10 hand-placed __syncthreads per fused kernel plus LDS buffer reuse across the three
fields).  A missing or misplaced barrier shows up as run-to-run differences long before it shows up as a parity
failure, so: the same frame, many times in a row with other frames' launches in flight around it, must be
BIT-identical every time -- at every size, for the fp32 and the fp16-stored spectrum, fused and staged.  The
comparison is a device-side order-independent checksum (ocean_checksum_displacement), validated against numpy here."""
import numpy as np
import pytest

import gfx_ocean_amd as g

pytestmark = pytest.mark.gpu


def host_checksum(rgba: np.ndarray) -> int:
    w = np.ascontiguousarray(rgba).view(np.uint32).ravel()
    k = (w + np.uint32(0x9E3779B9)).astype(np.uint64)                 # uint32 wrap-around add, as the kernel
    idx = np.arange(w.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over="ignore"):
        return int((k * idx).sum(dtype=np.uint64))


def test_device_checksum_matches_numpy_and_sees_one_bit():
    n = 512
    d = g.OceanDevice(n)
    try:
        h0, om = g.synth.make_inputs(n, seed=3)
        d.upload_spectrum(h0, om)
        d.frame(1.0)
        a = d.read_displacement()
        assert d.checksum() == host_checksum(a)
        d.frame(1.0 + 1e-3)                                          # any other frame: another sum
        assert d.checksum() != host_checksum(a)
        b = a.copy()
        b.view(np.uint32)[123, 45, 2] ^= 1                            # one flipped mantissa bit changes the sum
        assert host_checksum(b) != host_checksum(a)
    finally:
        d.destroy()


CASES = [(256, False), (512, False), (512, True), (1024, False), (2048, False), (2048, True), (4096, False),
         (4096, True), (8192, False), (8192, True), (16384, False)]   # 16384: the ring's three-slot, uneven-vmcnt geometry


@pytest.mark.parametrize("n,f16", CASES)
def test_200_consecutive_frames_are_bit_identical(n, f16):
    """Fused and staged: frame(t) interleaved with frames at other times (so that stale LDS / intermediate contents
    would differ), checksummed on the device after every repetition."""
    reps_fused, reps_staged = (200 if n <= 8192 else 60), (200 if n <= 2048 else (60 if n == 4096 else (24 if n == 8192 else 0)))
    h0, om = g.synth.make_inputs(n, seed=n + 1)
    r = g.OceanRenderer(n)
    try:
        r.upload(h0, om, spectrum_fp16=f16)
        d = r.device
        t = 2.75
        d.frame(t)
        want = d.checksum()
        if n <= 8192:                                                 # (the device checksum against numpy; 4 GiB of map at 16384: not needed again)
            assert want == host_checksum(d.read_displacement())
        for i in range(reps_fused):
            if i % 3 == 1:
                d.frame(0.01 * i)                                     # another frame in between, not synchronised
            d.frame(t)
            assert d.checksum() == want, f"fused frame {i} differs (N={n}, f16={f16})"
        if not f16 and reps_staged:                                   # the staged path reads the fp32 (dequantised) spectrum either way
            r.render(t)
            want_s = d.checksum()
            for i in range(reps_staged):
                if i % 3 == 1:
                    r.render(0.01 * i)
                r.render(t)
                assert d.checksum() == want_s, f"staged frame {i} differs (N={n})"
    finally:
        r.dispose()


def test_create_destroy_soak():
    """tools/soak.py's loop as a test: contexts of every size created, used and destroyed repeatedly (handle registry,
    leaks: 3 x 6 contexts of up to 5.6 GiB each would exhaust nothing if freed, and fail loudly if not)."""
    first = {}
    for rep in range(3):
        for n in (256, 512, 1024, 2048, 4096, 8192):
            d = g.OceanDevice(n)
            h0, om = g.synth.make_inputs(n, seed=7)
            d.upload_spectrum(h0, om, spectrum_fp16=(rep == 1))
            d.frame(1.0)
            c = d.checksum()
            d.time_frames(20)
            d.frame(1.0)
            assert d.checksum() == c
            key = (n, rep == 1)
            assert first.setdefault(key, c) == c                     # and across contexts
            d.destroy()
            with pytest.raises(g.OceanError):
                d.frame(0.0)
    assert first[(512, False)] != first[(512, True)]


_JITTER_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import gfx_ocean_amd as g
res = {}
for n in [int(v) for v in os.environ.get("OCEAN_RACE_SIZES", "256,512,1024,2048,4096,8192,16384").split(",")]:
    for f16 in ((False, True) if n <= 8192 else (False,)):      # 16384: the ring with three slots and uneven per-wave vmcnt budgets
        h0, om = g.synth.make_inputs(n, seed=n + 1)
        r = g.OceanRenderer(n)
        r.upload(h0, om, spectrum_fp16=f16)
        sums = set()
        for i in range(int(sys.argv[2]) if n <= 8192 else max(2, int(sys.argv[2]) // 2)):
            r.render_fused(2.75)
            sums.add(r.device.checksum())
        key = f"{n}:{int(f16)}"
        res[key + ":fused"] = sorted(sums)
        if not f16 and n <= 4096:
            sums = set()
            for i in range(max(4, int(sys.argv[2]) // 8)):
                r.render(2.75)
                sums.add(r.device.checksum())
            res[key + ":staged"] = sorted(sums)
        r.dispose()
print("SUMS " + json.dumps(res))
"""


def test_barrier_jitter_build_is_bit_identical(tmp_path):
    """Race hunting without a device sanitizer (none on this pool: profiles/r03_run16_asan_attempt_log.txt): the library is
    rebuilt with every workgroup barrier wrapped in pseudo-random wave-uniform sleeps (-DOCEAN_RACE_JITTER,
    csrc/ocean_device_intrinsics.hpp).  Same arithmetic, perturbed wave timing: every frame of that build -- all sizes,
    fp32 and fp16-stored spectrum, fused and staged -- must have the checksum the product build produces."""
    import json
    import os
    import subprocess
    import sys
    from gfx_ocean_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libocean_hip_jitter.so")
    subprocess.check_call(_lib.hipcc_command(out=so, extra=("-DOCEAN_RACE_JITTER",)))

    def sums(lib, reps):
        env = dict(os.environ)
        if lib:
            env["OCEAN_HIP_LIB"] = lib
        else:
            env.pop("OCEAN_HIP_LIB", None)
        p = subprocess.run([sys.executable, "-c", _JITTER_WORKER, root, str(reps)], capture_output=True, text=True, timeout=1500, env=env)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("SUMS ")][0][5:])

    want = sums(None, 2)
    got = sums(so, 64)
    assert set(want) == set(got)
    for key, v in want.items():
        assert len(v) == 1, (key, "the product build itself is not reproducible", v)
        assert got[key] == v, (key, "jittered barriers changed the result: a race", got[key], v)
