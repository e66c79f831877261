"""GPU tier: NATIVE callers drive the C ABI.  A C++ program through the host mirror
(csrc/host/ocean.hpp: Device / Propagation / Fft / Correction / render, what src/render.rs:223-225 and
:1101-1310 would bind) and a plain C99 program through include/ocean_hip.h decode the reference's own bincode
inputs, run the frame at N = 512, t = 1 on the GPU and compare with the golden crop produced by executing the
reference's shipped SPIR-V (tests/golden/spirv_frame512_t1.npz).  No Python between the caller and the library."""
import os
import subprocess

import numpy as np
import pytest

import gfx_ocean_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build(tmp_path, compiler, std, src, name):
    so = g.build_library()
    libdir = os.path.dirname(so)
    exe = str(tmp_path / name)
    subprocess.check_call([compiler, std, "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", src),
                           "-o", exe, "-L", libdir, "-locean_hip", "-Wl,-rpath," + libdir,
                           "-L", "/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return exe


def _crop_file(tmp_path):
    crop = np.load(os.path.join(GOLDEN, "spirv_frame512_t1.npz"))["crop"]      # float32 [64, 64, 4]
    path = str(tmp_path / "crop.f32")
    np.ascontiguousarray(crop, dtype="<f4").tofile(path)
    return path


@pytest.mark.gpu
def test_cpp_host_mirror_renders_the_reference_frame(tmp_path):
    exe = _build(tmp_path, "g++", "-std=c++17", "host_mirror_check.cpp", "host_mirror_check")
    p = subprocess.run([exe, os.path.join(GOLDEN, "spectrum.bin"), os.path.join(GOLDEN, "omega.bin"), _crop_file(tmp_path), "1.0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "native c++: staged" in p.stdout
    print(p.stdout.strip())


@pytest.mark.gpu
def test_plain_c_caller_renders_the_reference_frame(tmp_path):
    exe = _build(tmp_path, "gcc", "-std=c99", "c_abi_check.c", "c_abi_check")
    p = subprocess.run([exe, os.path.join(GOLDEN, "spectrum.bin"), os.path.join(GOLDEN, "omega.bin"), _crop_file(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "native c: fused" in p.stdout
    print(p.stdout.strip())


@pytest.mark.gpu
def test_displacement_map_in_memory_imported_from_a_file_descriptor(tmp_path):
    """INTEGRATION.md 4 / SURVEY 8f #4 "interop": the map lives in an allocation another owner exported as a POSIX file descriptor
    (the role of the reference's own image memory, src/render.rs:820-869); ocean_bind_displacement_fd imports it and frames land
    there -- read back through the exporter's mapping, bit-identical to the library-owned frame."""
    so = g.build_library()
    libdir = os.path.dirname(so)
    exe = str(tmp_path / "ext_interop_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "ext_interop_check.hip"), "-o", exe, "-L", libdir, "-locean_hip", "-Wl,-rpath," + libdir])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "bit for bit" in p.stdout
    print(p.stdout.strip())
