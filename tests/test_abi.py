"""The C-ABI library builds for gfx950, loads, and exports every symbol include/ocean_hip.h
declares (no compute calls: there is no GPU in this tier)."""
import os
import re
import subprocess

import pytest

import gfx_ocean_amd as g
from gfx_ocean_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so_path():
    return g.build_library()      # hipcc cross-compiles without a GPU


def header_symbols():
    with open(os.path.join(ROOT, "include", "ocean_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ocean_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(so_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so_path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing


def test_library_loads_and_reports_version(so_path):
    lib = g.load_library()
    assert lib.ocean_abi_version() == 4


def test_library_contains_gfx950_code(so_path):
    with open(so_path, "rb") as f:
        blob = f.read()
    assert b"gfx950" in blob and b"k_half_pass1" in blob and b"k_half_pass2" in blob and b"k_fft_lines" in blob


def test_null_handles_are_rejected_not_crashed(so_path):
    lib = g.load_library()
    assert lib.ocean_frame(None, 0.0, None) == -1
    assert lib.ocean_fft_rows(None, 0, None) == -1
    assert lib.ocean_sync(None) == -1
    lib.ocean_context_destroy(None)      # NULL-safe
    lib.ocean_fft_destroy(None)


def test_cpp_host_mirror_compiles_and_links(so_path, tmp_path):
    """gfx_ocean_amd/csrc/host/ocean.hpp (the C++ mirror of mod ocean / mod fft) against the C ABI."""
    exe = str(tmp_path / "host_mirror_check")
    libdir = os.path.dirname(so_path)
    subprocess.check_call(["g++", "-std=c++17", os.path.join(ROOT, "tests", "host_mirror_check.cpp"), "-o", exe,
                           "-L", libdir, "-locean_hip", "-Wl,-rpath," + libdir,
                           "-L", "/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([exe]) == 0


def test_header_is_plain_c_and_links_from_c(so_path, tmp_path):
    """include/ocean_hip.h under gcc -std=c99 -pedantic, and a C program against the library (error paths only)."""
    exe = str(tmp_path / "c_abi_check")
    libdir = os.path.dirname(so_path)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_check.c"), "-o", exe,
                           "-L", libdir, "-locean_hip", "-Wl,-rpath," + libdir,
                           "-L", "/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    assert subprocess.call([exe]) == 0


def test_rust_shim_lists_every_symbol():
    """The uncompiled Rust shim (no cargo in this image) must at least bind every exported symbol."""
    with open(os.path.join(ROOT, "gfx_ocean_amd", "rust", "src", "ffi.rs")) as f:
        src = f.read()
    bound = sorted(set(re.findall(r"pub fn (ocean_[a-z_0-9]+)\(", src)))
    assert bound == header_symbols()
