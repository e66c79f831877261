"""ctypes binding of libocean_hip.so (include/ocean_hip.h).  No fallback path."""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libocean_hip.so")
_CSRC = os.path.join(_HERE, "csrc")
_LIB = None

OCEAN_OK = 0
STATUS_NAMES = {0: "OCEAN_OK", -1: "OCEAN_E_INVALID_ARG", -2: "OCEAN_E_UNSUPPORTED_N", -3: "OCEAN_E_HIP",
                -4: "OCEAN_E_OOM", -5: "OCEAN_E_STATE", -6: "OCEAN_E_UNSUPPORTED"}

# every symbol include/ocean_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "ocean_abi_version", "ocean_device_count", "ocean_device_pci_bus_id", "ocean_context_create", "ocean_context_create_ex", "ocean_context_create_tile_rank", "ocean_context_flags", "ocean_context_destroy", "ocean_last_error", "ocean_resolution",
    "ocean_upload_spectrum", "ocean_upload_spectrum_f16", "ocean_spectrum_scale_log2", "ocean_read_spectrum",
    "ocean_fft_init", "ocean_fft_destroy", "ocean_propagation_init",
    "ocean_propagation_destroy", "ocean_correction_init", "ocean_correction_destroy", "ocean_propagate",
    "ocean_fft_rows", "ocean_fft_cols", "ocean_correct", "ocean_frame", "ocean_frame_ex", "ocean_sync",
    "ocean_context_create_tiles", "ocean_context_tiles", "ocean_upload_spectrum_tile", "ocean_upload_spectrum_device", "ocean_frame_tiles",
    "ocean_frame_batch", "ocean_batch_device_ptr", "ocean_batch_normals_device_ptr", "ocean_read_batch_normals", "ocean_read_batch_displacement", "ocean_time_frame_batch",
    "ocean_set_quirks", "ocean_quirks", "ocean_set_intermediate", "ocean_intermediate",
    "ocean_normals", "ocean_read_normals", "ocean_set_frame_normals", "ocean_frame_normals", "ocean_normals_device_ptr", "ocean_positions", "ocean_read_positions",
    "ocean_checksum_displacement", "ocean_packed_bytes", "ocean_pack_displacement",
    "ocean_read_displacement", "ocean_read_field", "ocean_write_field", "ocean_displacement_device_ptr",
    "ocean_bind_displacement", "ocean_bind_displacement_fd", "ocean_stream", "ocean_time_frames", "ocean_time_frame_batches", "ocean_frame_times", "ocean_frame_times_ex", "ocean_profile_frame", "ocean_profile_staged",
    "ocean_shard_create", "ocean_shard_destroy", "ocean_shard_last_error", "ocean_shard_upload", "ocean_shard_rows",
    "ocean_shard_cols", "ocean_shard_sync", "ocean_shard_stream",
    "ocean_tile_exchange_bytes", "ocean_tile_pass1", "ocean_tile_pass2",
]


def device_count() -> int:
    """Visible HIP devices (0 without a GPU or a driver)."""
    return max(0, int(load_library().ocean_device_count()))


def device_numa(ordinal: int):
    """(pci bus id, NUMA node or None, cpu list or None) of HIP device `ordinal`, from sysfs."""
    buf = ctypes.create_string_buffer(32)
    if load_library().ocean_device_pci_bus_id(int(ordinal), buf, 32) != OCEAN_OK:
        return None, None, None
    bus = buf.value.decode().lower()
    node, cpus = None, None
    try:
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read().strip())
        with open(f"/sys/bus/pci/devices/{bus}/local_cpulist") as f:
            cpus = parse_cpulist(f.read().strip())
    except (OSError, ValueError):
        pass
    return bus, (node if node is not None and node >= 0 else None), (cpus or None)


def parse_cpulist(text: str):
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


class OceanError(RuntimeError):
    """A non-zero status from the C ABI (the reference would `?`/`unwrap()` here)."""

    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class PropagateLocalsC(ctypes.Structure):
    _fields_ = [("time", ctypes.c_float), ("resolution", ctypes.c_int32), ("domain_size", ctypes.c_float)]


class CorrectionLocalsC(ctypes.Structure):
    _fields_ = [("resolution", ctypes.c_uint32)]


def library_path() -> str:
    return _SO


def hipcc_command(out: str = _SO, extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", _CSRC, *extra,
            "-shared", "-fPIC", os.path.join(_CSRC, "ocean_api.hip"), "-o", out]


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)] + [
        os.path.join(os.path.dirname(_HERE), "include", "ocean_hip.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(hipcc_command())
    return _SO


def load_library():
    """dlopen libocean_hip.so.  Raises OceanError if it has not been built: there is no CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("OCEAN_HIP_LIB", _SO)      # A/B builds of the same ABI (tools/ab_variants.sh)
    if not os.path.exists(so):
        raise OceanError(-3, f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(the HIP extension is the only implementation; there is no CPU fallback)")
    L = ctypes.CDLL(so)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    pp = ctypes.POINTER(vp)
    sig = {
        "ocean_abi_version": (i32, []),
        "ocean_device_count": (i32, []),
        "ocean_device_pci_bus_id": (i32, [i32, ctypes.c_char_p, i32]),
        "ocean_context_create": (i32, [i32, i32, pp]),
        "ocean_context_create_ex": (i32, [i32, i32, ctypes.c_uint32, pp]),
        "ocean_context_create_tile_rank": (i32, [i32, i32, i32, i32, pp]),
        "ocean_context_flags": (ctypes.c_uint32, [vp]),
        "ocean_context_destroy": (None, [vp]),
        "ocean_last_error": (ctypes.c_char_p, [vp]),
        "ocean_resolution": (i32, [vp]),
        "ocean_upload_spectrum": (i32, [vp, vp, vp]),
        "ocean_upload_spectrum_f16": (i32, [vp, vp, vp]),
        "ocean_spectrum_scale_log2": (i32, [vp]),
        "ocean_read_spectrum": (i32, [vp, vp]),
        "ocean_fft_init": (i32, [vp, pp]),
        "ocean_fft_destroy": (None, [vp]),
        "ocean_propagation_init": (i32, [vp, pp]),
        "ocean_propagation_destroy": (None, [vp]),
        "ocean_correction_init": (i32, [vp, pp]),
        "ocean_correction_destroy": (None, [vp]),
        "ocean_propagate": (i32, [vp, ctypes.POINTER(PropagateLocalsC), vp]),
        "ocean_fft_rows": (i32, [vp, i32, vp]),
        "ocean_fft_cols": (i32, [vp, i32, vp]),
        "ocean_correct": (i32, [vp, ctypes.POINTER(CorrectionLocalsC), vp]),
        "ocean_frame": (i32, [vp, f32, vp]),
        "ocean_frame_ex": (i32, [vp, ctypes.POINTER(PropagateLocalsC), vp]),
        "ocean_sync": (i32, [vp]),
        "ocean_context_create_tiles": (i32, [i32, i32, i32, pp]),
        "ocean_context_tiles": (i32, [vp]),
        "ocean_upload_spectrum_tile": (i32, [vp, i32, vp, vp]),
        "ocean_upload_spectrum_device": (i32, [vp, i32, vp, vp, vp]),
        "ocean_frame_tiles": (i32, [vp, f32, vp, ctypes.c_int64, vp]),
        "ocean_frame_batch": (i32, [vp, f32, f32, i32, vp, ctypes.c_int64, vp]),
        "ocean_batch_device_ptr": (vp, [vp]),
        "ocean_batch_normals_device_ptr": (vp, [vp]),
        "ocean_read_batch_normals": (i32, [vp, i32, vp]),
        "ocean_read_batch_displacement": (i32, [vp, i32, vp]),
        "ocean_time_frame_batch": (i32, [vp, i32, i32, f32, f32, ctypes.POINTER(f32)]),
        "ocean_set_quirks": (i32, [vp, ctypes.c_uint32]),
        "ocean_quirks": (ctypes.c_uint32, [vp]),
        "ocean_set_intermediate": (i32, [vp, i32]),
        "ocean_intermediate": (i32, [vp]),
        "ocean_normals": (i32, [vp, i32, vp]),
        "ocean_read_normals": (i32, [vp, vp]),
        "ocean_set_frame_normals": (i32, [vp, i32]),
        "ocean_frame_normals": (i32, [vp]),
        "ocean_normals_device_ptr": (vp, [vp]),
        "ocean_positions": (i32, [vp, i32, f32, f32, vp]),
        "ocean_read_positions": (i32, [vp, vp]),
        "ocean_checksum_displacement": (i32, [vp, vp, ctypes.POINTER(ctypes.c_uint64)]),
        "ocean_packed_bytes": (ctypes.c_int64, [vp, i32]),
        "ocean_pack_displacement": (i32, [vp, i32, vp, vp]),
        "ocean_read_displacement": (i32, [vp, vp]),
        "ocean_read_field": (i32, [vp, i32, vp]),
        "ocean_write_field": (i32, [vp, i32, vp]),
        "ocean_displacement_device_ptr": (vp, [vp]),
        "ocean_bind_displacement": (i32, [vp, vp]),
        "ocean_bind_displacement_fd": (i32, [vp, i32, ctypes.c_uint64, ctypes.c_uint64]),
        "ocean_stream": (vp, [vp]),
        "ocean_time_frames": (i32, [vp, i32, f32, f32, ctypes.POINTER(f32)]),
        "ocean_time_frame_batches": (i32, [vp, i32, i32, f32, f32, ctypes.POINTER(f32)]),
        "ocean_frame_times": (i32, [vp, i32, f32, f32, ctypes.POINTER(f32), ctypes.POINTER(f32), ctypes.POINTER(f32)]),
        "ocean_frame_times_ex": (i32, [vp, i32, f32, f32, ctypes.POINTER(f32), ctypes.POINTER(f32), ctypes.POINTER(f32), ctypes.POINTER(f32)]),
        "ocean_profile_frame": (i32, [vp, f32, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(f32),
                                      ctypes.POINTER(i32)]),
        "ocean_profile_staged": (i32, [vp, f32, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(f32),
                                       ctypes.POINTER(i32)]),
        "ocean_shard_create": (i32, [i32, i32, i32, i32, pp]),
        "ocean_shard_destroy": (None, [vp]),
        "ocean_shard_last_error": (ctypes.c_char_p, [vp]),
        "ocean_shard_upload": (i32, [vp, vp, vp, vp]),
        "ocean_shard_rows": (i32, [vp, ctypes.POINTER(PropagateLocalsC), vp, vp]),
        "ocean_shard_cols": (i32, [vp, vp, vp, vp]),
        "ocean_shard_sync": (i32, [vp]),
        "ocean_shard_stream": (vp, [vp]),
        "ocean_tile_exchange_bytes": (ctypes.c_int64, [vp, i32]),
        "ocean_tile_pass1": (i32, [vp, ctypes.POINTER(PropagateLocalsC), i32, i32, i32, i32, vp, vp]),
        "ocean_tile_pass2": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L
