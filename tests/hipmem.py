"""Raw device memory for the GPU tests, straight from the HIP runtime libocean_hip.so is linked against (ctypes on
libamdhip64: no torch in the test process, so there is exactly one HIP runtime in it)."""
import ctypes

import numpy as np

_HIP = None


def hip():
    global _HIP
    if _HIP is None:
        import gfx_ocean_amd as g
        g.load_library()                                    # pulls in /opt/rocm's libamdhip64 first
        # the very file that is already mapped (dlopen by soname could resolve to another copy of the runtime, which
        # then finds no device: the process already holds the first one's)
        paths = []
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line and line.split()[-1] not in paths:
                    paths.append(line.split()[-1])
        assert paths, "libocean_hip.so did not pull in libamdhip64"
        # a test that imported torch has mapped torch's bundled copy as well (same soname, second runtime): ours is the
        # one libocean_hip.so is linked against, /opt/rocm's
        ours = [q for q in paths if "/torch/" not in q]
        path = (ours or paths)[0]
        _HIP = ctypes.CDLL(path)
        _HIP.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _HIP.hipFree.argtypes = [ctypes.c_void_p]
        _HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _HIP.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _HIP.hipDeviceSynchronize.argtypes = []
    return _HIP


def free_bytes() -> int:
    """hipMemGetInfo's free figure for the current device (after everything enqueued has completed)."""
    assert hip().hipDeviceSynchronize() == 0
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip().hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    assert hip().hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return int(free.value)


class DeviceBuffer:
    """hipMalloc'ed bytes; .ptr is the device address the C ABI takes."""

    def __init__(self, nbytes: int):
        p = ctypes.c_void_p()
        st = hip().hipMalloc(ctypes.byref(p), int(nbytes))
        assert st == 0, f"hipMalloc({nbytes}) -> {st}"
        self.ptr, self.nbytes = p.value, int(nbytes)

    def to_host(self, dtype=np.float32, nbytes=None) -> np.ndarray:
        nbytes = self.nbytes if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype)
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data, self.ptr, nbytes, 2) == 0          # hipMemcpyDeviceToHost
        return out

    def from_host(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        assert hip().hipMemcpy(self.ptr + offset, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def copy_from_device(self, src_ptr: int, nbytes: int, offset: int = 0):
        """Device-to-device, complete on return.  (hipMemcpy D2D only ENQUEUES on the null stream, and the library's
        streams are non-blocking: without the wait a kernel launched next can overtake the copy.)"""
        assert offset + nbytes <= self.nbytes
        assert hip().hipMemcpy(self.ptr + offset, src_ptr, nbytes, 3) == 0          # hipMemcpyDeviceToDevice
        assert hip().hipDeviceSynchronize() == 0

    def fill(self, byte: int = 0):
        assert hip().hipMemset(self.ptr, byte, self.nbytes) == 0
        assert hip().hipDeviceSynchronize() == 0

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedHostBuffer:
    """hipHostMalloc'ed bytes: host memory registered with the runtime, which kernels read in place over the bus (what a mapped
    staging buffer is to the reference, src/render.rs:749-761).  .ptr is the address; .view(dtype) the host's window on it."""

    def __init__(self, nbytes: int):
        p = ctypes.c_void_p()
        hip().hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        hip().hipHostFree.argtypes = [ctypes.c_void_p]
        st = hip().hipHostMalloc(ctypes.byref(p), int(nbytes), 0)
        assert st == 0 and p.value, f"hipHostMalloc({nbytes}) -> {st}"
        self.ptr, self.nbytes = int(p.value), int(nbytes)

    def view(self, dtype=np.float32) -> np.ndarray:
        return np.frombuffer((ctypes.c_char * self.nbytes).from_address(self.ptr), dtype=dtype)

    def free(self):
        if self.ptr:
            hip().hipHostFree(ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class Stream:
    """A caller-owned hipStream_t (non-blocking, like the library's own): .handle is what the C ABI takes as `stream`."""

    def __init__(self):
        h = ctypes.c_void_p()
        hip().hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        hip().hipStreamDestroy.argtypes = [ctypes.c_void_p]
        hip().hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        assert hip().hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0          # hipStreamNonBlocking
        self.handle = ctypes.c_void_p(h.value)

    def synchronize(self):
        assert hip().hipStreamSynchronize(self.handle) == 0

    def destroy(self):
        if self.handle:
            hip().hipStreamDestroy(self.handle)
            self.handle = None


class CapturedGraph:
    """What `record(stream_handle)` launches on a capturing stream, as an instantiated hipGraph: .launch() replays it on the stream
    it was captured on.  (hipStreamBeginCapture / EndCapture / hipGraphInstantiate / hipGraphLaunch through ctypes.)"""

    def __init__(self, stream: Stream, record):
        h = hip()
        h.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
        h.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        h.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        h.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        h.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
        h.hipGraphDestroy.argtypes = [ctypes.c_void_p]
        h.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        self.stream = stream
        assert h.hipStreamBeginCapture(stream.handle, 2) == 0               # hipStreamCaptureModeRelaxed
        try:
            record(stream.handle)
        finally:
            g_ = ctypes.c_void_p()
            st = h.hipStreamEndCapture(stream.handle, ctypes.byref(g_))
        assert st == 0 and g_.value, f"hipStreamEndCapture -> {st}"
        self.graph = g_
        count = ctypes.c_size_t()
        assert h.hipGraphGetNodes(self.graph, None, ctypes.byref(count)) == 0
        self.nodes = int(count.value)
        e = ctypes.c_void_p()
        st = h.hipGraphInstantiate(ctypes.byref(e), self.graph, None, None, 0)
        assert st == 0 and e.value, f"hipGraphInstantiate -> {st}"
        self.exec = e

    def launch(self):
        assert hip().hipGraphLaunch(self.exec, self.stream.handle) == 0

    def destroy(self):
        if self.exec:
            hip().hipGraphExecDestroy(self.exec)
            hip().hipGraphDestroy(self.graph)
            self.exec = self.graph = None
