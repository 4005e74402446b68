"""Device buffers for tests that hand DEVICE pointers to the C ABI: a few HIP runtime calls
through ctypes (hipMalloc / hipFree / hipMemcpy / hipMemGetInfo).  Test plumbing only.  The
runtime is the one libpose_amd.so already brought into the process."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from pose_estimation_amd import _lib
        _lib.load()                                   # loads libamdhip64 as a dependency first
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return _hip


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")


class DeviceArray:
    """A device copy of a numpy array (or an uninitialised buffer of its shape)."""

    def __init__(self, host):
        self.host = np.ascontiguousarray(host)
        self.ptr = C.c_void_p()
        _chk(hip().hipMalloc(C.byref(self.ptr), max(self.host.nbytes, 8)), "hipMalloc")
        _chk(hip().hipMemcpy(self.ptr, self.host.ctypes.data_as(C.c_void_p), self.host.nbytes, 1), "hipMemcpy H2D")

    def get(self):
        out = np.empty_like(self.host)
        _chk(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def free_bytes():
    f, t = C.c_size_t(0), C.c_size_t(0)
    _chk(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
    _chk(hip().hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
    return f.value
