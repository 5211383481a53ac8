"""A few HIP runtime calls over ctypes, for GPU tests that need device buffers and streams of their own without
pulling PyTorch into the test process.  Opened by soname, so it is the copy of the runtime librtmi.so already uses."""
import ctypes as C

import numpy as np

_hip = None

H2D, D2H = 1, 2


def hip():
    global _hip
    if _hip is None:
        from raytracing_amd import _lib
        _lib.load()  # brings libamdhip64.so.7 into the process
        L = C.CDLL("libamdhip64.so.7")
        vp, sz = C.c_void_p, C.c_size_t
        sig = {
            "hipMalloc": [C.POINTER(vp), sz],
            "hipFree": [vp],
            "hipMemcpy": [vp, vp, sz, C.c_int],
            "hipMemset": [vp, C.c_int, sz],
            "hipStreamCreate": [C.POINTER(vp)],
            "hipStreamDestroy": [vp],
            "hipStreamSynchronize": [vp],
            "hipDeviceSynchronize": [],
            "hipSetDevice": [C.c_int],
            "hipGetDevice": [C.POINTER(C.c_int)],
            "hipGetDeviceCount": [C.POINTER(C.c_int)],
            "hipMemGetInfo": [C.POINTER(sz), C.POINTER(sz)],
            "hipStreamBeginCapture": [vp, C.c_int],
            "hipStreamEndCapture": [vp, C.POINTER(vp)],
            "hipGraphInstantiate": [C.POINTER(vp), vp, vp, vp, sz],
            "hipGraphLaunch": [vp, vp],
            "hipGraphExecDestroy": [vp],
            "hipGraphDestroy": [vp],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = args
        _hip = L
    return _hip


def ok(rc):
    assert rc == 0, "HIP call failed with error %d" % rc


class DeviceBuffer:
    """hipMalloc'ed bytes with numpy upload / download."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ok(hip().hipMalloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        ok(hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, H2D))
        return b

    def fill(self, byte):
        ok(hip().hipMemset(self.ptr, int(byte), self.nbytes))

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.zeros(n, dtype=dtype)
        ok(hip().hipMemcpy(out.ctypes.data, self.ptr, n * dtype.itemsize, D2H))
        return out

    def free(self):
        if self.ptr:
            ok(hip().hipFree(self.ptr))
            self.ptr = None


class Stream:
    def __init__(self):
        p = C.c_void_p()
        ok(hip().hipStreamCreate(C.byref(p)))
        self.handle = p.value

    def synchronize(self):
        ok(hip().hipStreamSynchronize(self.handle))

    def destroy(self):
        if self.handle:
            ok(hip().hipStreamDestroy(self.handle))
            self.handle = None


def free_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    ok(hip().hipMemGetInfo(C.byref(free), C.byref(total)))
    return free.value
