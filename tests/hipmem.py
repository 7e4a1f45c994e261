"""Test helper: raw device buffers through the HIP runtime the product library already loaded (no torch in the tests)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
    return _hip


class Dev:
    """A device copy of a numpy array (or an uninitialised buffer of its shape)."""

    def __init__(self, arr, fill=None):
        self.shape, self.dtype, self.nbytes = arr.shape, arr.dtype, max(arr.nbytes, 4)
        self.p = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.p), self.nbytes) == 0
        if fill is not None:
            assert hip().hipMemset(self.p, fill, self.nbytes) == 0
        else:
            self.put(arr)

    def put(self, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.shape == self.shape
        assert hip().hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, out.nbytes, 2) == 0
        return out

    def free(self):
        hip().hipFree(self.p)
