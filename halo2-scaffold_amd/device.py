"""Device-resident buffers (h2mi_malloc) so vectors stay in HBM between hot-path calls."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class DevBuf:
    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib.h2mi_malloc(int(nbytes), C.byref(p)), "h2mi_malloc")
        self.ptr = p.value
        self.nbytes = int(nbytes)

    @classmethod
    def from_numpy(cls, arr: np.ndarray) -> "DevBuf":
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        check(lib.h2mi_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes), "h2d")
        return b

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(lib.h2mi_memcpy_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes), "h2d")

    def patch(self, arr: np.ndarray, offset: int = 0):
        """stream-ordered small upload that does not wait for the device (h2mi_memcpy_h2d_async)"""
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes and arr.nbytes <= 4096
        check(lib.h2mi_memcpy_h2d_async(self.ptr + offset, arr.ctypes.data, arr.nbytes), "h2d_async")

    def to_numpy(self, dtype=np.uint64, shape=None, nbytes=None, offset=0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(lib.h2mi_memcpy_d2h(out.ctypes.data, self.ptr + offset, nbytes), "d2h")
        return out.reshape(shape) if shape is not None else out

    def copy_from(self, other: "DevBuf", nbytes=None):
        nbytes = min(self.nbytes, other.nbytes) if nbytes is None else nbytes
        check(lib.h2mi_memcpy_d2d(self.ptr, other.ptr, nbytes), "d2d")

    def free(self):
        if self.ptr:
            lib.h2mi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SideStream:
    """a second device stream (h2mi_stream_create) for work that need not queue behind the library stream's chain:
    `after_library()` orders it behind everything issued so far on the library stream (e.g. the column uploads it
    reads), `join_library()` makes the library stream wait for it.  `.handle` goes into the `stream` argument of the
    *_dev entry points."""

    def __init__(self):
        h = C.c_void_p()
        check(lib.h2mi_stream_create(C.byref(h)), "stream_create")
        self.handle = h.value

    def after_library(self):
        check(lib.h2mi_stream_wait(self.handle, None), "stream_wait")

    def join_library(self):
        check(lib.h2mi_stream_wait(None, self.handle), "stream_wait")

    def free(self):
        if self.handle:
            lib.h2mi_stream_destroy(self.handle)
            self.handle = None
