"""ctypes wrapper of oracle/libh2ref.so (the C restatement, h2ref.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libh2ref.so")


def build(force=False):
    src = os.path.join(_HERE, "h2ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        # -march=native binaries do not travel between hosts: rebuild when the source is newer, and
        # fall back to a portable build if the native one cannot be loaded.
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def use_native():
    """(bench.py's cpu_baseline leg) rebuild with -march=native ON THIS HOST and switch to that library, so
    the timed CPU baseline gets the host's mulx/adx code generation; never shipped between machines."""
    global _lib
    so = os.path.join(_HERE, "libh2ref_native.so")
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libh2ref_native.so"])
    _lib = None
    lib(path=so)


def lib(path=None):
    global _lib
    if _lib is None:
        if path is None:
            build()
            path = _SO
        try:
            _lib = C.CDLL(path)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
        vp, sz = C.c_void_p, C.c_size_t
        _lib.h2ref_msm.argtypes = [vp, vp, sz, C.c_int, vp]
        _lib.h2ref_ntt.argtypes = [vp, vp, C.c_uint32, C.c_int]
        _lib.h2ref_field_op.argtypes = [C.c_int, C.c_int, vp, vp, vp, sz]
        _lib.h2ref_g1_mul_gen.argtypes = [vp, sz, C.c_int, vp]
        _lib.h2ref_normalize.argtypes = [vp, sz, vp]
        _lib.h2ref_sum.argtypes = [vp, sz, vp]
        for f in ("h2ref_msm", "h2ref_ntt", "h2ref_field_op", "h2ref_g1_mul_gen", "h2ref_normalize", "h2ref_sum"):
            getattr(_lib, f).restype = None
    return _lib


def msm(scalars: np.ndarray, bases: np.ndarray, threads: int = 1) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    assert len(scalars) == len(bases)
    out = np.zeros(12, dtype=np.uint64)
    lib().h2ref_msm(scalars.ctypes.data, bases.ctypes.data, len(scalars), threads, out.ctypes.data)
    return out


def ntt(a: np.ndarray, omega: np.ndarray, log_n: int, threads: int = 1) -> None:
    assert a.flags["C_CONTIGUOUS"] and a.dtype == np.uint64 and len(a) == 1 << log_n
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib().h2ref_ntt(a.ctypes.data, omega.ctypes.data, log_n, threads)


def field_op(field: int, op: int, a: np.ndarray, b: np.ndarray = None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64)
        bp = b.ctypes.data
    lib().h2ref_field_op(field, op, a.ctypes.data, bp, out.ctypes.data, len(a))
    return out


def g1_mul_gen(scalars: np.ndarray, threads: int = 1) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros((len(scalars), 8), dtype=np.uint64)
    lib().h2ref_g1_mul_gen(scalars.ctypes.data, len(scalars), threads, out.ctypes.data)
    return out


def normalize(jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((len(jac), 8), dtype=np.uint64)
    lib().h2ref_normalize(jac.ctypes.data, len(jac), out.ctypes.data)
    return out


def g1_sum(jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    lib().h2ref_sum(jac.ctypes.data, len(jac), out.ctypes.data)
    return out
