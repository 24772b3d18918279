"""Wire encodings of field elements and G1 points: Fr::to_repr / from_repr and
G1Affine::to_bytes / from_bytes as create_proof's transcript and ParamsKZG::{write,read} use them
[RECALL halo2curves 0.3.x — conventions listed in csrc/h2mi_serde.hip].  Single values are converted on the
host (a proof writes tens of them); vectors go through the device kernels."""
import ctypes as C

import numpy as np

from . import field as F
from ._lib import check, lib
from .device import DevBuf


class DecodeError(ValueError):
    """an encoding from_repr / from_bytes rejects (the crate returns CtOption::none -> io::Error)."""


def fr_to_repr(limbs) -> bytes:
    """Fr (4 Montgomery limbs) -> 32 little-endian canonical bytes."""
    return F.fr_from_mont_limbs(np.asarray(limbs, dtype=np.uint64).reshape(4)).to_bytes(32, "little")


def fr_from_repr(b: bytes) -> np.ndarray:
    v = int.from_bytes(b, "little")
    if len(b) != 32 or v >= F.FR_MODULUS:
        raise DecodeError("Fr encoding out of range")
    return F.fr_to_mont_limbs(v)


def fr_from_bytes_wide(b: bytes) -> np.ndarray:
    """Fr::from_bytes_wide / from_uniform_bytes: 64 little-endian bytes reduced mod r (Challenge255)."""
    assert len(b) == 64
    return F.fr_to_mont_limbs(int.from_bytes(b, "little") % F.FR_MODULUS)


def g1_to_bytes(affine: np.ndarray) -> np.ndarray:
    """(n, 8) u64 affine points (Montgomery; (0,0) = identity) -> (n, 32) bytes."""
    a = np.ascontiguousarray(affine, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros((len(a), 32), dtype=np.uint8)
    check(lib.h2mi_g1_compress(a.ctypes.data, len(a), out.ctypes.data), "g1_compress")
    return out


def g1_from_bytes(data) -> np.ndarray:
    """(n, 32) bytes -> (n, 8) u64 affine points; raises DecodeError if any encoding is invalid."""
    d = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((len(d), 8), dtype=np.uint64)
    bad = C.c_uint64()
    check(lib.h2mi_g1_decompress(d.ctypes.data, len(d), out.ctypes.data, C.byref(bad)), "g1_decompress")
    if bad.value:
        raise DecodeError(f"{bad.value} invalid G1 encodings")
    return out


def g1_to_bytes_dev(d_affine: DevBuf, n: int, d_out: DevBuf = None) -> DevBuf:
    d_out = d_out or DevBuf(n * 32)
    check(lib.h2mi_g1_compress_dev(d_affine.ptr, n, d_out.ptr, None), "g1_compress_dev")
    return d_out


def g1_from_bytes_dev(d_in: DevBuf, n: int, d_out: DevBuf = None) -> DevBuf:
    d_out = d_out or DevBuf(n * 64)
    bad = C.c_uint64()
    check(lib.h2mi_g1_decompress_dev(d_in.ptr, n, d_out.ptr, C.byref(bad)), "g1_decompress_dev")
    if bad.value:
        raise DecodeError(f"{bad.value} invalid G1 encodings")
    return d_out
