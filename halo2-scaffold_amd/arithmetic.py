"""Mirror of halo2_proofs::arithmetic::{best_multiexp, best_fft} over the C ABI.

Same names, argument meaning and error behaviour as the functions the reference reaches through
create_proof (reference examples/standard_plonk.rs:41-49): length mismatches raise (the Rust code
asserts), results are returned, nothing is computed on the CPU.
"""
import numpy as np

from ._lib import check, lib


def _fr_array(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("expected an (n, 4) uint64 array of Montgomery-form Fr limbs")
    return a


def best_multiexp(coeffs, bases) -> np.ndarray:
    """sum_i coeffs[i] * bases[i]; coeffs (n,4) u64 Fr, bases (n,8) u64 G1Affine -> (12,) u64 Jacobian.

    `bases` may also be a registered-bases handle (int) from ParamsKZG.
    """
    coeffs = _fr_array(coeffs)
    out = np.zeros(12, dtype=np.uint64)
    if isinstance(bases, (int, np.integer)):
        check(lib.h2mi_msm_bn254_g1(int(bases), None, coeffs.ctypes.data, len(coeffs), out.ctypes.data), "best_multiexp")
        return out
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    if bases.ndim != 2 or bases.shape[1] != 8:
        raise ValueError("expected an (n, 8) uint64 array of G1Affine limbs")
    if len(coeffs) != len(bases):  # assert_eq!(coeffs.len(), bases.len())
        raise AssertionError("coeffs.len() != bases.len()")
    check(lib.h2mi_msm_bn254_g1(0, bases.ctypes.data, coeffs.ctypes.data, len(coeffs), out.ctypes.data), "best_multiexp")
    return out


def best_fft(a: np.ndarray, omega, log_n: int) -> None:
    """in-place DFT of `a` ((n,4) u64 Fr) with root `omega` ((4,) u64 Fr); natural order in and out."""
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.ndim == 2 and a.shape[1] == 4):
        raise ValueError("a must be a C-contiguous (n,4) uint64 array (it is transformed in place)")
    if len(a) != (1 << log_n):  # assert_eq!(n, 1 << log_n)
        raise AssertionError("a.len() != 1 << log_n")
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    check(lib.h2mi_ntt_bn254_fr(a.ctypes.data, omega.ctypes.data, int(log_n)), "best_fft")


def eval_polynomial(poly: np.ndarray, point) -> np.ndarray:
    """halo2_proofs::arithmetic::eval_polynomial: sum_i poly[i] * point^i -> (4,) u64 Fr."""
    from .device import DevBuf

    poly = _fr_array(poly)
    point = np.ascontiguousarray(point, dtype=np.uint64)
    d = DevBuf.from_numpy(poly)
    o = DevBuf(32)
    check(lib.h2mi_fr_eval_poly_dev(d.ptr, len(poly), point.ctypes.data, o.ptr, None), "eval_polynomial")
    out = o.to_numpy(shape=(4,))
    d.free()
    o.free()
    return out


def kate_division(a: np.ndarray, b) -> np.ndarray:
    """halo2_proofs::arithmetic::kate_division: coefficients of a(X) / (X - b), remainder dropped."""
    from . import field as F
    from .device import DevBuf

    a = _fr_array(a)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    bv = F.fr_from_mont_limbs(b)
    if len(a) < 2:
        return np.zeros((0, 4), dtype=np.uint64)
    if bv == 0:  # division by X: shift
        return a[1:].copy()
    binv = F.fr_to_mont_limbs(F.fr_inv(bv))
    d = DevBuf.from_numpy(a)
    o = DevBuf((len(a) - 1) * 32)
    check(lib.h2mi_fr_kate_division_dev(d.ptr, len(a), b.ctypes.data, binv.ctypes.data, o.ptr, None), "kate_division")
    out = o.to_numpy(shape=(len(a) - 1, 4))
    d.free()
    o.free()
    return out


def lincomb(polys, scalars) -> np.ndarray:
    """sum_k scalars[k] * polys[k] (the challenge-weighted polynomial sums of the SHPLONK prover)."""
    import ctypes as C

    from .device import DevBuf

    polys = [_fr_array(p) for p in polys]
    n = len(polys[0])
    assert all(len(p) == n for p in polys) and 1 <= len(polys) <= 24
    sc = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.uint64) for s in scalars]))
    bufs = [DevBuf.from_numpy(p) for p in polys]
    ptrs = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
    o = DevBuf(n * 32)
    check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, len(bufs), n, o.ptr, None), "lincomb")
    out = o.to_numpy(shape=(n, 4))
    for b in bufs + [o]:
        b.free()
    return out
