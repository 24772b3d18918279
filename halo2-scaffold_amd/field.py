"""Scalar-field / base-field constants and Montgomery packing used by the host layer.

These are the handful of per-domain constants EvaluationDomain::new computes on the CPU in the
reference stack (omega, omega^-1, n^-1, the coset generator); bulk arithmetic never happens here.
Constants: SURVEY.md 8a-0 (halo2curves::bn256::{Fr, Fq}, reference src/scaffold.rs:14).
"""
import numpy as np

FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_S = 28
FR_MULTIPLICATIVE_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_MULTIPLICATIVE_GENERATOR, (FR_MODULUS - 1) >> FR_S, FR_MODULUS)
FR_ZETA = pow(FR_MULTIPLICATIVE_GENERATOR, (FR_MODULUS - 1) // 3, FR_MODULUS)
_R = 1 << 256
_R_INV = pow(_R, -1, FR_MODULUS)


def fr_to_mont_limbs(a: int) -> np.ndarray:
    v = (a % FR_MODULUS) * _R % FR_MODULUS
    return np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint64).copy()  # a writable array of its own (callers keep it alive across the ABI call)


def fr_from_mont_limbs(l) -> int:
    v = int.from_bytes(np.ascontiguousarray(l, dtype=np.uint64).tobytes()[:32], "little")
    return v * _R_INV % FR_MODULUS


def fr_inv(a: int) -> int:
    return pow(a, -1, FR_MODULUS)


def omega_for(k: int) -> int:
    """generator of the 2^k-th roots of unity (EvaluationDomain::new: ROOT_OF_UNITY squared S-k times)."""
    if not 0 <= k <= FR_S:
        raise ValueError("log_n out of range")
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), FR_MODULUS)
