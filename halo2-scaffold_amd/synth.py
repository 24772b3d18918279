"""Synthetic prover inputs: counter-based SplitMix64, one word per (seed, index).

Element i of a vector is limbs splitmix64(seed<<32 | 4i+j), j = 0..3, top limb masked to 62 bits and
reduced once; the limbs ARE the in-memory Montgomery representation (a uniform Montgomery
representative is a uniform field element).  "witness-like": 90 % zero, 5 % in {0,1}, 5 % uniform —
the shape of real halo2 advice columns (mostly empty rows and boolean cells).
"""
import numpy as np

from .field import FR_MODULUS, fr_to_mont_limbs

SEED = 0x48324D49


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) << np.uint64(32)) + idx.astype(np.uint64)
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _cond_sub(limbs: np.ndarray, mod: int) -> np.ndarray:
    m = [np.uint64((mod >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    ge = np.ones(len(limbs), dtype=bool)
    decided = np.zeros(len(limbs), dtype=bool)
    for i in (3, 2, 1, 0):
        gt, lt = limbs[:, i] > m[i], limbs[:, i] < m[i]
        ge = np.where(~decided & lt, False, ge)
        decided |= gt | lt
    out = limbs.copy()
    borrow = np.zeros(len(limbs), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(4):
            a = limbs[:, i]
            d = a - m[i]
            b1 = (a < m[i]).astype(np.uint64)
            d2 = d - borrow
            b2 = (d < borrow).astype(np.uint64)
            out[:, i] = np.where(ge, d2, a)
            borrow = b1 | b2
    return out


def uniform_fr(n: int, seed: int = SEED, start: int = 0) -> np.ndarray:
    idx = np.arange(4 * start, 4 * (start + n), dtype=np.uint64)
    w = _splitmix64(idx, seed).reshape(n, 4)
    w[:, 3] &= np.uint64((1 << 62) - 1)
    return _cond_sub(w, FR_MODULUS)


def witness_like_fr(n: int, seed: int = SEED) -> np.ndarray:
    u = uniform_fr(n, seed)
    ar = np.arange(n, dtype=np.uint64)
    sel = _splitmix64(ar, seed ^ 0x5EED) % np.uint64(100)
    bit = _splitmix64(ar, seed ^ 0xB175) & np.uint64(1)
    out = np.zeros((n, 4), dtype=np.uint64)
    out[(sel >= 90) & (sel < 95) & (bit == 1)] = fr_to_mont_limbs(1)
    uni = sel >= 95
    out[uni] = u[uni]
    return out


def circuit_like_fr(n: int, seed: int = SEED, used_rows: int = 3, blinding_rows: int = 5) -> np.ndarray:
    """an advice column of the reference's StandardPlonk circuit at 2^k rows (src/circuits/standard_plonk.rs:
    83-108): `used_rows` assigned cells at the top, zeros below, random blinding factors in the last rows."""
    out = np.zeros((n, 4), dtype=np.uint64)
    u = uniform_fr(used_rows + blinding_rows, seed)
    out[: min(used_rows, n)] = u[: min(used_rows, n)]
    if n > blinding_rows:
        out[n - blinding_rows :] = u[used_rows:]
    return out
