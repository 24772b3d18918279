"""Fr vectors over oracle/libh2ref.so for the large-size oracle prover (oracle/fastflex.py).

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  `FV` wraps an (n, 4) uint64 array of Montgomery limbs and overloads
+, -, *, % so that the SAME gate lambdas and term functions the Python-integer oracle runs per point
(oracle/flex.py: `cs.gates`, `_permutation_terms`, `_lookup_terms`) run once over a whole coset: an int operand is a
broadcast scalar, `% R` is the identity (every C helper returns fully reduced values).  The element-wise definitions are
halo2_proofs::arithmetic's (eval_polynomial, kate_division, the `parallelize` loops of evaluation.rs); each helper is
checked against oracle/bn254.py's integers in tests/test_oracle_fast.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import bn254 as o
from . import cref

R = o.R
THREADS = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)

_ready = False


def _lib():
    global _ready
    lib = cref.lib()
    if not _ready:
        vp, sz = C.c_void_p, C.c_size_t
        lib.h2ref_vec_binop.argtypes = [C.c_int, vp, vp, C.c_int, vp, sz, C.c_int]
        lib.h2ref_vec_dot.argtypes = [vp, vp, sz, vp]
        lib.h2ref_vec_horner.argtypes = [vp, sz, vp, vp]
        lib.h2ref_vec_kate.argtypes = [vp, sz, vp, vp]
        lib.h2ref_vec_powers.argtypes = [vp, vp, sz, vp]
        lib.h2ref_vec_running_product.argtypes = [vp, vp, sz, vp]
        lib.h2ref_vec_batch_inv.argtypes = [vp, vp, sz]
        for f in ("h2ref_vec_binop", "h2ref_vec_dot", "h2ref_vec_horner", "h2ref_vec_kate", "h2ref_vec_powers", "h2ref_vec_running_product",
                  "h2ref_vec_batch_inv"):
            getattr(lib, f).restype = None
        _ready = True
    return lib


def mont(v: int) -> np.ndarray:
    """int -> 4 Montgomery limbs"""
    return np.array(o.int_to_limbs(o.to_mont(v % R, R)), dtype=np.uint64)


def unmont(limbs) -> int:
    return o.from_mont(o.limbs_to_int(limbs), R)


class FV:
    __slots__ = ("a",)
    __array_priority__ = 1000

    def __init__(self, a: np.ndarray):
        assert a.dtype == np.uint64 and a.ndim == 2 and a.shape[1] == 4 and a.flags["C_CONTIGUOUS"]
        self.a = a

    # ---- constructors ----
    @classmethod
    def zeros(cls, n):
        return cls(np.zeros((n, 4), dtype=np.uint64))

    @classmethod
    def full(cls, n, v: int):
        return cls(np.ascontiguousarray(np.broadcast_to(mont(v), (n, 4))))

    @classmethod
    def from_ints(cls, vals):
        return cls(o.pack(list(vals), R))

    @classmethod
    def from_sparse(cls, n, cells: dict):
        out = cls.zeros(n)
        for r, v in cells.items():
            out.a[r] = mont(v)
        return out

    @classmethod
    def powers(cls, base: int, n: int, start: int = 1):
        out = np.empty((n, 4), dtype=np.uint64)
        b, s = mont(base), mont(start)
        _lib().h2ref_vec_powers(b.ctypes.data, s.ctypes.data, n, out.ctypes.data)
        return cls(out)

    # ---- access ----
    def __len__(self):
        return len(self.a)

    def copy(self):
        return FV(self.a.copy())

    def get(self, i) -> int:
        return unmont(self.a[i])

    def set(self, i, v: int):
        self.a[i] = mont(v)

    def to_ints(self):
        return o.unpack(self.a, R)

    def roll(self, shift):
        """out[i] = self[(i + shift) % n]"""
        return FV(np.ascontiguousarray(np.roll(self.a, -shift, axis=0)))

    def head(self, m):
        return FV(np.ascontiguousarray(self.a[:m]))

    def padded(self, m):
        out = np.zeros((m, 4), dtype=np.uint64)
        out[: len(self.a)] = self.a
        return FV(out)

    # ---- arithmetic ----
    def _bin(self, op, other):
        out = np.empty_like(self.a)
        if isinstance(other, FV):
            assert len(other) == len(self)
            _lib().h2ref_vec_binop(op, self.a.ctypes.data, other.a.ctypes.data, 0, out.ctypes.data, len(self.a), THREADS)
        else:
            s = mont(int(other))
            _lib().h2ref_vec_binop(op, self.a.ctypes.data, s.ctypes.data, 1, out.ctypes.data, len(self.a), THREADS)
        return FV(out)

    def __mul__(self, other):
        return self._bin(0, other)

    __rmul__ = __mul__

    def __add__(self, other):
        return self._bin(1, other)

    __radd__ = __add__

    def __sub__(self, other):
        return self._bin(2, other)

    def __rsub__(self, other):
        return self._bin(3, other)

    def __mod__(self, m):
        assert m == R
        return self

    # ---- reductions / scans ----
    def dot(self, other: "FV") -> int:
        out = np.zeros(4, dtype=np.uint64)
        m = min(len(self), len(other))
        _lib().h2ref_vec_dot(self.a.ctypes.data, other.a.ctypes.data, m, out.ctypes.data)
        return unmont(out)

    def eval(self, x: int) -> int:
        out = np.zeros(4, dtype=np.uint64)
        xm = mont(x)
        _lib().h2ref_vec_horner(self.a.ctypes.data, len(self.a), xm.ctypes.data, out.ctypes.data)
        return unmont(out)

    def kate_division(self, b: int) -> "FV":
        """quotient by (X - b) with the same length (top coefficient zero)"""
        out = np.empty_like(self.a)
        bm = mont(b)
        _lib().h2ref_vec_kate(self.a.ctypes.data, len(self.a), bm.ctypes.data, out.ctypes.data)
        return FV(out)

    def batch_inv(self) -> "FV":
        out = np.empty_like(self.a)
        _lib().h2ref_vec_batch_inv(self.a.ctypes.data, out.ctypes.data, len(self.a))
        return FV(out)

    def running_product(self, start: int, n_out: int) -> "FV":
        """out[0] = start, out[i + 1] = out[i] * self[i]"""
        out = np.empty((n_out, 4), dtype=np.uint64)
        s = mont(start)
        _lib().h2ref_vec_running_product(self.a.ctypes.data, s.ctypes.data, n_out, out.ctypes.data)
        return FV(out)

    def ntt(self, omega: int, log_n: int) -> "FV":
        """best_fft, out of place"""
        out = self.a.copy()
        cref.ntt(out, mont(omega), log_n, THREADS)
        return FV(out)


class FastDomain:
    """oracle/bn254.py Domain on FV vectors (poly/domain.rs EvaluationDomain restated there)"""

    def __init__(self, k, degree):
        self.d = o.Domain(k, degree)
        for name in ("k", "n", "extended_k", "omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv", "quotient_poly_degree"):
            setattr(self, name, getattr(self.d, name))
        self._coset_pows = None
        self._coset_inv_pows = None

    def lagrange_to_coeff(self, v: FV) -> FV:
        return v.ntt(self.d.omega_inv, self.k) * self.d.ifft_divisor

    def coeff_to_extended(self, v: FV) -> FV:
        size = 1 << self.extended_k
        if self._coset_pows is None:
            self._coset_pows = FV.powers(self.g_coset, self.n)
        return (v * self._coset_pows).padded(size).ntt(self.extended_omega, self.extended_k)

    def extended_to_coeff(self, v: FV) -> FV:
        c = v.ntt(self.extended_omega_inv, self.extended_k) * self.d.extended_ifft_divisor
        keep = self.n * self.quotient_poly_degree
        c = c.head(keep)
        if self._coset_inv_pows is None:
            self._coset_inv_pows = FV.powers(self.g_coset_inv, keep)
        return c * self._coset_inv_pows
