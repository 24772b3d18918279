"""Quotient step of create_proof for the reference's StandardPlonk circuit, on the device.

Mirror of halo2_proofs plonk/evaluation.rs `evaluate_h` followed by `divide_by_vanishing_poly`, specialised
to the constraint system of reference src/circuits/standard_plonk.rs:29-48 (one degree-3 gate, equality on
a, b, c => three permutation sets of one column).  Inputs are extended-domain DevBufs; the per-domain
scalars (coset generator, DELTA, the few inverses of X^n - 1 on the coset) are computed on the host as
EvaluationDomain::new / keygen do.
"""
import ctypes as C

import numpy as np

from . import field as F
from ._lib import check, lib
from ._lib import check as _check
from .device import DevBuf
from .domain import EvaluationDomain

FR_DELTA = pow(F.FR_MULTIPLICATIVE_GENERATOR, 1 << F.FR_S, F.FR_MODULUS)  # halo2curves Fr::DELTA
BLINDING_FACTORS = 5  # ConstraintSystem::blinding_factors() for this circuit


class _Cosets(C.Structure):
    _fields_ = [("advice", C.c_void_p * 3), ("fixed", C.c_void_p * 5), ("sigma", C.c_void_p * 3), ("z", C.c_void_p * 3),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active", C.c_void_p)]


def vanishing_inverses(domain: EvaluationDomain) -> np.ndarray:
    """(X^n - 1)^-1 on the extended coset: 2^(extended_k - k) distinct values."""
    r = F.FR_MODULUS
    rot = 1 << (domain.extended_k - domain.k)
    vals = []
    for i in range(rot):
        X = domain.g_coset * pow(domain.extended_omega, i, r) % r
        vals.append(F.fr_inv((pow(X, domain.n, r) - 1) % r))
    return np.stack([F.fr_to_mont_limbs(v) for v in vals])


def evaluate_h(domain: EvaluationDomain, advice, fixed, sigma, z, l0: DevBuf, l_last: DevBuf, l_active: DevBuf, beta: int, gamma: int, y: int,
               out: DevBuf, stream=None) -> None:
    """h(X) on the extended coset (already divided by X^n - 1) into `out`; all operands are DevBufs of
    extended_len() elements."""
    assert len(advice) == 3 and len(fixed) == 5 and len(sigma) == 3 and len(z) == 3
    cs = _Cosets()
    for i in range(3):
        cs.advice[i], cs.sigma[i], cs.z[i] = advice[i].ptr, sigma[i].ptr, z[i].ptr
    for i in range(5):
        cs.fixed[i] = fixed[i].ptr
    cs.l0, cs.l_last, cs.l_active = l0.ptr, l_last.ptr, l_active.ptr
    m = F.fr_to_mont_limbs
    t_inv = np.ascontiguousarray(vanishing_inverses(domain))
    args = [m(beta), m(gamma), m(y), m(FR_DELTA), m(domain.g_coset), m(domain.extended_omega)]
    check(lib.h2mi_plonk_evaluate_h_standard_dev(C.byref(cs), domain.k, domain.extended_k, BLINDING_FACTORS, *[a.ctypes.data for a in args],
                                                 t_inv.ctypes.data, out.ptr, stream), "evaluate_h")


def permutation_product(k: int, values, sigmas, column_indices, beta: int, gamma: int, usable_rows: int, d_z: DevBuf,
                        d_start: DevBuf = None, d_last: DevBuf = None) -> None:
    """One chunk of the permutation argument's grand product on the device (plonk/permutation/prover.rs): fills
    d_z[0 .. usable_rows] from the chunk's columns (`values`, Lagrange basis) and their permutation columns
    (`sigmas`); `column_indices[j]` is column j's position in the argument (its identity image is
    delta^index * omega^row).  The blinding rows of d_z are left as the caller set them."""
    m = len(values)
    assert m == len(sigmas) == len(column_indices) and 1 <= m <= 64
    r = F.FR_MODULUS
    vp = (C.c_void_p * m)(*[b.ptr for b in values])
    sp = (C.c_void_p * m)(*[b.ptr for b in sigmas])
    bd = np.ascontiguousarray(np.stack([F.fr_to_mont_limbs(beta * pow(FR_DELTA, int(c), r) % r) for c in column_indices]))
    b_, g_ = F.fr_to_mont_limbs(beta), F.fr_to_mont_limbs(gamma)
    w = F.fr_to_mont_limbs(F.omega_for(k))
    check(lib.h2mi_plonk_permutation_product_dev(vp, sp, m, k, usable_rows, b_.ctypes.data, g_.ctypes.data, bd.ctypes.data, w.ctypes.data,
                                                 d_start.ptr if d_start is not None else None, d_z.ptr,
                                                 d_last.ptr if d_last is not None else None, None), "permutation_product")


class ActiveRows:
    """keygen-time support of the copy constraints for h2mi_plonk_permutation_products_sparse_dev: the sorted positions
    set * usable_rows + row at which some column of the set is moved by the permutation (`mapping`: the non-identity
    entries of permutation/keygen.rs `Assembly`, keyed (column index in the argument, row)).  Rows the permutation leaves
    alone contribute num / den = 1 exactly, so the grand products only change at these positions."""

    def __init__(self, mapping, chunk_len: int, usable_rows: int):
        pos = sorted({(col // chunk_len) * usable_rows + row for (col, row), target in mapping.items() if target != (col, row) and row < usable_rows})
        self.count = len(pos)
        self.buf = DevBuf.from_numpy(np.array(pos if pos else [0], dtype=np.uint32))

    def free(self):
        self.buf.free()


def permutation_products(k: int, values, sigmas, chunk_len: int, beta: int, gamma: int, usable_rows: int, d_zs, active: ActiveRows = None) -> None:
    """Every set of the permutation argument in one device pass (plonk/permutation/prover.rs `Argument::commit`):
    `values` / `sigmas` are the equality-enabled columns in argument order, chunked by `chunk_len` = cs.degree() - 2;
    d_zs[s] receives rows 0 .. usable_rows of set s, chained through the previous set's last value.  With `active`
    (ActiveRows from keygen) the products are computed over the constrained positions only — same values."""
    m = len(values)
    sets = -(-m // chunk_len)
    assert m == len(sigmas) and 1 <= m <= 64 and len(d_zs) == sets
    r = F.FR_MODULUS
    vp = (C.c_void_p * m)(*[b.ptr for b in values])
    sp = (C.c_void_p * m)(*[b.ptr for b in sigmas])
    zp = (C.c_void_p * sets)(*[b.ptr for b in d_zs])
    bd = np.ascontiguousarray(np.stack([F.fr_to_mont_limbs(beta * pow(FR_DELTA, j, r) % r) for j in range(m)]))
    b_, g_ = F.fr_to_mont_limbs(beta), F.fr_to_mont_limbs(gamma)
    w = F.fr_to_mont_limbs(F.omega_for(k))
    if active is not None and active.count * 8 <= sets * usable_rows:  # dense supports gain nothing from the indirection
        check(lib.h2mi_plonk_permutation_products_sparse_dev(vp, sp, m, chunk_len, k, usable_rows, b_.ctypes.data, g_.ctypes.data, bd.ctypes.data,
                                                             w.ctypes.data, active.buf.ptr, active.count, zp, None), "permutation_products_sparse")
        return
    check(lib.h2mi_plonk_permutation_products_dev(vp, sp, m, chunk_len, k, usable_rows, b_.ctypes.data, g_.ctypes.data, bd.ctypes.data, w.ctypes.data,
                                                  zp, None), "permutation_products")


# ---- lookup argument, single-expression lookups (the range check of the reference's RangeWithInstanceCircuitBuilder,
# src/scaffold.rs:434-485 with LOOKUP_BITS, :44-48) -----------------------------------------------------------------------
class LookupTable:
    """keygen-time description of a fixed lookup table column for the device's counting sort: the distinct values of
    its usable rows in ascending (canonical integer) order, their multiplicities, uploaded once."""

    def __init__(self, table_values, usable_rows: int):
        """table_values: the fixed column as integers (as the circuit assigns them), at least usable_rows of them"""
        counts = {}
        for v in table_values[:usable_rows]:
            v %= F.FR_MODULUS
            counts[v] = counts.get(v, 0) + 1
        vals = sorted(counts)
        self.n_unique = len(vals)
        self.usable_rows = usable_rows
        can = np.zeros((self.n_unique, 4), dtype=np.uint64)
        mont = np.zeros((self.n_unique, 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            can[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
            mont[i] = F.fr_to_mont_limbs(v)
        self.sorted_canonical = DevBuf.from_numpy(can)
        self.sorted_mont = DevBuf.from_numpy(mont)
        self.mult = DevBuf.from_numpy(np.array([counts[v] for v in vals], dtype=np.uint32))

    def free(self):
        for b in (self.sorted_canonical, self.sorted_mont, self.mult):
            b.free()


def lookup_permute(k: int, d_input: DevBuf, table: LookupTable, d_permuted_input: DevBuf, d_permuted_table: DevBuf) -> int:
    """lookup/prover.rs permute_expression_pair on the device; rows beyond usable_rows are left to the caller (blinding).
    Returns the number of inputs that are not table values (the crate raises ConstraintSystemFailure when non-zero)."""
    missing = C.c_uint64()
    _check(lib.h2mi_plonk_lookup_permute_dev(d_input.ptr, table.sorted_canonical.ptr, table.sorted_mont.ptr, table.mult.ptr, table.n_unique, k,
                                             table.usable_rows, d_permuted_input.ptr, d_permuted_table.ptr, C.byref(missing), None),
           "lookup_permute")
    return missing.value


def lookup_product(k: int, d_input: DevBuf, d_table: DevBuf, d_permuted_input: DevBuf, d_permuted_table: DevBuf, beta: int, gamma: int,
                   usable_rows: int, d_z: DevBuf) -> None:
    b_, g_ = F.fr_to_mont_limbs(beta), F.fr_to_mont_limbs(gamma)
    _check(lib.h2mi_plonk_lookup_product_dev(d_input.ptr, d_table.ptr, d_permuted_input.ptr, d_permuted_table.ptr, k, usable_rows, b_.ctypes.data,
                                             g_.ctypes.data, d_z.ptr, None), "lookup_product")


class _RangeCosets(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lookup_advice", C.c_void_p), ("lookup_selector", C.c_void_p), ("q", C.c_void_p), ("table", C.c_void_p),
                ("perm_value", C.c_void_p * 4), ("perm_sigma", C.c_void_p * 4), ("perm_z", C.c_void_p * 4), ("lookup_permuted_input", C.c_void_p),
                ("lookup_permuted_table", C.c_void_p), ("lookup_z", C.c_void_p), ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active", C.c_void_p),
                ("n_perm", C.c_uint32), ("chunk_len", C.c_uint32), ("has_lookup", C.c_uint32)]


def evaluate_h_range(domain: EvaluationDomain, a: DevBuf, lookup_advice: DevBuf, q: DevBuf, table: DevBuf, perm_values, perm_sigmas, perm_zs,
                     lk_input: DevBuf, lk_table: DevBuf, lk_z: DevBuf, l0: DevBuf, l_last: DevBuf, l_active: DevBuf, beta: int, gamma: int, y: int,
                     out: DevBuf, blinding_factors: int = BLINDING_FACTORS, lookup_selector: DevBuf = None, chunk_len: int = None) -> None:
    """h(X) on the extended coset (already divided by X^n - 1) for the halo2-lib constraint systems.  Lookup input: the
    column `lookup_advice` (chunks of two by default), or `lookup_selector` * a (halo2-base with one advice column: degree
    5, chunks of three), or none (all lookup arguments None: degree 3, chunks of one)."""
    m = len(perm_values)
    has_lookup = lookup_advice is not None or lookup_selector is not None
    chunk = chunk_len if chunk_len is not None else (3 if lookup_selector is not None else 2 if has_lookup else 1)
    assert 1 <= m <= 4 and len(perm_sigmas) == m and len(perm_zs) == -(-m // chunk)
    cs = _RangeCosets()
    cs.a, cs.q = a.ptr, q.ptr
    cs.chunk_len, cs.has_lookup = chunk, 1 if has_lookup else 0
    if has_lookup:
        cs.table = table.ptr
        if lookup_selector is not None:
            cs.lookup_selector = lookup_selector.ptr
        else:
            cs.lookup_advice = lookup_advice.ptr
        cs.lookup_permuted_input, cs.lookup_permuted_table, cs.lookup_z = lk_input.ptr, lk_table.ptr, lk_z.ptr
    for j in range(m):
        cs.perm_value[j], cs.perm_sigma[j] = perm_values[j].ptr, perm_sigmas[j].ptr
    for s in range(len(perm_zs)):
        cs.perm_z[s] = perm_zs[s].ptr
    cs.l0, cs.l_last, cs.l_active = l0.ptr, l_last.ptr, l_active.ptr
    cs.n_perm = m
    mm = F.fr_to_mont_limbs
    t_inv = np.ascontiguousarray(vanishing_inverses(domain))
    args = [mm(beta), mm(gamma), mm(y), mm(FR_DELTA), mm(domain.g_coset), mm(domain.extended_omega)]
    _check(lib.h2mi_plonk_evaluate_h_range_dev(C.byref(cs), domain.k, domain.extended_k, blinding_factors, *[x.ctypes.data for x in args],
                                               t_inv.ctypes.data, out.ptr, None), "evaluate_h_range")


class _FlexCosets(C.Structure):
    """include/h2mi.h h2mi_flex_cosets"""
    G, P, L = 32, 64, 8  # H2MI_FLEX_MAX_GATES / _PERM / _LOOKUPS
    _fields_ = [("n_gates", C.c_uint32), ("gate_a", C.c_void_p * G), ("gate_q", C.c_void_p * G), ("n_perm", C.c_uint32), ("chunk_len", C.c_uint32),
                ("perm_value", C.c_void_p * P), ("perm_sigma", C.c_void_p * P), ("perm_z", C.c_void_p * P), ("n_lookups", C.c_uint32),
                ("lookup_input", C.c_void_p * L), ("lookup_input_b", C.c_void_p * L), ("lookup_table", C.c_void_p * L),
                ("lookup_permuted_input", C.c_void_p * L), ("lookup_permuted_table", C.c_void_p * L), ("lookup_z", C.c_void_p * L),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active", C.c_void_p)]


def evaluate_h_flex(domain: EvaluationDomain, gates, perm_values, perm_sigmas, perm_zs, chunk_len: int, lookups, l0: DevBuf, l_last: DevBuf,
                    l_active: DevBuf, beta: int, gamma: int, y: int, out: DevBuf, blinding_factors: int = BLINDING_FACTORS) -> None:
    """h(X) on the extended coset for the GENERAL halo2-base shapes (h2mi_plonk_evaluate_h_flex_dev): `gates` = [(advice coset, selector
    coset)] (<= 32 vertical gates), the permutation argument over <= 64 columns, `lookups` = [(input coset, second input factor or None,
    table coset, permuted input, permuted table, product)] (<= 8) — what builder.config() configures when one column overflows."""
    m = len(perm_values)
    assert 1 <= len(gates) <= _FlexCosets.G and m <= _FlexCosets.P and len(perm_sigmas) == m and len(perm_zs) == (-(-m // chunk_len) if m else 0) and len(lookups) <= _FlexCosets.L
    cs = _FlexCosets()
    cs.n_gates = len(gates)
    for g, (a, q) in enumerate(gates):
        cs.gate_a[g], cs.gate_q[g] = a.ptr, q.ptr
    cs.n_perm, cs.chunk_len = m, chunk_len
    for j in range(m):
        cs.perm_value[j], cs.perm_sigma[j] = perm_values[j].ptr, perm_sigmas[j].ptr
    for s, z in enumerate(perm_zs):
        cs.perm_z[s] = z.ptr
    cs.n_lookups = len(lookups)
    for l, (a_in, b_in, table, pin, ptab, z) in enumerate(lookups):
        cs.lookup_input[l], cs.lookup_table[l] = a_in.ptr, table.ptr
        cs.lookup_input_b[l] = b_in.ptr if b_in is not None else None
        cs.lookup_permuted_input[l], cs.lookup_permuted_table[l], cs.lookup_z[l] = pin.ptr, ptab.ptr, z.ptr
    cs.l0, cs.l_last, cs.l_active = l0.ptr, l_last.ptr, l_active.ptr
    mm = F.fr_to_mont_limbs
    t_inv = np.ascontiguousarray(vanishing_inverses(domain))
    args = [mm(beta), mm(gamma), mm(y), mm(FR_DELTA), mm(domain.g_coset), mm(domain.extended_omega)]
    _check(lib.h2mi_plonk_evaluate_h_flex_dev(C.byref(cs), domain.k, domain.extended_k, blinding_factors, *[x.ctypes.data for x in args],
                                              t_inv.ctypes.data, out.ptr, None), "evaluate_h_flex")
