"""keygen_vk / keygen_pk for the reference's StandardPlonk circuit, on the device.

Mirror of halo2_proofs::plonk::{keygen_vk, keygen_pk} as the reference calls them (examples/standard_plonk.rs:33-34,
src/scaffold.rs:132,135,284,287): synthesize the circuit without witnesses, turn the fixed columns and the copy
constraints (permutation/keygen.rs `Assembly::build_vk / build_pk`) into Lagrange columns, commit them
(ParamsKZG::commit_lagrange = one MSM each) and keep their coefficient and extended-coset forms plus l_0, l_last,
l_active for `evaluate_h`.  The columns are built where they are consumed: zero / power / fill kernels plus a few
32-byte patches for the assigned cells; nothing of size n crosses PCIe.

vk.transcript_repr stand-in: the crate hashes the Debug text of the pinned verifying key, which cannot be
reproduced without the crate; the same Blake2b-512 ("Halo2-Verify-Key") runs here over k, the constraint-system
degree and the compressed fixed / permutation commitments (see also oracle/prover.py).
"""
import ctypes as C
import hashlib
import struct

import numpy as np

from . import field as F
from . import serde
from ._lib import check, lib
from .circuits import PermutationAssembly
from .device import DevBuf
from . import plonk as gp
from .domain import EvaluationDomain
from .params import ParamsKZG

R = F.FR_MODULUS
FR_DELTA = pow(F.FR_MULTIPLICATIVE_GENERATOR, 1 << F.FR_S, R)  # halo2curves Fr::DELTA


def _m(v: int) -> np.ndarray:
    return F.fr_to_mont_limbs(v)


def _patch(buf: DevBuf, row: int, value: int):
    buf.upload(_m(value), offset=row * 32)


def commit_points(params: ParamsKZG, columns, lagrange: bool) -> np.ndarray:
    """commit every column (DevBufs of n elements) -> (len, 8) affine points on the host"""
    k = len(columns)
    out, aff = DevBuf(96 * k), DevBuf(64 * k)
    for i, c in enumerate(columns):
        params.commit_dev(c, out, lagrange=lagrange, out_offset=96 * i)
    check(lib.h2mi_join(), "join")
    check(lib.h2mi_g1_batch_normalize_dev(out.ptr, k, aff.ptr, None), "normalize")
    pts = aff.to_numpy(shape=(k, 8))
    out.free()
    aff.free()
    return pts


class VerifyingKey:
    def __init__(self, k: int, cs_degree: int, fixed_commitments: np.ndarray, permutation_commitments: np.ndarray):
        self.k = k
        self.cs_degree = cs_degree
        self.domain = EvaluationDomain(cs_degree, k)
        self.fixed_commitments = fixed_commitments
        self.permutation_commitments = permutation_commitments
        self.transcript_repr = self._transcript_repr()

    def to_bytes(self) -> bytes:
        pts = np.concatenate([self.fixed_commitments, self.permutation_commitments])
        return struct.pack("<II", self.k, self.cs_degree) + serde.g1_to_bytes(pts).tobytes()

    def _transcript_repr(self) -> int:
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        s = self.to_bytes()
        h.update(struct.pack("<Q", len(s)))
        h.update(s)
        return int.from_bytes(h.digest(), "little") % R


class _Columns:
    """Lagrange / coefficient / extended-coset forms of a group of columns"""

    def __init__(self, domain: EvaluationDomain, lagrange, keep_lagrange: bool):
        n, ext = domain.n, domain.extended_len()
        self.polys, self.cosets = [], []
        for col in lagrange:
            p, e = DevBuf(n * 32), DevBuf(ext * 32)
            domain.lagrange_to_coeff_oop_dev(col, p)
            domain.coeff_to_extended_oop_dev(p, e)
            self.polys.append(p)
            self.cosets.append(e)
        self.values = lagrange if keep_lagrange else None
        if not keep_lagrange:
            check(lib.h2mi_sync(), "sync")
            for col in lagrange:
                col.free()

    def free(self):
        for b in self.polys + self.cosets + (self.values or []):
            b.free()


def _fixed_columns(circuit, n: int):
    syn = circuit.without_witnesses().synthesize()
    cols = []
    for assigned in syn.fixed:
        d = DevBuf(n * 32)
        check(lib.h2mi_memset_zero(d.ptr, n * 32), "zero")
        for row, v in assigned.items():
            _patch(d, row, v)
        cols.append(d)
    return cols, syn


def _sigma_columns(circuit, syn, domain: EvaluationDomain):
    """permutation/keygen.rs build_vk / build_pk: sigma_j[i] = DELTA^(j') omega^(i') for (j', i') = mapping[(j, i)]"""
    n = domain.n
    asm = PermutationAssembly()
    for left, right in syn.copies:
        asm.copy(left, right)
    omega_pows = DevBuf(n * 32)
    check(lib.h2mi_fr_powers_dev(omega_pows.ptr, n, domain._omega.ctypes.data, None), "powers")
    cols = []
    for j, _ in enumerate(circuit.PERMUTATION_COLUMNS):
        d = DevBuf(n * 32)
        ptrs = (C.c_void_p * 1)(omega_pows.ptr)
        sc = _m(pow(FR_DELTA, j, R))
        check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, 1, n, d.ptr, None), "identity permutation")
        cols.append(d)
    for (col, row), (tc, tr) in asm.mapping.items():
        if (col, row) != (tc, tr):
            _patch(cols[col], row, pow(FR_DELTA, tc, R) * pow(domain.omega, tr, R) % R)
    check(lib.h2mi_sync(), "sync")
    omega_pows.free()
    return cols, asm.mapping


def keygen_vk(params: ParamsKZG, circuit) -> VerifyingKey:
    domain = EvaluationDomain(circuit.CS_DEGREE, params.k)
    fixed, syn = _fixed_columns(circuit, domain.n)
    sigma, _ = _sigma_columns(circuit, syn, domain)
    fc = commit_points(params, fixed, lagrange=True)
    pc = commit_points(params, sigma, lagrange=True)
    for b in fixed + sigma:
        b.free()
    return VerifyingKey(params.k, circuit.CS_DEGREE, fc, pc)


class ProvingKey:
    def __init__(self, vk: VerifyingKey, circuit, fixed: _Columns, permutation: _Columns, l0: DevBuf, l_last: DevBuf, l_active: DevBuf):
        self.vk = vk
        self.circuit = circuit
        self.fixed = fixed              # fixed_polys / fixed_cosets
        self.permutation = permutation  # permutations (Lagrange: the grand product reads them), polys, cosets
        self.l0, self.l_last, self.l_active = l0, l_last, l_active
        self.active_rows = None

    def get_vk(self) -> VerifyingKey:
        return self.vk

    def release(self):
        self.fixed.free()
        self.permutation.free()
        for b in (self.l0, self.l_last, self.l_active):
            b.free()
        if self.active_rows is not None:
            self.active_rows.free()


def keygen_pk(params: ParamsKZG, vk: VerifyingKey, circuit) -> ProvingKey:
    domain = vk.domain
    n, ext = domain.n, domain.extended_len()
    fixed, syn = _fixed_columns(circuit, n)
    sigma, mapping = _sigma_columns(circuit, syn, domain)
    fcols = _Columns(domain, fixed, keep_lagrange=False)
    pcols = _Columns(domain, sigma, keep_lagrange=True)
    # l_0, l_last (row n - blinding_factors - 1), l_active = 1 - (l_last + l_blind): ones on the usable rows
    u = n - (circuit.BLINDING_FACTORS + 1)
    lag = [DevBuf(n * 32) for _ in range(3)]
    for d in lag[:2]:
        check(lib.h2mi_memset_zero(d.ptr, n * 32), "zero")
    _patch(lag[0], 0, 1)
    _patch(lag[1], u, 1)
    one = _m(1)
    check(lib.h2mi_fr_fill_dev(lag[2].ptr, n, one.ctypes.data, None), "fill")
    check(lib.h2mi_memset_zero(lag[2].ptr + u * 32, (n - u) * 32), "zero")
    lcols = _Columns(domain, lag, keep_lagrange=False)
    for p in lcols.polys:
        p.free()
    pk = ProvingKey(vk, circuit, fcols, pcols, *lcols.cosets)
    # the rows the copy constraints touch: the only ones at which a permutation grand product changes
    pk.active_rows = gp.ActiveRows(mapping, circuit.CS_DEGREE - 2, u)
    return pk
