"""keygen_vk / keygen_pk for the reference's StandardPlonk circuit — callers of the library's keygen.

Mirror of halo2_proofs::plonk::{keygen_vk, keygen_pk} as the reference calls them (examples/standard_plonk.rs:33-34,
src/scaffold.rs:132,135,284,287).  What happens HERE is what a Rust fork's caller does too: synthesize the circuit without
witnesses, hand its fixed cells and copy constraints (constrain_equal calls in order) to h2mi_prover_keygen together with the
constraint system as data (engine.ConstraintSystem).  The library (csrc/h2mi_prover.cpp) builds the Lagrange columns where they
are consumed, the sigma polynomials (permutation/keygen.rs Assembly), the commitments (ParamsKZG::commit_lagrange = one MSM each),
the coefficient and extended-coset forms, l_0 / l_last / l_active; nothing of size n crosses PCIe.

vk.transcript_repr stand-in: the crate hashes the Debug text of the pinned verifying key, which cannot be reproduced without
the crate; the same Blake2b-512 ("Halo2-Verify-Key") runs here over k, the constraint-system degree and the compressed fixed /
permutation commitments (see also oracle/prover.py).  It is the CALLER's value: the library never sees it.
"""
import hashlib
import struct

import numpy as np

from . import engine
from . import field as F
from . import serde
from .domain import EvaluationDomain
from .params import ParamsKZG

R = F.FR_MODULUS
FR_DELTA = pow(F.FR_MULTIPLICATIVE_GENERATOR, 1 << F.FR_S, R)  # halo2curves Fr::DELTA


def _m(v: int) -> np.ndarray:
    return F.fr_to_mont_limbs(v)


def transcript_repr(k: int, cs_degree: int, fixed_commitments: np.ndarray, permutation_commitments: np.ndarray):
    """(vk bytes, transcript_repr) of the stand-in above"""
    pts = np.concatenate([fixed_commitments, permutation_commitments])
    s = struct.pack("<II", k, cs_degree) + serde.g1_to_bytes(pts).tobytes()
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(struct.pack("<Q", len(s)))
    h.update(s)
    return s, int.from_bytes(h.digest(), "little") % R


class VerifyingKey:
    def __init__(self, k: int, cs_degree: int, fixed_commitments: np.ndarray, permutation_commitments: np.ndarray):
        self.k = k
        self.cs_degree = cs_degree
        self.domain = EvaluationDomain(cs_degree, k)
        self.fixed_commitments = fixed_commitments
        self.permutation_commitments = permutation_commitments
        self._bytes, self.transcript_repr = transcript_repr(k, cs_degree, fixed_commitments, permutation_commitments)

    def to_bytes(self) -> bytes:
        return self._bytes


def constraint_system(circuit, k: int) -> engine.ConstraintSystem:
    """StandardPlonkConfig::configure (reference src/circuits/standard_plonk.rs:29-48) as data"""
    return engine.ConstraintSystem.build(
        k, circuit.N_ADVICE, circuit.N_FIXED, circuit.N_INSTANCE, circuit.CS_DEGREE, circuit.BLINDING_FACTORS, engine.GATES_STANDARD_PLONK, [],
        [(engine.ADVICE, c) for c in circuit.PERMUTATION_COLUMNS], [], circuit.ADVICE_QUERIES, circuit.FIXED_QUERIES)


def _keygen(params: ParamsKZG, circuit, vk_only: bool) -> engine.Keys:
    syn = circuit.without_witnesses().synthesize()
    copies = [(lc, lr, rc, rr) for (lc, lr), (rc, rr) in syn.copies]  # columns are already indices into the permutation argument
    return engine.Keys(constraint_system(circuit, params.k), params, syn.fixed, copies, vk_only=vk_only)


def keygen_vk(params: ParamsKZG, circuit) -> VerifyingKey:
    keys = _keygen(params, circuit, vk_only=True)
    vk = VerifyingKey(params.k, circuit.CS_DEGREE, keys.fixed_commitments, keys.permutation_commitments)
    keys.release()
    return vk


class _Columns:
    """coefficient / extended-coset (and Lagrange) forms of a group of the key's columns: views of library-owned vectors"""

    def __init__(self, keys: engine.Keys, kinds, count: int):
        values, polys, cosets = kinds
        self.values = keys.views(values, count)
        self.polys = keys.views(polys, count)
        self.cosets = keys.views(cosets, count)


class ProvingKey:
    def __init__(self, vk: VerifyingKey, circuit, keys: engine.Keys):
        self.vk, self.circuit, self.keys = vk, circuit, keys
        self.fixed = _Columns(keys, (engine.PKBUF_FIXED, engine.PKBUF_FIXED_POLY, engine.PKBUF_FIXED_COSET), circuit.N_FIXED)
        self.permutation = _Columns(keys, (engine.PKBUF_SIGMA, engine.PKBUF_SIGMA_POLY, engine.PKBUF_SIGMA_COSET), len(circuit.PERMUTATION_COLUMNS))

    l0 = property(lambda self: self.keys.view(engine.PKBUF_L0_COSET))
    l_last = property(lambda self: self.keys.view(engine.PKBUF_L_LAST_COSET))
    l_active = property(lambda self: self.keys.view(engine.PKBUF_L_ACTIVE_COSET))

    def get_vk(self) -> VerifyingKey:
        return self.vk

    def release(self):
        self.keys.release()


def keygen_pk(params: ParamsKZG, vk: VerifyingKey, circuit) -> ProvingKey:
    keys = _keygen(params, circuit, vk_only=False)
    if not (np.array_equal(keys.fixed_commitments, vk.fixed_commitments) and np.array_equal(keys.permutation_commitments, vk.permutation_commitments)):
        keys.release()
        raise ValueError("keygen_pk: the verifying key belongs to another circuit or SRS")
    return ProvingKey(vk, circuit, keys)
