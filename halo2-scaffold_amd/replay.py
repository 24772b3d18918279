"""The MSM/NTT sequence one Halo2/KZG proof issues — "prover hot-path replay".

The shape of a proof (how many commitments and transforms of which size) is fixed by the circuit's
constraint system; `ProofShape` carries it and `ProofReplay` issues exactly that sequence, in
create_proof's order, on synthetic vectors resident in HBM.  This is NOT create_proof():
gate evaluation, transcript hashing and witness generation are out of scope.

Shapes:
  STANDARD_PLONK  reference src/circuits/standard_plonk.rs:29-48 (3 advice + 5 fixed columns, one
                  degree-3 gate, equality on a, b, c), proved by examples/standard_plonk.rs:41-49.
  HALO2_LIB_GATE  reference src/scaffold.rs:379-421 GateWithInstanceCircuitBuilder (halo2_lib.rs /
                  poseidon.rs closures): one FlexGate advice column at these sizes, one instance column,
                  degree-3 basic gate  [column counts restated from memory of halo2-base: SURVEY 3.3].
  RANGE_LOOKUP    reference src/scaffold.rs:434-485 RangeWithInstanceCircuitBuilder (range.rs): with a single advice
                  column the lookup's input is q_lookup * a (no lookup-advice column) => constraint degree 5: one
                  permutation set of three columns, four h pieces, extended domain 4n (see flex.FlexGateCS).

Multi-GPU: every rank owns one contiguous slice of every base set; each MSM runs on the slice and the
96-byte partial points are combined at EVERY transcript join (all-gather + device fold, dist.PhaseCombiner): a
prover cannot draw the next challenge before it holds the phase's full commitments.  Every
NTT runs on ONE GPU (NTT is single-GPU by design): a transform whose output feeds a later
commitment (h(X), the polynomials opened by SHPLONK) is replayed by every rank, which then reads its own
slice with no exchange; the independent "leaf" transforms (coefficient / extended forms that only feed
gate evaluation) are spread round-robin over the ranks as replicas — no collective.
"""
from dataclasses import dataclass

import numpy as np

from . import synth
from ._lib import check, lib
from .device import DevBuf
from .domain import EvaluationDomain
from .params import ParamsKZG


@dataclass(frozen=True)
class ProofShape:
    name: str
    n_advice: int
    n_instance: int
    n_perm_columns: int  # columns with equality enabled
    n_lookups: int
    cs_degree: int       # max(gate degree, 3, lookup: 2 + input degree + table degree)

    @property
    def n_perm_z(self) -> int:  # permutation columns are chunked by (degree - 2)
        chunk = self.cs_degree - 2
        return -(-self.n_perm_columns // chunk)

    @property
    def msm_per_proof(self) -> int:
        # advice + (A', S') per lookup + z per chunk + z per lookup + random poly + h pieces + 2 SHPLONK
        return self.n_advice + 2 * self.n_lookups + self.n_perm_z + self.n_lookups + 1 + (self.cs_degree - 1) + 2

    @property
    def ntt_per_proof(self) -> dict:
        polys = self.n_advice + self.n_perm_z + 3 * self.n_lookups
        return {"intt_n": self.n_instance + polys, "coset_ntt_ext": self.n_instance + polys, "coset_intt_ext": 1}


STANDARD_PLONK = ProofShape("standard_plonk", n_advice=3, n_instance=0, n_perm_columns=3, n_lookups=0, cs_degree=3)
HALO2_LIB_GATE = ProofShape("halo2_lib_gate", n_advice=1, n_instance=1, n_perm_columns=3, n_lookups=0, cs_degree=3)
RANGE_LOOKUP = ProofShape("range_lookup", n_advice=1, n_instance=1, n_perm_columns=3, n_lookups=1, cs_degree=5)
SHAPES = {s.name: s for s in (STANDARD_PLONK, HALO2_LIB_GATE, RANGE_LOOKUP)}

# StandardPlonk constants kept for callers / tests that name them
N_ADVICE, N_PERM_Z, CS_DEGREE = STANDARD_PLONK.n_advice, STANDARD_PLONK.n_perm_z, STANDARD_PLONK.cs_degree
MSM_PER_PROOF = STANDARD_PLONK.msm_per_proof
NTT_PER_PROOF = STANDARD_PLONK.ntt_per_proof


class ProofReplay:
    def __init__(self, shape: ProofShape, k: int, rank: int = 0, world: int = 1, srs_secret: int = 0x5EC2E7 + 0x48324D49, dist="uniform",
                 combine_backend=None, torch_device=None, spread_leaf_ntts: bool = True, with_evaluate_h: bool = False):
        """combine_backend: None (single GPU), "nccl" (RCCL all-gather from / to device memory on the library's
        stream) or "gloo" (host all-gather: CPU rehearsal of the N > 1 schedule)."""
        self.shape = shape
        # optional: compute h(X) from the extended forms with the device evaluate_h (StandardPlonk only; the
        # proving-key cosets - fixed, sigma, l_0, l_last, l_active - are synthetic dense vectors)
        self.with_evaluate_h = with_evaluate_h and shape.name == "standard_plonk" and world == 1
        self.spread = spread_leaf_ntts and world > 1
        self.k, self.n = k, 1 << k
        self.rank, self.world = rank, world
        self.domain = EvaluationDomain(shape.cs_degree, k)
        n = self.n
        assert n % world == 0
        self.lo, self.hi = rank * n // world, (rank + 1) * n // world
        self.n_local = self.hi - self.lo
        # SRS: every rank generates the full g / g_lagrange on its GPU, registers only its slice
        # (the slice's window tables are built straight from the device-resident SRS: no host round trip)
        full = ParamsKZG.setup(k, srs_secret, register=world == 1)
        if world == 1:
            self.params = full
        else:
            self.params = full.register_slice(self.lo, self.hi)
            full.release()
        gen = {"witness": synth.witness_like_fr, "circuit": synth.circuit_like_fr}.get(dist, synth.uniform_fr)
        sh = shape
        # Lagrange-basis vectors: advice, lookup (A', S', z) triples, permutation products, instance
        self.advice = [DevBuf.from_numpy(gen(n, synth.SEED + 10 + i)) for i in range(sh.n_advice)]
        self.lookup = [DevBuf.from_numpy(synth.uniform_fr(n, synth.SEED + 40 + i)) for i in range(3 * sh.n_lookups)]
        self.perm_z = [DevBuf.from_numpy(synth.uniform_fr(n, synth.SEED + 60 + i)) for i in range(sh.n_perm_z)]
        self.instance = [DevBuf.from_numpy(gen(n, synth.SEED + 80 + i)) for i in range(sh.n_instance)]
        self.cols = self.advice + self.perm_z  # kept for bench.py's stats probe
        self.random_poly = DevBuf.from_numpy(synth.uniform_fr(n, synth.SEED + 20))
        npoly = sh.n_advice + sh.n_perm_z + 3 * sh.n_lookups + sh.n_instance
        self.work = [DevBuf(n * 32) for _ in range(npoly)]
        ext = self.domain.extended_len()
        self.ext = [DevBuf(ext * 32) for _ in range(npoly)]
        self.h = DevBuf(ext * 32)
        self.combiner = None
        if combine_backend is not None:
            from .dist import PhaseCombiner

            self.combiner = PhaseCombiner(sh.msm_per_proof, combine_backend, torch_device)
            self.out = None
            self.out_ptr = self.combiner.partial_ptr
        else:
            self.out = DevBuf(96 * sh.msm_per_proof)
            self.out_ptr = self.out.ptr
        self.probe_out = DevBuf(96)  # result slot for stats probes outside a step
        for e in self.ext:
            check(lib.h2mi_memset_zero(e.ptr, ext * 32), "zero")
        self.h.upload(synth.uniform_fr(ext, synth.SEED + 30))
        self._h_src = DevBuf(ext * 32)
        self._h_src.copy_from(self.h)
        self.counts = {"msm": 0, "intt_n": 0, "coset_ntt_ext": 0, "coset_intt_ext": 0}
        if self.with_evaluate_h:
            self.pk_cosets = [DevBuf.from_numpy(synth.uniform_fr(ext, synth.SEED + 100 + i)) for i in range(5 + 3 + 3)]
        check(lib.h2mi_sync(), "sync")

    # ---- primitives ----
    def _msm(self, buf: DevBuf, lagrange: bool, offset_elems: int = 0):
        h = self.params.g_lagrange_handle if lagrange else self.params.g_handle
        src = buf.ptr + (offset_elems + self.lo) * 32
        check(lib.h2mi_msm_bn254_g1_dev(h, src, self.n_local, self.out_ptr + 96 * self._slot, None), "msm")
        self._slot += 1
        self.counts["msm"] += 1

    def _msm_many(self, bufs, lagrange: bool, offsets=None):
        """the commitments of one phase over one base set as ONE call (h2mi_msm_bn254_g1_phase_dev): up to 2^17 points per rank — an
        8-GPU rank's slice of a 2^20-row proof — their partition and accumulation kernels are launched once for the group"""
        import ctypes as C

        if not bufs:
            return
        h = self.params.g_lagrange_handle if lagrange else self.params.g_handle
        offsets = offsets or [0] * len(bufs)
        ptrs = (C.c_void_p * len(bufs))(*[b.ptr + (off + self.lo) * 32 for b, off in zip(bufs, offsets)])
        check(lib.h2mi_msm_bn254_g1_phase_dev(h, ptrs, len(bufs), self.n_local, self.out_ptr + 96 * self._slot, 0, None), "msm")
        self._slot += len(bufs)
        self.counts["msm"] += len(bufs)

    def _mine(self) -> bool:
        """round-robin owner of the next leaf transform"""
        self._leaf += 1
        return (not self.spread) or (self._leaf % self.world == self.rank)

    def _to_coeff_and_extended(self, lagr: DevBuf, w: DevBuf, e: DevBuf, coeff_needed: bool = False):
        """lagrange_to_coeff then coeff_to_extended, both out of place (lagr -> w -> e: no clones, no zero padding).
        `coeff_needed`: the coefficient form feeds a later commitment, so every rank computes it."""
        n = self.n
        mine = self._mine()
        if mine or coeff_needed:
            self.domain.lagrange_to_coeff_oop_dev(lagr, w)
            self.counts["intt_n"] += 1
        if mine:
            self.domain.coeff_to_extended_oop_dev(w, e)
            self.counts["coset_ntt_ext"] += 1

    def step(self, phase_joins: bool = True):
        """one proof's worth of hot-path work, queued on the library streams (asynchronous).

        The commitments follow create_proof's order.  Where the real prover must hash
        commitments into the transcript before it can continue (challenges theta, beta/gamma, y, x, u), the
        replay joins the MSM pipeline (`h2mi_join`), so no overlap is claimed that a prover could not have;
        a column's transforms are queued as soon as the column exists (see below)."""
        sh, d, n = self.shape, self.domain, self.n
        self._slot = 0
        self._phase_start = 0

        def join():
            """transcript join: the phase's commitments must exist in full — on every rank — before the next
            challenge: device-side join of the MSM pipeline, then (N > 1) all-gather + fold of the phase's partial points"""
            if not phase_joins:
                return
            check(lib.h2mi_join(), "join")
            if self.combiner is not None:
                self.combiner.combine(self._phase_start, self._slot - self._phase_start)
            self._phase_start = self._slot

        self._leaf = -1
        # buffers are handed out in create_proof's order of use (instance, permutation products, advice, lookups)
        w_it, e_it = iter(self.work), iter(self.ext)
        inst_we = [(c, next(w_it), next(e_it)) for c in self.instance]
        perm_we = [(c, next(w_it), next(e_it)) for c in self.perm_z]
        adv_we = [(c, next(w_it), next(e_it)) for c in self.advice]
        look_we = [(c, next(w_it), next(e_it)) for c in self.lookup]
        first_adv_w = adv_we[0][1]
        # The transforms of a column depend on that column only, not on the challenges, so they are queued as
        # soon as the column exists — before the join that follows its commitment — and run while the batched
        # bucket reductions of that phase (latency-bound, on the tail stream) finish.  create_proof itself runs
        # them later (inside evaluate_h's preparation); the data dependencies are the same.
        # phase 2: advice commitments (Lagrange basis) -> challenge theta; their coefficient / extended forms
        self._msm_many(list(self.advice), lagrange=True)
        for i, (c, w, e) in enumerate(adv_we):
            self._to_coeff_and_extended(c, w, e, coeff_needed=i == 0 or w is self.work[0])
        join()
        # phase 3: lookups: commit permuted input / table columns -> challenges beta, gamma
        for i in range(sh.n_lookups):
            self._msm(self.lookup[3 * i], lagrange=True)
            self._msm(self.lookup[3 * i + 1], lagrange=True)
        for j, (c, w, e) in enumerate(look_we):
            if j % 3 != 2:  # the permuted input / table columns; the product column exists only after beta, gamma
                self._to_coeff_and_extended(c, w, e)
        if sh.n_lookups:
            join()
        # phases 4-6: commitments of the permutation products, the lookup products and the random polynomial
        # (coefficient basis) -> challenge y.  All commitments of the phase are queued first and the columns'
        # transforms behind them: an MSM's partition then runs beside the accumulation of the one before it,
        # instead of waiting on the library stream behind transforms that are themselves slowed by that
        # accumulation (measured gain over the interleaved order: 0.1-0.3 ms per proof at k = 20, 4 % at k = 17).
        self._msm_many([c for c, w, e in perm_we] + [self.lookup[3 * i + 2] for i in range(sh.n_lookups)], lagrange=True)
        self._msm(self.random_poly, lagrange=False)
        for c, w, e in perm_we:
            self._to_coeff_and_extended(c, w, e, coeff_needed=w is self.work[0])
        for i in range(sh.n_lookups):
            self._to_coeff_and_extended(*look_we[3 * i + 2])
        for c, w, e in inst_we:
            self._to_coeff_and_extended(c, w, e, coeff_needed=w is self.work[0])
        join()
        # phase 8: h(X) back to coefficients, split into degree-1 pieces of n, commit each -> challenge x
        if self.with_evaluate_h:
            from . import plonk as gp

            # ext buffers in allocation order: 3 permutation products, then 3 advice columns
            zc, adv, pk = self.ext[0:3], self.ext[3:6], self.pk_cosets
            gp.evaluate_h(d, adv, pk[0:5], pk[5:8], zc, pk[8], pk[9], pk[10], 0xBE7A, 0x6A33A, 0x1234567, self.h)
        else:
            self.h.copy_from(self._h_src)
        d.extended_to_coeff_dev(self.h)
        self.counts["coset_intt_ext"] += 1
        self._msm_many([self.h] * (sh.cs_degree - 1), lagrange=False, offsets=[piece * n for piece in range(sh.cs_degree - 1)])
        join()
        # phase 10: SHPLONK h(X) commitment -> challenge u -> L(X)/(X-u) commitment
        self._msm(first_adv_w, lagrange=False)
        join()
        self._msm(self.work[0], lagrange=False)
        join()
        assert self._slot == sh.msm_per_proof

    def finish(self) -> np.ndarray:
        """wait and fetch the proof's commitments -> (msm_per_proof, 12); N > 1: the per-phase combined points."""
        if self.combiner is not None:
            if self._phase_start < self._slot:  # step(phase_joins=False): one combine for the whole proof
                check(lib.h2mi_join(), "join")
                self.combiner.combine(self._phase_start, self._slot - self._phase_start)
                self._phase_start = self._slot
            check(lib.h2mi_sync(), "sync")
            return self.combiner.result()
        check(lib.h2mi_sync(), "sync")
        return self.out.to_numpy(shape=(self.shape.msm_per_proof, 12))

    def release(self):
        self.params.release()
        for b in self.advice + self.lookup + self.perm_z + self.instance + self.work + self.ext + [self.h, self._h_src, self.probe_out, self.random_poly]:
            b.free()
        if self.out is not None:
            self.out.free()
        if self.combiner is not None:
            self.combiner.release()


class StandardPlonkReplay(ProofReplay):
    def __init__(self, k: int, **kw):
        super().__init__(STANDARD_PLONK, k, **kw)
