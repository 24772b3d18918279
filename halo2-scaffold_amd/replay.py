"""The MSM/NTT sequence one StandardPlonk proof issues — "prover hot-path replay".

Shape from the reference circuit (src/circuits/standard_plonk.rs:29-48: 3 advice + 5 fixed columns, one
degree-3 gate, equality on a, b, c; examples/standard_plonk.rs:41-49 calls create_proof on it) and the
create_proof call stack of SURVEY.md 3.3: per proof 11 MSM(n) + 6 iNTT(n) + 6 coset-NTT(2n) +
1 coset-iNTT(2n).  This is NOT create_proof(): gate evaluation, transcript hashing and witness
generation are out of scope; vectors are synthetic and stay resident in HBM.

Multi-GPU: every rank replays the NTTs on its own GPU (NTT is single-GPU by design) and owns one
contiguous slice of every base set; each MSM runs on the slice and the 96-byte partial points are
combined by `combine` (all-gather + fold, see dist.py).
"""
import numpy as np

from . import synth
from ._lib import check, lib
from .device import DevBuf
from .domain import EvaluationDomain
from .params import ParamsKZG

N_ADVICE = 3       # reference src/circuits/standard_plonk.rs:13-15
N_PERM_Z = 3       # equality enabled on a, b, c (standard_plonk.rs:34), chunk length 1
CS_DEGREE = 3      # q_ab * a * b (standard_plonk.rs:47)
MSM_PER_PROOF = N_ADVICE + N_PERM_Z + 1 + (CS_DEGREE - 1) + 2
NTT_PER_PROOF = {"intt_n": N_ADVICE + N_PERM_Z, "coset_ntt_ext": N_ADVICE + N_PERM_Z, "coset_intt_ext": 1}


class StandardPlonkReplay:
    def __init__(self, k: int, rank: int = 0, world: int = 1, srs_secret: int = 0x5EC2E7 + 0x48324D49, dist="uniform", combine=None):
        self.k, self.n = k, 1 << k
        self.rank, self.world = rank, world
        self.combine = combine
        self.domain = EvaluationDomain(CS_DEGREE, k)
        n = self.n
        assert n % world == 0
        self.lo, self.hi = rank * n // world, (rank + 1) * n // world
        self.n_local = self.hi - self.lo
        # SRS: every rank generates the full g / g_lagrange on its GPU, registers only its slice
        full = ParamsKZG.setup(k, srs_secret)
        if world == 1:
            self.params = full
        else:
            g = full.get_g()[self.lo : self.hi]
            gl = full.get_g_lagrange()[self.lo : self.hi]
            full.release()
            self.params = ParamsKZG(k)
            self.params.n = self.n_local
            self.params._g_dev = DevBuf.from_numpy(g)
            self.params._gl_dev = DevBuf.from_numpy(gl)
            self.params._register()
        gen = synth.witness_like_fr if dist == "witness" else synth.uniform_fr
        # Lagrange-basis columns: 3 advice, 3 permutation products, 1 random (vanishing) polynomial
        self.cols = [DevBuf.from_numpy(gen(n, synth.SEED + 10 + i)) for i in range(N_ADVICE + N_PERM_Z)]
        self.random_poly = DevBuf.from_numpy(synth.uniform_fr(n, synth.SEED + 20))
        self.work = [DevBuf(n * 32) for _ in range(N_ADVICE + N_PERM_Z)]
        ext = self.domain.extended_len()
        self.ext = [DevBuf(ext * 32) for _ in range(N_ADVICE + N_PERM_Z)]
        self.h = DevBuf(ext * 32)
        self.out = DevBuf(96 * MSM_PER_PROOF)
        self._zeros = np.zeros(((ext - n), 4), dtype=np.uint64)
        for e in self.ext:
            e.upload(self._zeros, offset=n * 32)
        self.h.upload(synth.uniform_fr(ext, synth.SEED + 30))
        self._h_src = DevBuf(ext * 32)
        self._h_src.copy_from(self.h)
        check(lib.h2mi_sync(), "sync")

    def _msm(self, slot: int, buf: DevBuf, lagrange: bool, offset_elems: int = 0):
        h = self.params.g_lagrange_handle if lagrange else self.params.g_handle
        src = buf.ptr + (offset_elems + self.lo) * 32
        check(lib.h2mi_msm_bn254_g1_dev(h, src, self.n_local, self.out.ptr + 96 * slot, None), "msm")

    def step(self, phase_joins: bool = True):
        """one proof's worth of hot-path work, queued on the library streams (asynchronous).

        The order follows create_proof.  Where the real prover must hash commitments
        into the transcript before it can continue (challenges theta/beta/gamma, y, x, u), the replay
        joins the MSM tails (`h2mi_join`), so no overlap is claimed that a prover could not have."""
        d, n = self.domain, self.n
        join = (lambda: check(lib.h2mi_join(), "join")) if phase_joins else (lambda: None)
        slot = 0
        # phase 2: advice commitments (Lagrange basis) -> challenges theta, beta, gamma
        for c in self.cols[:N_ADVICE]:
            self._msm(slot, c, lagrange=True)
            slot += 1
        join()
        # phase 4: permutation products: commit, lagrange_to_coeff, coeff_to_extended per column
        for c, w, e in zip(self.cols[N_ADVICE:], self.work[N_ADVICE:], self.ext[N_ADVICE:]):
            self._msm(slot, c, lagrange=True)
            slot += 1
            w.copy_from(c, n * 32)
            d.lagrange_to_coeff_dev(w)
            e.copy_from(w, n * 32)
            d.coeff_to_extended_dev(e)
        # phase 6: random polynomial commitment (coefficient basis) -> challenge y
        self._msm(slot, self.random_poly, lagrange=False)
        slot += 1
        join()
        # phase 7: advice lagrange_to_coeff + coeff_to_extended (evaluate_h inputs)
        for c, w, e in zip(self.cols[:N_ADVICE], self.work[:N_ADVICE], self.ext[:N_ADVICE]):
            w.copy_from(c, n * 32)
            d.lagrange_to_coeff_dev(w)
            e.copy_from(w, n * 32)
            d.coeff_to_extended_dev(e)
        # phase 8: h(X) back to coefficients, split into degree-1 pieces of n, commit each -> challenge x
        self.h.copy_from(self._h_src)
        d.extended_to_coeff_dev(self.h)
        for piece in range(CS_DEGREE - 1):
            self._msm(slot, self.h, lagrange=False, offset_elems=piece * n)
            slot += 1
        join()
        # phase 10: SHPLONK h(X) commitment -> challenge u -> L(X)/(X-u) commitment
        self._msm(slot, self.work[0], lagrange=False)
        slot += 1
        join()
        self._msm(slot, self.work[1], lagrange=False)
        slot += 1
        join()
        assert slot == MSM_PER_PROOF
        # restore the zero padding of the extended buffers for the next proof
        for e in self.ext:
            check(lib.h2mi_memset_zero(e.ptr + n * 32, (self.domain.extended_len() - n) * 32), "zero")

    def finish(self) -> np.ndarray:
        """wait, fetch the MSM_PER_PROOF partial results, combine across ranks -> (MSM_PER_PROOF, 12)."""
        check(lib.h2mi_sync(), "sync")
        part = self.out.to_numpy(shape=(MSM_PER_PROOF, 12))
        if self.world > 1 and self.combine is not None:
            return self.combine(part)
        return part
