"""create_proof for the reference's StandardPlonk circuit — a caller of the library's prover.

Mirror of halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK, Challenge255, _, Blake2bWrite, _>
as the reference calls it (examples/standard_plonk.rs:41-49: one circuit, no instances).  This side does what the caller of a
Rust fork does: synthesize the witness (a handful of field operations), own the Blake2b transcript, hash vk.transcript_repr
into it; engine.Prover.drive then alternates seven phase calls into libh2mi.so (h2mi_prover.h) with the transcript's writes and
challenges.  Every vector lives in HBM inside the library; the only device -> host traffic is the 64-byte commitments and
32-byte evaluations the transcript absorbs.

rng: the reference passes OsRng (its proofs are not reproducible); here `seed` drives the library's counter-based SplitMix64
streams — the same streams oracle/prover.py draws, so proofs can be compared byte for byte.
"""
from . import engine
from . import field as F
from .keygen import ProvingKey, _m
from .params import ParamsKZG
from .transcript import Blake2bWrite

R = F.FR_MODULUS


class _Shplonk:
    def __init__(self, prover: engine.Prover):
        self._p = prover

    h_x = property(lambda self: self._p.view(engine.BUF_SHPLONK_H))
    h2_x = property(lambda self: self._p.view(engine.BUF_SHPLONK_H2))


class ProverWorkspace:
    """one library prover (device buffers, streams), reused from proof to proof (the reference's examples prove repeatedly
    against one pk: examples/linear_regression.rs:178-185).  The attributes are read-only views of the library's vectors, for
    callers that check them (the test-suite evaluates the quotient identity on them).
    combiner: a dist.PhaseCombiner with >= 8 slots when `params` is one rank's slice of the SRS (one process per GPU)."""

    def __init__(self, params: ParamsKZG, pk: ProvingKey, combiner=None):
        self.combiner = combiner
        self.prover = p = engine.Prover(pk.keys, params, combiner=combiner)
        na, nz = pk.circuit.N_ADVICE, len(pk.circuit.PERMUTATION_COLUMNS)  # chunk length cs_degree - 2 = 1: one product per column
        self.advice, self.advice_polys, self.advice_cosets = (p.views(kind, na) for kind in (engine.BUF_ADVICE, engine.BUF_ADVICE_POLY, engine.BUF_ADVICE_COSET))
        self.z, self.z_polys, self.z_cosets = (p.views(kind, nz) for kind in (engine.BUF_PERM_Z, engine.BUF_PERM_Z_POLY, engine.BUF_PERM_Z_COSET))
        self.shplonk = _Shplonk(p)

    random_poly = property(lambda self: self.prover.view(engine.BUF_RANDOM_POLY))
    h = property(lambda self: self.prover.view(engine.BUF_H))
    h_poly = property(lambda self: self.prover.view(engine.BUF_H_POLY))

    def release(self):
        self.prover.release()


def create_proof(params: ParamsKZG, pk: ProvingKey, circuit, seed: int, transcript: Blake2bWrite = None, ws: ProverWorkspace = None,
                 trace: dict = None) -> bytes:
    """-> proof bytes (transcript.finalize()).  `trace`, if given, receives the challenges and the workspace (whose views
    reach the intermediate polynomials left in HBM)."""
    own_ws = ws is None
    ws = ws or ProverWorkspace(params, pk)
    transcript = transcript or Blake2bWrite.init()
    transcript.common_scalar(_m(pk.vk.transcript_repr))  # vk.hash_into
    syn = circuit.synthesize()  # witness cells: the control plane
    try:
        ws.prover.drive(syn.advice, [], seed, transcript, trace)
    except BaseException:
        if own_ws:
            ws.release()
        raise
    if trace is not None:
        trace["ws"] = ws
    proof = transcript.finalize()
    if own_ws and trace is None:
        ws.release()
    return proof
