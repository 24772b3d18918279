"""create_proof for the reference's StandardPlonk circuit with every vector resident in HBM.

Mirror of halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK, Challenge255, _, Blake2bWrite, _>
as the reference calls it (examples/standard_plonk.rs:41-49: one circuit, no instances) — plonk/prover.rs restated
from memory of v2023_02_02, in its order:

  vk hashed into the transcript; advice columns (witness cells + blinding rows) committed in the Lagrange basis;
  theta; beta, gamma; the three permutation grand products (device scans), committed; the vanishing argument's random
  polynomial, committed; y; coefficient and extended-coset forms (iNTT, coset NTT); evaluate_h + division by X^n - 1
  (one element-wise kernel over the pk's cosets); coset iNTT; the two h pieces committed; x; every query evaluated
  (device Horner) and written; ProverSHPLONK (shplonk.py).

The host does what the crate's single-threaded control flow does — witness cells, Blake2b, challenge arithmetic on
single field elements, launch order; every pass over a length-n (or 2n) vector is a HIP kernel behind the C ABI, and
the only device -> host traffic is the 64-byte commitments and 32-byte evaluations the transcript absorbs.

rng: the reference passes OsRng (its proofs are not reproducible); here `seed` drives counter-based SplitMix64 streams
(seed+1 advice blinding rows, seed+2 permutation-product blinding rows, seed+3 the random polynomial — generated on
the device by h2mi_fr_random_dev), the same streams oracle/prover.py draws, so proofs can be compared byte for byte.
"""
import ctypes as C

import numpy as np

from . import field as F
from . import plonk as gp
from . import synth
from ._lib import check, lib
from .device import DevBuf, SideStream
from .keygen import ProvingKey, _m, _patch
from .params import ParamsKZG
from .shplonk import ProverSHPLONK
from .transcript import Blake2bWrite

R = F.FR_MODULUS


class ProverWorkspace:
    """device buffers of one prover, reused from proof to proof (the reference's examples prove repeatedly against
    one pk: examples/linear_regression.rs:178-185)"""

    def __init__(self, params: ParamsKZG, pk: ProvingKey, combiner=None):
        """combiner: a dist.PhaseCombiner with >= 4 slots when `params` is one rank's slice of the SRS (one process per
        GPU): every commitment is then this rank's partial point, combined across ranks at the phase's join"""
        self.combiner = combiner
        d = pk.vk.domain
        n, ext = d.n, d.extended_len()
        na = pk.circuit.N_ADVICE
        nz = len(pk.circuit.PERMUTATION_COLUMNS)  # chunk length cs_degree - 2 = 1: one product per column
        self.advice = [DevBuf(n * 32) for _ in range(na)]
        self.advice_polys = [DevBuf(n * 32) for _ in range(na)]
        self.advice_cosets = [DevBuf(ext * 32) for _ in range(na)]
        self.z = [DevBuf(n * 32) for _ in range(nz)]
        self.z_polys = [DevBuf(n * 32) for _ in range(nz)]
        self.z_cosets = [DevBuf(ext * 32) for _ in range(nz)]
        self.random_poly = DevBuf(n * 32)
        self.h = DevBuf(ext * 32)
        self.h_poly = DevBuf(n * 32)
        self.points = DevBuf(96 * 4)     # Jacobian results of the commitments of one phase
        self.evals = DevBuf(32 * 32)
        self.shplonk = ProverSHPLONK(n)
        self.side = SideStream()         # transforms of the advice columns, beside the permutation argument's chain

    def release(self):
        for b in (self.advice + self.advice_polys + self.advice_cosets + self.z + self.z_polys + self.z_cosets +
                  [self.random_poly, self.h, self.h_poly, self.points, self.evals]):
            b.free()
        self.shplonk.release()
        self.side.free()


_Q = F.FQ_MODULUS
_RINV_Q = pow(1 << 256, -1, _Q)


def _write_phase_points(ws: ProverWorkspace, transcript, k: int):
    """fetch the k Jacobian results of a phase (the copy joins the MSM pipeline), normalise them on the host as
    G1::batch_normalize does (one modular inversion each is microseconds here; a lone device thread takes 0.3 ms)
    and write them to the transcript"""
    if ws.combiner is not None:  # sliced SRS: all-gather + fold of the phase's partial points, then the same on every rank
        check(lib.h2mi_join(), "join")
        ws.combiner.combine(0, k)
        jac = ws.combiner.combined.to_numpy(shape=(ws.combiner.slots, 12))[:k]
    else:
        jac = ws.points.to_numpy(shape=(4, 12), nbytes=96 * 4)[:k]
    for row in jac:
        X, Y, Z = (sum(int(row[4 * c + i]) << (64 * i) for i in range(4)) * _RINV_Q % _Q for c in range(3))
        if Z == 0:
            raise ValueError("cannot write points at infinity to the transcript")
        zi = pow(Z, -1, _Q)
        zi2 = zi * zi % _Q
        transcript.write_point_xy(X * zi2 % _Q, Y * zi2 % _Q * zi % _Q)


def _commit_phase(params: ParamsKZG, ws: ProverWorkspace, transcript, columns, lagrange: bool):
    """commit the columns of one phase (MSMs queued back to back, bucket reductions batched by the join) and write the
    points to the transcript"""
    _commit_columns(params, ws, columns, lagrange, inorder=True)
    _write_phase_points(ws, transcript, len(columns))


def _patch_cells(addrs, values: np.ndarray):
    """values[i] (Montgomery limbs) -> device address addrs[i], one launch per 64 cells on the library stream"""
    ptrs = (C.c_void_p * len(addrs))(*addrs)
    check(lib.h2mi_fr_patch_cells_dev(ptrs, values.ctypes.data, len(addrs), None), "patch_cells")


def _commit_columns(params: ParamsKZG, ws: ProverWorkspace, columns, lagrange: bool, sparse: bool = False, inorder: bool = False):
    """queue the commitments of one phase into result slots 0 .. len - 1 with ONE call (h2mi_msm_bn254_g1_batch_dev): below 2^17 rows
    their partition and accumulation kernels are launched once for the whole phase"""
    h = params.g_lagrange_handle if lagrange else params.g_handle
    out = ws.combiner.partial_ptr if ws.combiner is not None else ws.points.ptr
    ptrs = (C.c_void_p * len(columns))(*[buf.ptr + (offset_elems + params.lo) * 32 for buf, offset_elems in columns])
    # flags: 1 = the sparse promise (batched launches at every size), 2 = in order (the group is all its phase commits and is read back
    # next: its bucket reductions follow its accumulation on one stream) — H2MI_MSM_SPARSE / H2MI_MSM_INORDER of h2mi.h
    check(lib.h2mi_msm_bn254_g1_phase_dev(h, ptrs, len(columns), params.n, out, (1 if sparse else 0) | (2 if inorder else 0), None), "commit")


def _commit(params: ParamsKZG, ws: ProverWorkspace, buf: DevBuf, offset_elems: int, lagrange: bool, slot: int):
    """queue one commitment into result slot `slot` of the phase: the whole column, or this rank's slice of it"""
    h = params.g_lagrange_handle if lagrange else params.g_handle
    out = (ws.combiner.partial_ptr if ws.combiner is not None else ws.points.ptr) + 96 * slot
    check(lib.h2mi_msm_bn254_g1_dev(h, buf.ptr + (offset_elems + params.lo) * 32, params.n, out, None), "commit")


def create_proof(params: ParamsKZG, pk: ProvingKey, circuit, seed: int, transcript: Blake2bWrite = None, ws: ProverWorkspace = None,
                 trace: dict = None) -> bytes:
    """-> proof bytes (transcript.finalize()).  `trace`, if given, receives the challenges and device buffers of the
    intermediate polynomials (tests evaluate the quotient identity on them)."""
    own_ws = ws is None
    ws = ws or ProverWorkspace(params, pk)
    transcript = transcript or Blake2bWrite.init()
    cs = pk.circuit
    d = pk.vk.domain
    n, ext = d.n, d.extended_len()
    bf = cs.BLINDING_FACTORS
    u = n - (bf + 1)  # unusable_rows_start; also the l_last row
    sq = lambda: F.fr_from_mont_limbs(transcript.squeeze_challenge())
    import time as _time

    marks = [("start", _time.perf_counter())]
    mark = (lambda name: marks.append((name, _time.perf_counter()))) if trace is not None else (lambda name: None)

    transcript.common_scalar(_m(pk.vk.transcript_repr))  # vk.hash_into

    # ---- advice: witness cells (host, a handful) + blinding rows, committed in the Lagrange basis ----------------
    syn = circuit.synthesize()
    blind = synth.uniform_fr(cs.N_ADVICE * (bf + 1), seed + 1)
    # assigned cells and blinding rows of every column in ONE launch (h2mi_fr_patch_cells_dev: the cells travel in the kernel's
    # arguments) instead of two small copies per column on the library stream in front of the phase's commitments
    addrs, vals = [], []
    for j, col in enumerate(ws.advice):
        check(lib.h2mi_memset_zero(col.ptr, n * 32), "zero")
        for r in sorted(syn.advice[j]):
            addrs.append(col.ptr + r * 32)
            vals.append(_m(syn.advice[j][r]))
        for r in range(bf + 1):
            addrs.append(col.ptr + (u + r) * 32)
            vals.append(blind[j * (bf + 1) + r])
    _patch_cells(addrs, np.ascontiguousarray(np.stack(vals)))
    _commit_columns(params, ws, [(c, 0) for c in ws.advice], True, sparse=True, inorder=True)  # a handful of assigned rows
    check(lib.h2mi_msm_flush(), "flush")  # the bucket reductions start now, not when the host reaches the join below
    # The coefficient / extended forms of the advice columns depend on no challenge (create_proof computes them after
    # y): on a side stream they run beside the commitments' bucket reductions, the transcript round trip and the
    # permutation argument's chain of small scans — a stretch in which the device is otherwise nearly idle, because the
    # advice columns of this circuit are almost empty — instead of queueing behind the z commitments (2 ms of 16 at k = 20)
    ws.side.after_library()
    for col, p, e in zip(ws.advice, ws.advice_polys, ws.advice_cosets):
        d.lagrange_to_coeff_oop_dev(col, p, stream=ws.side.handle)
        d.coeff_to_extended_oop_dev(p, e, stream=ws.side.handle)
    _write_phase_points(ws, transcript, len(ws.advice))
    mark("advice committed")
    theta = sq()  # drawn even without lookups
    beta, gamma = sq(), sq()
    # ---- vanishing argument: random polynomial (n coefficients from the prover's rng).  Its commitment is written after
    # the z commitments but depends on nothing: queued here, the one dense MSM of this phase accumulates beside the
    # permutation argument's latency-bound scans instead of after them
    check(lib.h2mi_fr_random_dev(ws.random_poly.ptr, n, seed + 3, 0, None), "random_poly")
    _commit(params, ws, ws.random_poly, 0, False, len(ws.z))

    # ---- permutation argument: one grand product per column (chunk length cs.degree() - 2 = 1), one device pass ---
    zblind = synth.uniform_fr(len(ws.z) * bf, seed + 2)
    gp.permutation_products(d.k, [ws.advice[c] for c in cs.PERMUTATION_COLUMNS], pk.permutation.values, cs.CS_DEGREE - 2, beta, gamma, u, ws.z,
                            active=pk.active_rows)
    _patch_cells([z.ptr + (u + 1 + r) * 32 for z in ws.z for r in range(bf)], np.ascontiguousarray(zblind[: len(ws.z) * bf]))
    _commit_columns(params, ws, [(z, 0) for z in ws.z], True, sparse=True)  # constant but for the copy constraints
    check(lib.h2mi_msm_flush(), "flush")
    # the coefficient / extended forms depend on the columns only (create_proof computes them after y): queued behind
    # the commitments, they run beside the MSMs' accumulation instead of delaying the grand products
    for col, p, e in zip(ws.z, ws.z_polys, ws.z_cosets):
        d.lagrange_to_coeff_oop_dev(col, p)
        d.coeff_to_extended_oop_dev(p, e)
    ws.side.join_library()  # evaluate_h and the openings read the advice forms
    mark("queued z/random commits")
    _write_phase_points(ws, transcript, len(ws.z) + 1)
    mark("z, random committed")
    y = sq()

    # ---- quotient: evaluate_h on the extended coset, divide by X^n - 1, back to coefficients, commit the pieces ----
    gp.evaluate_h(d, ws.advice_cosets, pk.fixed.cosets, pk.permutation.cosets, ws.z_cosets, pk.l0, pk.l_last, pk.l_active, beta, gamma, y, ws.h)
    d.extended_to_coeff_dev(ws.h)
    pieces = d.quotient_poly_degree
    _commit_phase(params, ws, transcript, [(ws.h, i * n) for i in range(pieces)], lagrange=False)
    mark("h pieces committed")
    x = sq()
    xn = pow(x, n, R)

    # ---- evaluations: every (column, rotation) query at x, in the order create_proof writes them -------------------
    rot = lambda r: x * pow(d.omega, r % n, R) % R
    x_next, x_last = rot(1), rot(-(bf + 1))
    # h(X) = sum_i xn^i h_i(X): the polynomial vanishing.open() queries
    ptrs = (C.c_void_p * pieces)(*[ws.h.ptr + i * n * 32 for i in range(pieces)])
    sc = np.ascontiguousarray(np.stack([_m(pow(xn, i, R)) for i in range(pieces)]))
    check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, pieces, n, ws.h_poly.ptr, None), "h_poly")
    written = []  # (poly, point) whose evaluation goes to the transcript, in order
    for c, r in cs.ADVICE_QUERIES:
        written.append((ws.advice_polys[c], rot(r)))
    for c, r in cs.FIXED_QUERIES:
        written.append((pk.fixed.polys[c], rot(r)))
    written.append((ws.random_poly, x))
    for sp in pk.permutation.polys:
        written.append((sp, x))
    for i, zp in enumerate(ws.z_polys):
        written += [(zp, x), (zp, x_next)]
        if i + 1 < len(ws.z_polys):
            written.append((zp, x_last))
    extra = [(ws.h_poly, x)]  # opened but not written (the verifier recomputes it)
    # every evaluation in ONE call, grouped by distinct point (x: 17 polynomials, omega x: 3, omega^last x: 2)
    todo = written + extra
    slot = {}
    points = list(dict.fromkeys(p for _, p in todo))
    ordered, counts = [], []
    for pt in points:
        group = [poly for poly, p in todo if p == pt]
        counts.append(len(group))
        for g in group:
            slot[(id(g), pt)] = len(slot)
            ordered.append(g)
    ptrs = (C.c_void_p * len(ordered))(*[g.ptr for g in ordered])
    cnt_arr = (C.c_size_t * len(counts))(*counts)
    pts_l = np.ascontiguousarray(np.stack([_m(pt) for pt in points]))  # named: the arrays must outlive the call that reads them
    check(lib.h2mi_fr_eval_polys_multi_dev(ptrs, cnt_arr, pts_l.ctypes.data, len(points), n, ws.evals.ptr, None), "eval")
    ev = ws.evals.to_numpy(shape=(32, 4))
    value = {key: F.fr_from_mont_limbs(ev[i]) for key, i in slot.items()}
    for poly, pt in written:
        transcript.write_scalar_int(value[(id(poly), pt)])

    mark("evaluations written")
    # ---- queries in create_proof's order, then SHPLONK ---------------------------------------------------------------
    queries = []
    q = lambda poly, pt: queries.append((poly, pt, value[(id(poly), pt)]))
    for c, r in cs.ADVICE_QUERIES:
        q(ws.advice_polys[c], rot(r))
    for zp in ws.z_polys:  # permutation.open: every set at x and omega x ...
        q(zp, x)
        q(zp, x_next)
    for zp in reversed(ws.z_polys[:-1]):  # ... then all but the last at omega^last x, in reverse
        q(zp, x_last)
    for c, r in cs.FIXED_QUERIES:
        q(pk.fixed.polys[c], rot(r))
    for sp in pk.permutation.polys:
        q(sp, x)
    q(ws.h_poly, x)
    q(ws.random_poly, x)

    def commit_and_write(poly: DevBuf):
        # a lone commitment, read back at once: in order on one stream, nothing deferred (h2mi_msm_bn254_g1_inorder_dev)
        out = ws.combiner.partial_ptr if ws.combiner is not None else ws.points.ptr
        check(lib.h2mi_msm_bn254_g1_inorder_dev(params.g_handle, poly.ptr + params.lo * 32, params.n, out, None), "commit")
        _write_phase_points(ws, transcript, 1)

    ws.shplonk.create_proof(transcript, queries, commit_and_write)
    mark("shplonk done")
    if trace is not None:
        trace.update(theta=theta, beta=beta, gamma=gamma, y=y, x=x, ws=ws,
                     phase_ms=[(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(marks, marks[1:])])
    proof = transcript.finalize()
    if own_ws and trace is None:
        ws.release()
    return proof
