"""ctypes binding of include/h2mi_prover.h: the device-resident prover of libh2mi.so behind its phase-level C ABI.

`Keys` is keygen_vk + keygen_pk (reference examples/standard_plonk.rs:33-34, src/scaffold.rs:284,287); `Prover` owns one
h2mi_prover_t and `drive()` is the body of create_proof (examples/standard_plonk.rs:41-49, src/scaffold.rs:322-331) as a
caller sees it: seven phase calls with the Blake2b transcript in between.  The transcript, the witness cells and
vk.transcript_repr stay on this side; every pass over a vector is inside the library (csrc/h2mi_prover.cpp).  The circuit
modules (keygen.py / prover.py: StandardPlonk; flex.py: the halo2-lib builders) describe their constraint system as data and
call these two classes — there is no other prover orchestration in Python.
"""
import ctypes as C
import itertools
import weakref

import numpy as np

from . import field as F
from ._lib import H2miError, check, lib
from .transcript import Blake2bWrite

ADVICE, FIXED, INSTANCE = 0, 1, 2  # H2MI_COL_*
GATES_STANDARD_PLONK, GATES_FLEX_VERTICAL = 1, 2
CELLS_CANONICAL = 1
KEYGEN_VK_ONLY = 1
EUNSAT = -7
MAX_GATES, MAX_PERM, MAX_LOOKUPS, MAX_QUERIES = 32, 64, 8, 192  # H2MI_MAX_* (include/h2mi_prover.h)

# h2mi_prover_buffer kinds
(BUF_ADVICE, BUF_ADVICE_POLY, BUF_ADVICE_COSET, BUF_INSTANCE, BUF_PERM_Z, BUF_PERM_Z_POLY, BUF_PERM_Z_COSET, BUF_LOOKUP_PERMUTED_INPUT,
 BUF_LOOKUP_PERMUTED_TABLE, BUF_LOOKUP_Z, BUF_RANDOM_POLY, BUF_H, BUF_H_POLY, BUF_SHPLONK_H, BUF_SHPLONK_H2) = range(15)
(PKBUF_FIXED, PKBUF_FIXED_POLY, PKBUF_FIXED_COSET, PKBUF_SIGMA, PKBUF_SIGMA_POLY, PKBUF_SIGMA_COSET, PKBUF_L0_COSET, PKBUF_L_LAST_COSET,
 PKBUF_L_ACTIVE_COSET) = range(64, 73)


class Column(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32)]


class Query(C.Structure):
    _fields_ = [("column", C.c_uint32), ("rotation", C.c_int32)]


class Lookup(C.Structure):
    _fields_ = [("input", Column), ("selector_fixed", C.c_int32), ("table_fixed", C.c_uint32)]


class ConstraintSystem(C.Structure):
    """h2mi_constraint_system: ConstraintSystem<Fr> after configure(), as the numbers create_proof reads off it"""
    _fields_ = [("k", C.c_uint32), ("n_advice", C.c_uint32), ("n_fixed", C.c_uint32), ("n_instance", C.c_uint32), ("degree", C.c_uint32),
                ("blinding_factors", C.c_uint32), ("gates", C.c_uint32), ("n_gates", C.c_uint32), ("gate_advice", C.c_uint32 * MAX_GATES),
                ("gate_selector", C.c_uint32 * MAX_GATES), ("n_perm", C.c_uint32), ("perm_columns", Column * MAX_PERM), ("n_lookups", C.c_uint32),
                ("lookups", Lookup * MAX_LOOKUPS), ("n_advice_queries", C.c_uint32), ("n_fixed_queries", C.c_uint32),
                ("advice_queries", Query * MAX_QUERIES), ("fixed_queries", Query * MAX_QUERIES)]

    @classmethod
    def build(cls, k, n_advice, n_fixed, n_instance, degree, blinding_factors, gates, gate_columns, perm_columns, lookups, advice_queries,
              fixed_queries) -> "ConstraintSystem":
        """gate_columns: [(advice column, selector fixed column)]; perm_columns: [(kind, index)]; lookups: [(input advice column,
        selector fixed column or None, table fixed column)]; queries: [(column, rotation)] in creation order"""
        cs = cls()
        cs.k, cs.n_advice, cs.n_fixed, cs.n_instance, cs.degree, cs.blinding_factors, cs.gates = k, n_advice, n_fixed, n_instance, degree, blinding_factors, gates
        assert len(gate_columns) <= MAX_GATES and len(perm_columns) <= MAX_PERM and len(lookups) <= MAX_LOOKUPS
        assert len(advice_queries) <= MAX_QUERIES and len(fixed_queries) <= MAX_QUERIES
        cs.n_gates = len(gate_columns)
        for g, (a, q) in enumerate(gate_columns):
            cs.gate_advice[g], cs.gate_selector[g] = a, q
        cs.n_perm = len(perm_columns)
        for j, (kind, index) in enumerate(perm_columns):
            cs.perm_columns[j] = Column(kind, index)
        cs.n_lookups = len(lookups)
        for l, (inp, sel, table) in enumerate(lookups):
            cs.lookups[l] = Lookup(Column(ADVICE, inp), -1 if sel is None else sel, table)
        cs.n_advice_queries, cs.n_fixed_queries = len(advice_queries), len(fixed_queries)
        for i, (c, r) in enumerate(advice_queries):
            cs.advice_queries[i] = Query(c, r)
        for i, (c, r) in enumerate(fixed_queries):
            cs.fixed_queries[i] = Query(c, r)
        return cs


class ColumnCells(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("values", C.c_void_p), ("count", C.c_size_t), ("flags", C.c_uint32)]


def pack_cells(columns):
    """[{row: value} or [value at row 0, 1, ..]] (integers below r) -> (h2mi_column_cells array, objects to keep alive).  The values
    cross as canonical 32-byte integers (H2MI_CELLS_CANONICAL): one C-level conversion per value and one join here, the Montgomery
    conversion where they land (long runs: on the device)."""
    arr = (ColumnCells * max(len(columns), 1))()
    keep = []
    for i, cells in enumerate(columns):
        if isinstance(cells, dict):
            rows = sorted(cells)
            dense = bool(rows) and rows[0] == 0 and rows[-1] == len(rows) - 1
            vals = map(cells.__getitem__, rows)
        else:
            rows, dense, vals = range(len(cells)), True, cells
        count = len(rows)
        if not count:
            continue
        try:
            raw = b"".join(map(int.to_bytes, vals, itertools.repeat(32), itertools.repeat("little")))
        except OverflowError:
            raise ValueError("a cell value is not reduced modulo r")
        v = np.frombuffer(raw, dtype=np.uint8)
        keep.append(v)
        arr[i].values, arr[i].count, arr[i].flags = v.ctypes.data, count, CELLS_CANONICAL
        if not dense:
            r = np.array(rows, dtype=np.uint32)
            keep.append(r)
            arr[i].rows = r.ctypes.data
    return arr, keep


class DevView:
    """a device vector owned by the library (h2mi_prover_buffer / h2mi_prover_pk_buffer): the DevBuf read interface, no ownership"""

    def __init__(self, ptr: int, count: int):
        self.ptr, self.nbytes = ptr, count * 32

    def to_numpy(self, dtype=np.uint64, shape=None, nbytes=None, offset=0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(lib.h2mi_memcpy_d2h(out.ctypes.data, self.ptr + offset, nbytes), "d2h")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        pass


class _Views:
    """list-like access to one kind of library-owned buffers"""

    def __init__(self, getter, handle, kind: int, count: int):
        self._get, self._h, self._kind, self._count = getter, handle, kind, count

    def __len__(self):
        return self._count

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._count))]
        if i < 0:
            i += self._count
        if not 0 <= i < self._count:
            raise IndexError(i)
        p, n = C.c_void_p(), C.c_size_t()
        check(self._get(self._h, self._kind, i, C.byref(p), C.byref(n)), "buffer")
        return DevView(p.value, n.value)

    def __iter__(self):
        return (self[i] for i in range(self._count))


class Keys:
    """h2mi_prover_keygen: keygen_vk + keygen_pk for a constraint system given as data.  fixed: one {row: value} dict (or dense
    list) per fixed column as synthesize() assigns them; copies: [(left column, left row, right column, right row)] per
    constrain_equal in call order, columns as indices into the permutation argument.  `params` is the WHOLE SRS."""

    def __init__(self, cs: ConstraintSystem, params, fixed, copies, vk_only: bool = False):
        self.cs = cs
        cells, keep = pack_cells(fixed)
        cp = np.ascontiguousarray(np.array(copies, dtype=np.uint32).reshape(-1, 4))
        h = C.c_void_p()
        check(lib.h2mi_prover_keygen(C.byref(cs), params.g_lagrange_handle, cells, cp.ctypes.data, len(cp), KEYGEN_VK_ONLY if vk_only else 0, C.byref(h)),
              "keygen")
        del keep
        self.handle = h.value
        self._provers = weakref.WeakSet()  # the library refuses to release a key while a prover created against it is alive
        self.fixed_commitments = np.zeros((cs.n_fixed, 8), dtype=np.uint64)
        self.permutation_commitments = np.zeros((cs.n_perm, 8), dtype=np.uint64)
        check(lib.h2mi_prover_vk_commitments(self.handle, self.fixed_commitments.ctypes.data, self.permutation_commitments.ctypes.data), "vk commitments")

    def views(self, kind: int, count: int) -> _Views:
        return _Views(lib.h2mi_prover_pk_buffer, self.handle, kind, count)

    def view(self, kind: int) -> DevView:
        return self.views(kind, 1)[0]

    def release(self):
        if self.handle:
            for p in list(self._provers):
                p.release()
            check(lib.h2mi_prover_pk_release(self.handle), "pk_release")
            self.handle = None


class _Counts(C.Structure):
    _fields_ = [("advice", C.c_uint32), ("lookups", C.c_uint32), ("products", C.c_uint32), ("quotient", C.c_uint32), ("evaluations", C.c_uint32)]


_COMBINE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)


class Prover:
    """one h2mi_prover_t: the device buffers, streams and phase state of one create_proof at a time, reused from proof to proof
    (the reference's drivers prove repeatedly against one pk / SRS, e.g. examples/linear_regression.rs:126-195).
    params: the whole SRS, or one rank's slice of it (ParamsKZG.register_slice) together with `combiner`, a dist.PhaseCombiner
    with a slot per commitment of the largest phase (8 covers the reference's shapes): every commitment is then this rank's partial point, combined across ranks whenever a phase reads its
    points back (the library calls back into combiner.combine)."""

    def __init__(self, keys: Keys, params, combiner=None):
        self.keys, self.combiner = keys, combiner
        h = C.c_void_p()
        check(lib.h2mi_prover_create(keys.handle, params.g_handle, params.g_lagrange_handle, params.lo, params.n, C.byref(h)), "prover_create")
        self.handle = h.value
        keys._provers.add(self)
        self.counts = _Counts()
        check(lib.h2mi_prover_get_counts(self.handle, C.byref(self.counts)), "prover counts")
        self._points = np.zeros((max(self.counts.advice, self.counts.lookups, self.counts.products, self.counts.quotient, 8), 8), dtype=np.uint64)
        self._evals = np.zeros((self.counts.evaluations, 4), dtype=np.uint64)
        self._cb = None
        self._cb_error = None
        if combiner is not None:
            c = self.counts
            assert combiner.slots >= max(c.advice, c.lookups, c.products, c.quotient), "the combiner needs a slot per commitment of the largest phase"

            def _combine(_ctx, count):
                try:
                    combiner.combine(0, count)
                    return 0
                except BaseException as e:  # an exception must not unwind through the C frames
                    self._cb_error = e
                    return 1

            self._cb = _COMBINE_FN(_combine)
            check(lib.h2mi_prover_set_combiner(self.handle, combiner.partial_ptr, combiner.combined.ptr, self._cb, None), "set_combiner")

    def set_rng_key(self, key: bytes = None) -> None:
        """h2mi_prover_set_rng_key: blinding from ChaCha20 under a 256-bit key (the `seed` of later proofs is then the per-proof
        nonce, below 2^61) instead of the reproducible 32-bit seeded streams; None returns to those"""
        if key is not None and len(key) != 32:
            raise ValueError("the key is 32 bytes")
        check(lib.h2mi_prover_set_rng_key(self.handle, key), "set_rng_key")

    def views(self, kind: int, count: int) -> _Views:
        return _Views(lib.h2mi_prover_buffer, self.handle, kind, count)

    def view(self, kind: int) -> DevView:
        return self.views(kind, 1)[0]

    def _phase(self, rc: int, what: str):
        if rc and self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        if rc == EUNSAT:
            raise ValueError("lookup input not in the table (ConstraintSystemFailure)")
        check(rc, what)

    def drive(self, advice, instance, seed: int, transcript: Blake2bWrite, trace: dict = None) -> None:
        """create_proof between the transcript's challenges.  advice: one {row: value} dict or dense list per advice column;
        instance: the public inputs (integers).  The caller has hashed vk.transcript_repr and the public inputs already."""
        c, h, pts = self.counts, self.handle, self._points
        pp = pts.ctypes.data
        sq = transcript.squeeze_challenge  # 4 Montgomery limbs

        def write_points(k):
            for i in range(k):
                transcript.write_point(pts[i])  # raises on the identity, as the crate's transcript does

        import time

        marks = [("start", time.perf_counter())]
        mark = (lambda name: marks.append((name, time.perf_counter()))) if trace is not None else (lambda name: None)
        cells, keep = pack_cells(advice)
        inst = np.ascontiguousarray(np.stack([F.fr_to_mont_limbs(v) for v in instance])) if len(instance) else np.zeros((1, 4), dtype=np.uint64)
        self._phase(lib.h2mi_prover_advice(h, cells, inst.ctypes.data, len(instance), seed, pp), "advice")
        del keep
        write_points(c.advice)
        mark("advice committed")
        theta = sq()  # drawn even without lookups
        if c.lookups:
            self._phase(lib.h2mi_prover_lookups(h, theta.ctypes.data, pp), "lookups")
            write_points(c.lookups)
            mark("permuted lookup columns committed")
        beta, gamma = sq(), sq()
        self._phase(lib.h2mi_prover_products(h, beta.ctypes.data, gamma.ctypes.data, pp), "products")
        write_points(c.products)
        mark("z, random committed")
        y = sq()
        self._phase(lib.h2mi_prover_quotient(h, y.ctypes.data, pp), "quotient")
        write_points(c.quotient)
        mark("h pieces committed")
        x = sq()
        self._phase(lib.h2mi_prover_evaluations(h, x.ctypes.data, self._evals.ctypes.data), "evaluations")
        for e in self._evals:
            transcript.write_scalar(e)
        mark("evaluations written")
        sy, sv = sq(), sq()  # ProverSHPLONK: y, v
        self._phase(lib.h2mi_prover_shplonk_quotient(h, sy.ctypes.data, sv.ctypes.data, pp), "shplonk quotient")
        write_points(1)
        su = sq()
        self._phase(lib.h2mi_prover_shplonk_open(h, su.ctypes.data, pp), "shplonk open")
        write_points(1)
        mark("shplonk done")
        if trace is not None:
            m = F.fr_from_mont_limbs
            trace.update(theta=m(theta), beta=m(beta), gamma=m(gamma), y=m(y), x=m(x),
                         phase_ms=[(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(marks, marks[1:])])

    def release(self):
        if self.handle:
            check(lib.h2mi_prover_destroy(self.handle), "prover_destroy")
            self.handle = None


__all__ = ["ConstraintSystem", "Keys", "Prover", "DevView", "H2miError"]
