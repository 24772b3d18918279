"""keygen + create_proof for the halo2-lib constraint systems the reference proves through `scaffold::prove`
(src/scaffold.rs:246-366), with every vector resident in HBM — BASELINE configs[2] (halo2_lib.rs: x^2 + 72) and
configs[3] (range.rs: range_check(x, 64) with a LOOKUP_BITS table).

The reference builds these circuits with halo2-base (GateWithInstanceCircuitBuilder / RangeWithInstanceCircuitBuilder,
src/scaffold.rs:379-485: FlexGate's vertical gate q (a + a(wX) a(w^2 X) - a(w^3 X)) on the advice column, a constants
column, RangeConfig's table and — these circuits fit one advice column, so — its complex selector q_lookup on the
looked-up cells' own rows (`range.q_lookup`, :464-469) instead of a lookup-advice column, and one instance column added —
and equality-enabled — last, :394-395, :449-450).  halo2-base is an un-vendored dependency; its cell layout (load_witness, mul, add, mul_add,
range_check's limb decomposition) is restated from memory in `halo2_lib_closure` / `range_closure` below — the witness
generation a prover runs on the CPU.  Everything after it is the same device pipeline as prover.py, generalised:
an instance column (public inputs are hashed into the transcript and take part in the permutation argument), advice
queries at rotations 0..3, permutation sets of `degree - 2` columns, and for the Range builder one lookup argument with
the input expression q_lookup * a (row values by one element-wise product, permuted columns by counting sort against the
keygen-sorted table, lookup grand product, five more terms in evaluate_h; degree 5: extended domain 4n, four h pieces).  Checked against oracle/flex.py: byte-identical proofs at
small k, accepted by its verifier at k = 16 and above (tests/test_gpu_flex.py).
rng stand-in as in prover.py, plus streams seed+4 (blinding rows of the permuted lookup columns) and seed+5 (of the
lookup product).
"""
import ctypes as C
import itertools
import hashlib
import struct

import numpy as np

from . import field as F
from . import plonk as gp
from . import serde, synth
from ._lib import check, lib
from .circuits import PermutationAssembly
from .device import DevBuf, SideStream
from .domain import EvaluationDomain
from .keygen import FR_DELTA, _m, commit_points
from .params import ParamsKZG
from .prover import _Q, _RINV_Q
from .shplonk import ProverSHPLONK
from .transcript import Blake2bWrite

R = F.FR_MODULUS
ADVICE, FIXED, INSTANCE = "advice", "fixed", "instance"


class FlexGateCS:
    """the constraint system of the Gate builder (lookup = False) or the Range builder (lookup = True) when the circuit
    fits ONE advice column (every example of the reference at its DEGREE).  Columns in configure()'s allocation order
    [RECALL halo2-base]: RangeConfig takes the lookup table column first; FlexGateConfig the constants column
    (enable_equality at once: it leads the permutation argument), then the gate advice column and its simple selector;
    the scaffold adds the instance column last.  With a single advice column RangeConfig looks up q_lookup * a (complex
    selector) instead of adding a lookup-advice column.  keygen appends the selector columns: complex ones first.
      Gate:  fixed 0 constants, 1 q_enable                          degree 3: permutation sets of one, 2 h pieces
      Range: fixed 0 table, 1 constants, 2 q_lookup, 3 q_enable     degree 2 + 2 + 1 = 5: sets of three, 4 h pieces
    Queries in creation order (enable_equality queries its column at Rotation::cur)."""

    def __init__(self, lookup: bool, num_advice: int = 1, num_lookup_advice: int = 0, k: int = None, minimum_rows: int = 9):
        self.lookup = lookup
        self.num_advice, self.num_lookup_advice = num_advice, num_lookup_advice
        self.k, self.minimum_rows = k, minimum_rows  # the multi-column layout needs the row budget 2^k - minimum_rows
        if num_advice > 1:
            self._init_multi()
            return
        assert num_lookup_advice == 0
        self.n_advice = 1
        if lookup:
            self.col_table, self.col_const, self.col_qlookup, self.col_q = 0, 1, 2, 3
            self.n_fixed = 4
            self.fixed_queries = [(1, 0), (0, 0), (2, 0), (3, 0)]
        else:
            self.col_table = self.col_qlookup = None
            self.col_const, self.col_q = 0, 1
            self.n_fixed = 2
            self.fixed_queries = [(0, 0), (1, 0)]
        self.col_qs = [self.col_q]
        self.perm_columns = [(FIXED, self.col_const), (ADVICE, 0), (INSTANCE, 0)]
        self.advice_queries = [(0, 0), (0, 1), (0, 2), (0, 3)]
        self.degree = 5 if lookup else 3          # the lookup of a degree-2 input is what raises it
        self.blinding_factors = 6                 # max(3, four queries on the gate column) + 2
        self.chunk = self.degree - 2

    def _init_multi(self):
        """more than one gate column (round 4): what `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268) configures when
        the cells overflow 2^k - minimum_rows rows [RECALL halo2-base 0.3]: FlexGateConfig allocates the constants column, then
        per gate column an advice column (equality-enabled) with its own simple selector and vertical gate; RangeConfig takes the
        table column first and — there being more than one gate column — no q_lookup but num_lookup_advice lookup-advice columns
        (advice, equality-enabled, after the gate columns), one lookup argument each with the column itself as input: degree
        2 + 1 + 1 = 4, permutation sets of two, three h pieces; keygen appends one fixed column per selector (gates of different
        columns share rows: compress_selectors cannot merge them); the scaffold adds the instance column last."""
        A, Lc = self.num_advice, self.num_lookup_advice
        assert 2 <= A <= 4 and Lc <= 2, "the device quotient kernel takes up to four gate columns and two lookup-advice columns"
        assert (Lc >= 1) == bool(self.lookup), "the Range builder needs a lookup-advice column, the Gate builder has none"
        assert self.k is not None, "the multi-column layout needs k (rows per column = 2^k - minimum_rows)"
        self.n_advice = A + Lc
        self.col_table, self.col_const = (0, 1) if self.lookup else (None, 0)
        self.col_qlookup = None
        self.col_qs = [self.col_const + 1 + j for j in range(A)]
        self.col_q = None
        self.n_fixed = self.col_const + 1 + A
        self.fixed_queries = [(self.col_const, 0)] + ([(self.col_table, 0)] if self.lookup else []) + [(c, 0) for c in self.col_qs]
        self.perm_columns = [(FIXED, self.col_const)] + [(ADVICE, j) for j in range(A + Lc)] + [(INSTANCE, 0)]
        self.advice_queries = [(j, r) for j in range(A) for r in range(4)] + [(A + l, 0) for l in range(Lc)]
        self.degree = 4 if self.lookup else 3
        self.blinding_factors = 6
        self.chunk = self.degree - 2


def configure(lookup: bool, k: int, closure, minimum_rows: int = 9) -> FlexGateCS:
    """GateThreadBuilder::config (src/scaffold.rs:268 `builder.config(k, Some(minimum_rows))`): run the closure once to count its
    cells and cells to look up, and take ceil(count / (2^k - minimum_rows)) columns of each kind.  `closure(cs) -> Assignment`."""
    probe = FlexGateCS(lookup)
    asg = closure(probe)
    max_rows = (1 << k) - minimum_rows
    cells = len(asg.advice[0])
    num_advice = max(1, -(-cells // max_rows))
    if num_advice == 1:
        return FlexGateCS(lookup, k=k, minimum_rows=minimum_rows)
    looked_up = len(asg.fixed[probe.col_qlookup]) if lookup else 0
    return FlexGateCS(lookup, num_advice, max(1, -(-looked_up // max_rows)) if lookup else 0, k=k, minimum_rows=minimum_rows)


class Assignment:
    def __init__(self, cs: FlexGateCS):
        self.cs = cs
        self.advice = [dict() for _ in range(cs.n_advice)]
        self.fixed = [dict() for _ in range(cs.n_fixed)]
        self.instance = []   # public inputs (column 0)
        self.copies = []     # ((kind, column, row), (kind, column, row)) in constrain_equal order


class Context:
    """halo2-base `Context` on one advice column [layout restated from memory]: cells are appended in program order;
    Existing(cell) re-assigns the value and constrains it equal to the original; Constant(v) cells are tied to one fixed
    cell per distinct value afterwards; `cells_to_lookup` get q_lookup enabled on their own rows (single-column form)."""

    def __init__(self, asg: Assignment):
        self.asg = asg
        self.cells, self.const_cells, self.lookup_cells = [], [], []
        self.gates, self.eqs = [], []  # rows (flat cell indices) with the gate enabled; (new cell, source cell) equalities in call order

    def load_witness(self, v: int) -> int:
        self.cells.append(v % R)
        return len(self.cells) - 1

    def assign_region_last(self, items, gate_offsets) -> int:
        base = len(self.cells)
        for kind, v in items:
            row = len(self.cells)
            if kind == "existing":
                self.cells.append(self.cells[v])
                self.eqs.append((row, v))
            else:
                self.cells.append(v % R)
                if kind == "constant":
                    self.const_cells.append((row, v % R))
        for off in gate_offsets:
            self.gates.append(base + off)
        return len(self.cells) - 1

    # GateInstructions
    def mul(self, a, b):
        return self.assign_region_last([("constant", 0), ("existing", a), ("existing", b), ("witness", self.cells[a] * self.cells[b])], [0])

    def add(self, a, b):
        return self.assign_region_last([("existing", a), ("existing", b), ("constant", 1), ("witness", self.cells[a] + self.cells[b])], [0])

    def add_constant(self, a, c):
        return self.assign_region_last([("existing", a), ("constant", c), ("constant", 1), ("witness", self.cells[a] + c)], [0])

    def mul_add_constant(self, a, b, c):
        return self.assign_region_last([("constant", c), ("existing", a), ("existing", b), ("witness", self.cells[a] * self.cells[b] + c)], [0])

    # RangeInstructions::range_check(a, range_bits)
    def range_check(self, a, range_bits: int, lookup_bits: int):
        x = self.cells[a]
        assert x < 1 << range_bits, "witness out of range"
        num_limbs = -(-range_bits // lookup_bits)
        limbs = [(x >> (lookup_bits * i)) & ((1 << lookup_bits) - 1) for i in range(num_limbs)]
        rows = [self.load_witness(limbs[0])]  # inner_product_left_last with bases[0] = 1: the first limb is the first accumulator
        acc, acc_row = limbs[0], rows[0]
        for i in range(1, num_limbs):  # [acc, limb_i, 2^(b i), acc'] sharing the accumulator cell: a gate every third row
            base = len(self.cells) - 1
            acc += limbs[i] << (lookup_bits * i)
            self.cells.append(limbs[i])
            rows.append(len(self.cells) - 1)
            self.cells.append((1 << (lookup_bits * i)) % R)
            self.const_cells.append((len(self.cells) - 1, (1 << (lookup_bits * i)) % R))
            self.cells.append(acc % R)
            self.gates.append(base)
            acc_row = len(self.cells) - 1
        self.eqs.append((a, acc_row))  # ctx.constrain_equal(&a, &acc)
        self.lookup_cells += rows
        rem = range_bits % lookup_bits
        if rem == 1:  # the top limb is one bit: assert_bit, 0 + x * x - x = 0 on [0, x, x, x]
            self.assign_region_last([("constant", 0), ("existing", rows[-1]), ("existing", rows[-1]), ("existing", rows[-1])], [0])
        elif rem:  # the top limb times 2^(lookup_bits - rem) must be in the table too
            self.lookup_cells.append(self.assign_region_last([("constant", 0), ("existing", rows[-1]), ("constant", 1 << (lookup_bits - rem)),
                                                              ("witness", limbs[-1] << (lookup_bits - rem))], [0]))

    def finish(self, public_rows):
        asg = self.asg
        cs = asg.cs
        if cs.num_advice > 1:
            return self._finish_multi(public_rows)
        asg.advice[0] = dict(enumerate(self.cells))
        for g in self.gates:
            asg.fixed[cs.col_q][g] = 1
        asg.copies += [((ADVICE, 0, new), (ADVICE, 0, src)) for new, src in self.eqs]
        consts = {}
        for row, v in self.const_cells:  # assign_constants: one fixed cell per distinct value, in order of first use
            if v not in consts:
                consts[v] = len(consts)
                asg.fixed[cs.col_const][consts[v]] = v
            asg.copies.append(((ADVICE, 0, row), (FIXED, cs.col_const, consts[v])))
        for row in self.lookup_cells:
            asg.fixed[cs.col_qlookup][row] = 1
        for i, row in enumerate(public_rows):  # layouter.constrain_instance(cell, instance, i): src/scaffold.rs:411, 480
            asg.instance.append(self.cells[row])
            asg.copies.append(((ADVICE, 0, row), (INSTANCE, 0, i)))

    def _finish_multi(self, public_rows):
        """assign_all over several gate columns [RECALL halo2-base 0.3 gates/builder.rs]: the cells run down the current column; a
        cell that lands on the column's last row (row >= max_rows - 1), or that starts a gate which no longer fits (row + 4 >
        max_rows), is assigned a second time at row 0 of the next column and tied to its first copy — two gates may overlap at it —
        and a gate starting there is enabled on the new column.  The cells to look up are copied into the lookup-advice columns.
        constrain_equal order: break copies as they occur, lookup copies, the closure's equalities, constants, public cells."""
        asg = self.asg
        cs = asg.cs
        A, Lc = cs.num_advice, cs.num_lookup_advice
        max_rows = (1 << cs.k) - cs.minimum_rows
        gate_at = set(self.gates)
        where = []  # flat cell index -> (column, row) of its first copy
        col, row = 0, 0
        for i, v in enumerate(self.cells):
            asg.advice[col][row] = v
            where.append((col, row))
            q = i in gate_at
            if (q and row + 4 > max_rows) or row >= max_rows - 1:
                if col + 1 >= A:
                    raise ValueError(f"NOT ENOUGH ADVICE COLUMNS: more than {A} gate columns needed at 2^{cs.k} rows")
                asg.copies.append(((ADVICE, col + 1, 0), (ADVICE, col, row)))
                col, row = col + 1, 0
                asg.advice[col][0] = v
            if q:
                asg.fixed[cs.col_qs[col]][row] = 1
            row += 1
        cell = lambda i: (ADVICE,) + where[i]
        lcol, lrow = 0, 0
        for i in self.lookup_cells:
            if lrow >= max_rows:
                lcol, lrow = lcol + 1, 0
            if lcol >= Lc:
                raise ValueError("NOT ENOUGH LOOKUP ADVICE COLUMNS")
            asg.advice[A + lcol][lrow] = self.cells[i]
            asg.copies.append((cell(i), (ADVICE, A + lcol, lrow)))
            lrow += 1
        asg.copies += [(cell(new), cell(src)) for new, src in self.eqs]
        consts = {}
        for r, v in self.const_cells:
            if v not in consts:
                consts[v] = len(consts)
                asg.fixed[cs.col_const][consts[v]] = v
            asg.copies.append((cell(r), (FIXED, cs.col_const, consts[v])))
        for i, r in enumerate(public_rows):
            asg.instance.append(self.cells[r])
            asg.copies.append((cell(r), (INSTANCE, 0, i)))


def halo2_lib_closure(cs: FlexGateCS, x: int) -> Assignment:
    """reference examples/halo2_lib.rs:14-60 `some_algorithm_in_zk`: x^2 + 72 three ways; make_public = [x, out]"""
    asg = Assignment(cs)
    ctx = Context(asg)
    xc = ctx.load_witness(x)
    x_sq = ctx.mul(xc, xc)
    out = ctx.add_constant(x_sq, 72)
    ctx.assign_region_last([("constant", 72), ("existing", xc), ("existing", xc), ("witness", x * x + 72)], [0])
    ctx.mul_add_constant(xc, xc, 72)
    ctx.finish([xc, out])
    return asg


def range_closure(cs: FlexGateCS, x: int, lookup_bits: int) -> Assignment:
    """reference examples/range.rs:10-34: make_public = [x]; range_check(x, 64); x + x.  The table column is what
    RangeConfig::load_lookup_table assigns: 0 .. 2^LOOKUP_BITS - 1 (src/scaffold.rs:462)."""
    asg = Assignment(cs)
    ctx = Context(asg)
    xc = ctx.load_witness(x)
    ctx.range_check(xc, 64, lookup_bits)
    ctx.add(xc, xc)
    ctx.finish([xc])
    asg.fixed[cs.col_table] = None  # dense: filled by keygen from `table_values`
    asg.table_values = list(range(1 << lookup_bits))
    return asg


def mock(asg: Assignment) -> None:
    """scaffold::mock (src/scaffold.rs:205-243: MockProver::run(..).assert_satisfied()) for these constraint systems, on the
    host: every enabled row satisfies the vertical gate, every copy constraint joins equal cells, every looked-up cell is
    a table value.  Raises ValueError naming the first violation — what the reference's users run before `prove`."""
    cs = asg.cs
    for j, cq in enumerate(cs.col_qs):  # gate column j with its own selector
        a = asg.advice[j]
        for r in sorted(asg.fixed[cq]):
            if (a.get(r, 0) + a.get(r + 1, 0) * a.get(r + 2, 0) - a.get(r + 3, 0)) % R:
                raise ValueError(f"gate not satisfied at row {r}" + (f" of column {j}" if len(cs.col_qs) > 1 else ""))
    value = {ADVICE: lambda c, r: asg.advice[c].get(r, 0), FIXED: lambda c, r: asg.fixed[c].get(r, 0),
             INSTANCE: lambda c, r: asg.instance[r] if r < len(asg.instance) else 0}
    for left, right in asg.copies:
        if value[left[0]](left[1], left[2]) % R != value[right[0]](right[1], right[2]) % R:
            raise ValueError(f"copy constraint {left} == {right} not satisfied")
    if cs.lookup:
        table = set(v % R for v in asg.table_values) | {0}
        if cs.num_advice == 1:
            looked_up = [(r, asg.advice[0].get(r, 0)) for r in sorted(asg.fixed[cs.col_qlookup])]
        else:
            looked_up = [(r, v) for l in range(cs.num_lookup_advice) for r, v in sorted(asg.advice[cs.num_advice + l].items())]
        for r, v in looked_up:
            if v % R not in table:
                raise ValueError(f"lookup not satisfied at row {r}")


# ---- keys ------------------------------------------------------------------------------------------------------------
def _column_from_cells(n: int, cells, into: DevBuf = None) -> DevBuf:
    """a column with the given {row: value} cells, zero elsewhere.  A long run of cells (a witness of thousands of
    cells, a lookup table) is uploaded as canonical 32-byte integers and brought to Montgomery form on the device
    (h2mi_fe_from_repr_dev, in place); the host only packs bytes."""
    d = into if into is not None else DevBuf(n * 32)
    check(lib.h2mi_memset_zero(d.ptr, n * 32), "zero")
    if cells:
        lo, hi = min(cells), max(cells) + 1
        if hi - lo <= 4 * len(cells) + 16:  # contiguous enough: one upload
            if len(cells) > 64:
                # one C-level conversion per value and one join (round 4: a generator with `% R` and `.to_bytes` per cell cost
                # 1.1 - 1.4 ms for poseidon's 7.4 k cells, all of it before the first commitment could be queued); the values
                # are reduced where they are assigned (Context), a stray unreduced one fails the conversion below (bad != 0)
                if hi - lo == len(cells):
                    vals = map(cells.__getitem__, range(lo, hi))
                else:
                    get = cells.get
                    vals = (get(r, 0) for r in range(lo, hi))
                raw = b"".join(map(int.to_bytes, vals, itertools.repeat(32), itertools.repeat("little")))
                d.upload(np.frombuffer(raw, dtype=np.uint8), offset=lo * 32)
                bad = C.c_uint64()
                check(lib.h2mi_fe_from_repr_dev(1, d.ptr + lo * 32, hi - lo, d.ptr + lo * 32, C.byref(bad)), "from_repr")
                if bad.value:
                    raise ValueError("a cell value is not reduced modulo r")
            else:
                arr = np.zeros((hi - lo, 4), dtype=np.uint64)
                for r, v in cells.items():
                    arr[r - lo] = _m(v)
                d.upload(arr, offset=lo * 32)
        else:
            for r in sorted(cells):
                d.upload(_m(cells[r]), offset=r * 32)
    return d


class FlexKeys:
    """keygen_vk + keygen_pk (src/scaffold.rs:284,287): fixed columns, sigma polynomials from the copy constraints
    (Assembly::copy over advice, constants and instance cells alike), l_0 / l_last / l_active, the lookup table's sorted
    form, and the verifying key's commitments."""

    def __init__(self, params: ParamsKZG, cs: FlexGateCS, asg: Assignment):
        self.cs = cs
        k = params.k
        self.domain = d = EvaluationDomain(cs.degree, k)
        n = d.n
        self.u = u = n - (cs.blinding_factors + 1)
        fixed_cells = list(asg.fixed)
        self.table = None
        if cs.lookup:
            tv = asg.table_values
            if len(tv) > u:
                raise ValueError(f"lookup table of {len(tv)} rows does not fit the {u} usable rows of a 2^{k} circuit (LOOKUP_BITS must be below DEGREE)")
            fixed_cells[cs.col_table] = dict(enumerate(tv))
            self.table = gp.LookupTable(tv + [0] * (u - len(tv)), u)
        self.fixed_values = [_column_from_cells(n, cells) for cells in fixed_cells]
        m = len(cs.perm_columns)
        index = {col: j for j, col in enumerate(cs.perm_columns)}
        asm = PermutationAssembly()
        for left, right in asg.copies:
            asm.copy((index[(left[0], left[1])], left[2]), (index[(right[0], right[1])], right[2]))
        omega_pows = DevBuf(n * 32)
        check(lib.h2mi_fr_powers_dev(omega_pows.ptr, n, d._omega.ctypes.data, None), "powers")
        self.sigma_values = []
        for j in range(m):
            col = DevBuf(n * 32)
            ptrs = (C.c_void_p * 1)(omega_pows.ptr)
            sc = _m(pow(FR_DELTA, j, R))
            check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, 1, n, col.ptr, None), "identity permutation")
            self.sigma_values.append(col)
        dpow = [pow(FR_DELTA, j, R) for j in range(m)]
        for (j, i), (tj, ti) in asm.mapping.items():
            if (j, i) != (tj, ti):
                self.sigma_values[j].patch(_m(dpow[tj] * pow(d.omega, ti, R) % R), offset=i * 32)  # stream-ordered, no host wait per cell
        check(lib.h2mi_sync(), "sync")
        omega_pows.free()
        self.active_rows = gp.ActiveRows(asm.mapping, cs.chunk, u)  # the support of the copy constraints (sparse grand products)
        self.fixed_commitments = commit_points(params, self.fixed_values, lagrange=True)
        self.permutation_commitments = commit_points(params, self.sigma_values, lagrange=True)
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        s = self.vk_bytes()
        h.update(struct.pack("<Q", len(s)))
        h.update(s)
        self.transcript_repr = int.from_bytes(h.digest(), "little") % R
        ext = d.extended_len()

        def forms(col):
            p, e = DevBuf(n * 32), DevBuf(ext * 32)
            d.lagrange_to_coeff_oop_dev(col, p)
            d.coeff_to_extended_oop_dev(p, e)
            return p, e

        self.fixed_polys, self.fixed_cosets = zip(*[forms(c) for c in self.fixed_values])
        self.sigma_polys, self.sigma_cosets = zip(*[forms(c) for c in self.sigma_values])
        lag = [DevBuf(n * 32) for _ in range(3)]
        for b in lag[:2]:
            check(lib.h2mi_memset_zero(b.ptr, n * 32), "zero")
        lag[0].upload(_m(1))
        lag[1].upload(_m(1), offset=u * 32)
        one = _m(1)
        check(lib.h2mi_fr_fill_dev(lag[2].ptr, n, one.ctypes.data, None), "fill")
        check(lib.h2mi_memset_zero(lag[2].ptr + u * 32, (n - u) * 32), "zero")
        lforms = [forms(b) for b in lag]
        check(lib.h2mi_sync(), "sync")
        self.l0, self.l_last, self.l_active = (e for _, e in lforms)
        for p, _ in lforms:
            p.free()
        for b in lag:
            b.free()

    def vk_bytes(self) -> bytes:
        pts = np.concatenate([self.fixed_commitments, self.permutation_commitments])
        return struct.pack("<II", self.domain.k, self.cs.degree) + serde.g1_to_bytes(pts).tobytes()

    def release(self):
        for b in (list(self.fixed_values) + list(self.sigma_values) + list(self.fixed_polys) + list(self.fixed_cosets) + list(self.sigma_polys)
                  + list(self.sigma_cosets) + [self.l0, self.l_last, self.l_active]):
            b.free()
        if self.table is not None:
            self.table.free()
        self.active_rows.free()


# ---- create_proof --------------------------------------------------------------------------------------------------------
class FlexWorkspace:
    """device buffers of one create_proof, kept for the next one against the same proving key (the reference's
    drivers prove repeatedly against one pk / SRS, e.g. examples/linear_regression.rs:126-195): create_proof takes
    its buffers in a fixed order with fixed sizes, so the pool hands the i-th request the i-th buffer.
    combiner: a dist.PhaseCombiner with >= 8 slots when `params` is one rank's slice of the SRS (one process per GPU):
    every commitment is then this rank's partial point, combined across ranks at every transcript write."""

    def __init__(self, params: ParamsKZG, pk: "FlexKeys", combiner=None):
        self.combiner = combiner
        self.points = DevBuf(96 * 8)
        self._pool, self._cursor = [], 0
        self.shplonk = ProverSHPLONK(pk.domain.n)
        self.side = SideStream()  # transforms that wait for no challenge run here, beside the library stream's chain

    def begin(self):
        self._cursor = 0

    def take(self, count: int) -> DevBuf:
        if self._cursor == len(self._pool):
            self._pool.append(DevBuf(count * 32))
        b = self._pool[self._cursor]
        assert b.nbytes == count * 32, "workspace reused with another proving key"
        self._cursor += 1
        return b

    def release(self):
        for b in self._pool + [self.points]:
            b.free()
        self.shplonk.release()
        self.side.free()


def _write_points(ws: FlexWorkspace, transcript, k: int):
    """fetch the k Jacobian results of a phase (the copy joins the MSM pipeline; with a sliced SRS: all-gather + fold of
    the partial points first), normalise on the host, write to the transcript"""
    if ws.combiner is not None:
        check(lib.h2mi_join(), "join")
        ws.combiner.combine(0, k)
        jac = ws.combiner.combined.to_numpy(shape=(ws.combiner.slots, 12))[:k]
    else:
        jac = ws.points.to_numpy(shape=(8, 12), nbytes=96 * 8)[:k]
    for row in jac:
        X, Y, Z = (sum(int(row[4 * c + i]) << (64 * i) for i in range(4)) * _RINV_Q % _Q for c in range(3))
        if Z == 0:
            raise ValueError("cannot write points at infinity to the transcript")
        zi = pow(Z, -1, _Q)
        zi2 = zi * zi % _Q
        transcript.write_point_xy(X * zi2 % _Q, Y * zi2 % _Q * zi % _Q)


def create_proof(params: ParamsKZG, pk: FlexKeys, asg: Assignment, seed: int, transcript: Blake2bWrite = None, trace: dict = None,
                 ws: FlexWorkspace = None) -> bytes:
    """create_proof for one circuit with one instance column: scaffold::prove's call (src/scaffold.rs:322-331,
    `&[&[&public_io]]`).  `params` is the whole SRS, or one rank's slice of it together with ws.combiner.  Without `ws` the
    device buffers live for this call only (released on every exit path)."""
    if ws is not None:
        return _create_proof(params, pk, asg, seed, transcript, trace, ws)
    own = FlexWorkspace(params, pk)
    try:
        return _create_proof(params, pk, asg, seed, transcript, trace, own)
    finally:
        check(lib.h2mi_sync(), "sync")
        own.release()


def _create_proof(params: ParamsKZG, pk: FlexKeys, asg: Assignment, seed: int, transcript, trace, ws: FlexWorkspace) -> bytes:
    cs, d = pk.cs, pk.domain
    n, ext, u, bf = d.n, d.extended_len(), pk.u, cs.blinding_factors
    transcript = transcript or Blake2bWrite.init()
    sq = lambda: F.fr_from_mont_limbs(transcript.squeeze_challenge())
    ws.begin()
    dev = ws.take
    out_base = ws.combiner.partial_ptr if ws.combiner is not None else ws.points.ptr

    def commit(buf, lagrange, slot, offset_elems=0):
        h = params.g_lagrange_handle if lagrange else params.g_handle
        check(lib.h2mi_msm_bn254_g1_dev(h, buf.ptr + (offset_elems + params.lo) * 32, params.n, out_base + 96 * slot, None), "commit")

    def commit_many(bufs, lagrange, slot0, sparse, inorder=False, offsets=None):
        """the commitments of one phase into slots slot0 .. with ONE call (batched launches below 2^17 rows; at every size for columns the
        caller knows to be sparse: witness columns of a padded circuit, grand products)"""
        h = params.g_lagrange_handle if lagrange else params.g_handle
        offsets = offsets or [0] * len(bufs)
        ptrs = (C.c_void_p * len(bufs))(*[b.ptr + (off + params.lo) * 32 for b, off in zip(bufs, offsets)])
        # flags: 1 = sparse promise, 2 = in order (the group is all its phase commits and is read back next): h2mi.h H2MI_MSM_*
        check(lib.h2mi_msm_bn254_g1_phase_dev(h, ptrs, len(bufs), params.n, out_base + 96 * slot0, (1 if sparse else 0) | (2 if inorder else 0), None), "commit")

    def forms(col, stream=None):
        p, e = dev(n), dev(ext)
        d.lagrange_to_coeff_oop_dev(col, p, stream=stream)
        d.coeff_to_extended_oop_dev(p, e, stream=stream)
        return p, e

    transcript.common_scalar(_m(pk.transcript_repr))
    for v in asg.instance:  # KZG: the public inputs are hashed as scalars, not committed
        transcript.common_scalar(_m(v))
    instance = _column_from_cells(n, dict(enumerate(asg.instance)), into=dev(n))
    # advice columns + blinding rows
    blind = synth.uniform_fr(cs.n_advice * (bf + 1), seed + 1)
    advice = []
    for j, cells in enumerate(asg.advice):
        assert not cells or max(cells) < u, "assignment reaches into the blinding rows"
        col = _column_from_cells(n, cells, into=dev(n))
        col.patch(blind[j * (bf + 1) : (j + 1) * (bf + 1)], offset=u * 32)
        advice.append(col)
    commit_many(advice, True, 0, sparse=True, inorder=True)
    check(lib.h2mi_msm_flush(), "flush")  # the bucket reductions start now, not when the host reaches the join
    # coefficient / extended forms of the advice and instance columns: no challenge enters them, so they run on the side
    # stream beside the transcript round trips, the lookup's counting sort and the grand products (see prover.py)
    side = ws.side
    side.after_library()
    advice_f = [forms(c, side.handle) for c in advice]
    if len(asg.instance) <= 16:  # a handful of public inputs: sum_r v_r * (l_0's coset rotated by r rows), no transform (round 3)
        inst_coset = dev(ext)
        vals = np.ascontiguousarray(np.stack([_m(v) for v in asg.instance])) if asg.instance else np.zeros((1, 4), dtype=np.uint64)
        check(lib.h2mi_plonk_instance_coset_dev(pk.l0.ptr, d.k, d.extended_k, vals.ctypes.data, len(asg.instance), inst_coset.ptr, side.handle), "instance coset")
        instance_f = (None, inst_coset)
    else:
        instance_f = forms(instance, side.handle)
    _write_points(ws, transcript, len(advice))
    theta = sq()
    # lookups: permuted input / table columns.  One advice column: ONE lookup of q_lookup * a (an element-wise product; zero wherever
    # the selector is off).  Several (round 4): one lookup per lookup-advice column, the column itself as input.
    A = cs.num_advice
    single = A == 1
    lk = []  # per lookup: [input rows, A', S', product]
    lk_f = []
    if cs.lookup:
        if single:
            lk_input = dev(n)
            check(lib.h2mi_fr_mul_dev(pk.fixed_values[cs.col_qlookup].ptr, advice[0].ptr, n, lk_input.ptr, None), "lookup input")
            inputs = [lk_input]
        else:
            inputs = [advice[A + l] for l in range(cs.num_lookup_advice)]
        lb = synth.uniform_fr(2 * (bf + 1) * len(inputs), seed + 4)
        for l, inp in enumerate(inputs):
            a_perm, s_perm = dev(n), dev(n)
            if gp.lookup_permute(d.k, inp, pk.table, a_perm, s_perm):
                raise ValueError("lookup input not in the table (ConstraintSystemFailure)")
            o0 = 2 * (bf + 1) * l
            a_perm.patch(lb[o0 : o0 + bf + 1], offset=u * 32)
            s_perm.patch(lb[o0 + bf + 1 : o0 + 2 * (bf + 1)], offset=u * 32)
            commit(a_perm, True, 2 * l)
            commit(s_perm, True, 2 * l + 1)
            lk.append([inp, a_perm, s_perm, None])
        check(lib.h2mi_msm_flush(), "flush")
        side.after_library()
        for _, a_perm, s_perm, _z in lk:
            lk_f.append([forms(a_perm, side.handle), forms(s_perm, side.handle)])
        _write_points(ws, transcript, 2 * len(lk))
    beta, gamma = sq(), sq()
    # vanishing argument's random polynomial: written after the grand products' commitments but dependent on nothing, so
    # its dense MSM is queued first and accumulates beside their latency-bound scans.  Result slot: after the
    # permutation sets and the lookup product, where the transcript expects it.
    random_poly = dev(n)
    check(lib.h2mi_fr_random_dev(random_poly.ptr, n, seed + 3, 0, None), "random_poly")
    commit(random_poly, False, -(-len(cs.perm_columns) // cs.chunk) + len(lk))
    # permutation argument
    col_of = {ADVICE: advice, FIXED: pk.fixed_values, INSTANCE: [instance]}
    perm_values = [col_of[kind][c] for kind, c in cs.perm_columns]
    n_sets = -(-len(perm_values) // cs.chunk)
    zs = [dev(n) for _ in range(n_sets)]
    gp.permutation_products(d.k, perm_values, list(pk.sigma_values), cs.chunk, beta, gamma, u, zs, active=pk.active_rows)
    zblind = synth.uniform_fr(n_sets * bf, seed + 2)
    zcells = (C.c_void_p * (n_sets * bf))(*[z.ptr + (u + 1 + r) * 32 for z in zs for r in range(bf)])  # every blinding row: one launch
    check(lib.h2mi_fr_patch_cells_dev(zcells, np.ascontiguousarray(zblind[: n_sets * bf]).ctypes.data, n_sets * bf, None), "z blinding rows")
    # coefficient / extended forms of the grand products: on the side stream, ordered behind the columns themselves and
    # AHEAD of their commitments' partition kernels (round 3: queued behind the commitments on the library stream, the six
    # 2^24-point transforms of the DEGREE 22 range proof started only when the partitions — starved by the random
    # polynomial's accumulation — had drained, 7 ms into the phase)
    side.after_library()
    z_f = [forms(z, side.handle) for z in zs]
    commit_many(zs, True, 0, sparse=True)
    slot = len(zs)
    lzb = synth.uniform_fr(bf * max(len(lk), 1), seed + 5)
    for l, entry in enumerate(lk):
        lz = dev(n)
        gp.lookup_product(d.k, entry[0], pk.fixed_values[cs.col_table], entry[1], entry[2], beta, gamma, u, lz)
        lz.patch(lzb[bf * l : bf * (l + 1)], offset=(u + 1) * 32)
        entry[3] = lz
        side.after_library()
        lk_f[l].append(forms(lz, side.handle))
        commit(lz, True, slot)
        slot += 1
    slot += 1  # the random polynomial's commitment, queued before the grand products (RANDOM_SLOT)
    check(lib.h2mi_msm_flush(), "flush")
    _write_points(ws, transcript, slot)
    y = sq()
    # evaluate_h and the openings read the side stream's forms.  Joined AFTER the read-back of the phase's points: the copy runs on
    # the library stream, and a join in front of it made the transcript wait for every transform instead of the bucket reductions only
    side.join_library()
    # quotient
    h = dev(ext)
    coset_of = {ADVICE: [e for _, e in advice_f], FIXED: list(pk.fixed_cosets), INSTANCE: [instance_f[1]]}
    if single:
        gp.evaluate_h_range(d, advice_f[0][1], None, pk.fixed_cosets[cs.col_q], pk.fixed_cosets[cs.col_table] if cs.lookup else None,
                            [coset_of[kind][c] for kind, c in cs.perm_columns], list(pk.sigma_cosets), [e for _, e in z_f],
                            lk_f[0][0][1] if cs.lookup else None, lk_f[0][1][1] if cs.lookup else None, lk_f[0][2][1] if cs.lookup else None,
                            pk.l0, pk.l_last, pk.l_active, beta, gamma, y, h, blinding_factors=bf,
                            lookup_selector=pk.fixed_cosets[cs.col_qlookup] if cs.lookup else None, chunk_len=cs.chunk)
    else:  # several gate columns: the general quotient kernel (one gate per column, one lookup per lookup-advice column)
        gp.evaluate_h_flex(d, [(advice_f[j][1], pk.fixed_cosets[cs.col_qs[j]]) for j in range(A)],
                           [coset_of[kind][c] for kind, c in cs.perm_columns], list(pk.sigma_cosets), [e for _, e in z_f], cs.chunk,
                           [(advice_f[A + l][1], None, pk.fixed_cosets[cs.col_table], lk_f[l][0][1], lk_f[l][1][1], lk_f[l][2][1]) for l in range(len(lk))],
                           pk.l0, pk.l_last, pk.l_active, beta, gamma, y, h, blinding_factors=bf)
    d.extended_to_coeff_dev(h)
    pieces = d.quotient_poly_degree
    commit_many([h] * pieces, False, 0, sparse=False, inorder=True, offsets=[i * n for i in range(pieces)])
    _write_points(ws, transcript, pieces)
    x = sq()
    xn = pow(x, n, R)
    rot = lambda r: x * pow(d.omega, r % n, R) % R
    x_next, x_last, x_inv = rot(1), rot(-(bf + 1)), rot(-1)
    h_poly = dev(n)
    ptrs = (C.c_void_p * pieces)(*[h.ptr + i * n * 32 for i in range(pieces)])
    sc = np.ascontiguousarray(np.stack([_m(pow(xn, i, R)) for i in range(pieces)]))
    check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, pieces, n, h_poly.ptr, None), "h_poly")
    advice_p, z_p = [p for p, _ in advice_f], [p for p, _ in z_f]
    written = [(advice_p[c], rot(r)) for c, r in cs.advice_queries] + [(pk.fixed_polys[c], rot(r)) for c, r in cs.fixed_queries]
    written.append((random_poly, x))
    written += [(sp, x) for sp in pk.sigma_polys]
    for i, zp in enumerate(z_p):
        written += [(zp, x), (zp, x_next)]
        if i + 1 < len(z_p):
            written.append((zp, x_last))
    lk_p = [tuple(p for p, _ in f3) for f3 in lk_f]  # per lookup: (A' poly, S' poly, product poly)
    for ap, sp_, lzp in lk_p:
        written += [(lzp, x), (lzp, x_next), (ap, x), (ap, x_inv), (sp_, x)]
    todo = written + [(h_poly, x)]
    evals = dev(len(todo) + 8)
    slot_of = {}
    ordered, counts, group_pts = [], [], []
    for pt in dict.fromkeys(p for _, p in todo):
        group = list(dict.fromkeys(id(poly) for poly, p in todo if p == pt))
        by_id = {id(poly): poly for poly, p in todo if p == pt}
        for c0 in range(0, len(group), 24):  # a group holds up to 24 polynomials (a four-column range circuit opens 32 at x)
            part = group[c0 : c0 + 24]
            counts.append(len(part))
            group_pts.append(_m(pt))
            for g in part:
                slot_of[(g, pt)] = len(slot_of)
                ordered.append(by_id[g])
    # every evaluation in one call (h2mi_fr_eval_polys_multi_dev: one launch set while they fit its descriptor, group by group beyond)
    gptrs = (C.c_void_p * len(ordered))(*[g.ptr for g in ordered])
    cnt_arr = (C.c_size_t * len(counts))(*counts)
    pts_l = np.ascontiguousarray(np.stack(group_pts))
    check(lib.h2mi_fr_eval_polys_multi_dev(gptrs, cnt_arr, pts_l.ctypes.data, len(counts), n, evals.ptr, None), "eval")
    ev = evals.to_numpy(shape=(len(todo) + 8, 4))
    value = {key: F.fr_from_mont_limbs(ev[i]) for key, i in slot_of.items()}
    for poly, pt in written:
        transcript.write_scalar_int(value[(id(poly), pt)])
    queries = []
    q = lambda poly, pt: queries.append((poly, pt, value[(id(poly), pt)]))
    for c, r in cs.advice_queries:
        q(advice_p[c], rot(r))
    for zp in z_p:
        q(zp, x)
        q(zp, x_next)
    for zp in reversed(z_p[:-1]):
        q(zp, x_last)
    for ap, sp_, lzp in lk_p:
        q(lzp, x)
        q(ap, x)
        q(sp_, x)
        q(ap, x_inv)
        q(lzp, x_next)
    for c, r in cs.fixed_queries:
        q(pk.fixed_polys[c], rot(r))
    for sp in pk.sigma_polys:
        q(sp, x)
    q(h_poly, x)
    q(random_poly, x)

    def commit_and_write(poly):
        # a lone commitment, read back at once: in order on one stream, nothing deferred (h2mi_msm_bn254_g1_inorder_dev)
        check(lib.h2mi_msm_bn254_g1_inorder_dev(params.g_handle, poly.ptr + params.lo * 32, params.n, out_base, None), "commit")
        _write_points(ws, transcript, 1)

    ws.shplonk.create_proof(transcript, queries, commit_and_write)
    proof = transcript.finalize()
    if trace is not None:
        trace.update(theta=theta, beta=beta, gamma=gamma, y=y, x=x)
    check(lib.h2mi_sync(), "sync")
    return proof
