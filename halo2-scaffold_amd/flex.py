"""keygen + create_proof for the halo2-lib constraint systems the reference proves through `scaffold::prove`
(src/scaffold.rs:246-366), with every vector resident in HBM — BASELINE configs[2] (halo2_lib.rs: x^2 + 72) and
configs[3] (range.rs: range_check(x, 64) with a LOOKUP_BITS table).  This module holds what the CALLER of the library's prover
holds: the constraint system as data (FlexGateCS.abi), the builders' cell layout (Context: witness generation on the CPU, as in the
reference), the transcript.  FlexKeys and create_proof are thin calls into libh2mi.so (engine.Keys / engine.Prover over
include/h2mi_prover.h); the device pipeline described below lives in csrc/h2mi_prover.cpp.

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
from . import engine
from . import field as F
from .domain import EvaluationDomain
from .keygen import _m, transcript_repr
from .params import ParamsKZG
from .transcript import Blake2bWrite

R = F.FR_MODULUS
ADVICE, FIXED, INSTANCE = "advice", "fixed", "instance"
_KIND = {ADVICE: engine.ADVICE, FIXED: engine.FIXED, INSTANCE: engine.INSTANCE}


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

    def __init__(self, lookup: bool, num_advice: int = 1, num_lookup_advice: int = 0, k: int = None, minimum_rows: int = 9, num_fixed: int = 1):
        self.lookup = lookup
        self.num_advice, self.num_lookup_advice, self.num_fixed = num_advice, num_lookup_advice, num_fixed
        self.k, self.minimum_rows = k, minimum_rows  # the multi-column layout needs the row budget 2^k - minimum_rows
        if num_advice > 1:
            self._init_multi()
            return
        assert num_lookup_advice == 0 and num_fixed == 1  # at most 2^k - minimum_rows cells: their constants fit one column
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
        columns share rows: compress_selectors cannot merge them); the scaffold adds the instance column last.  num_fixed constants
        columns (config: ceil(distinct constants / 2^k)), allocated where the single one was; assign_constants deals the distinct
        constants out round-robin (constant i: column i mod num_fixed, row i div num_fixed)."""
        A, Lc, Fc = self.num_advice, self.num_lookup_advice, self.num_fixed
        assert 2 <= A <= engine.MAX_GATES and Lc <= engine.MAX_LOOKUPS, "the prover ABI takes up to 32 gate columns and 8 lookup-advice columns"
        assert Lc == 0 or self.lookup, "the Gate builder has no lookup-advice column"  # Range builder, nothing looked up: Lc = 0, no lookup argument
        assert self.k is not None, "the multi-column layout needs k (rows per column = 2^k - minimum_rows)"
        self.n_advice = A + Lc
        self.col_table, self.col_const = (0, 1) if self.lookup else (None, 0)
        self.col_qlookup = None
        self.col_consts = [self.col_const + i for i in range(Fc)]
        self.col_qs = [self.col_const + Fc + j for j in range(A)]
        self.col_q = None
        self.n_fixed = self.col_const + Fc + A
        self.fixed_queries = [(c, 0) for c in self.col_consts] + ([(self.col_table, 0)] if Lc else []) + [(c, 0) for c in self.col_qs]
        self.perm_columns = [(FIXED, c) for c in self.col_consts] + [(ADVICE, j) for j in range(A + Lc)] + [(INSTANCE, 0)]
        self.advice_queries = [(j, r) for j in range(A) for r in range(4)] + [(A + l, 0) for l in range(Lc)]
        self.degree = 4 if Lc else 3  # the lookup arguments are what raises it
        self.blinding_factors = 6
        self.chunk = self.degree - 2

    def abi(self, k: int) -> engine.ConstraintSystem:
        """this constraint system as the data h2mi_prover_keygen takes: one vertical gate per gate column with its selector, the
        permutation argument's columns, the lookups (single column: q_lookup * a; several: one per lookup-advice column)"""
        A = self.num_advice
        if A == 1:
            lookups = [(0, self.col_qlookup, self.col_table)] if self.lookup else []
        else:
            lookups = [(A + l, None, self.col_table) for l in range(self.num_lookup_advice)]
        return engine.ConstraintSystem.build(
            k, self.n_advice, self.n_fixed, 1, self.degree, self.blinding_factors, engine.GATES_FLEX_VERTICAL, [(j, q) for j, q in enumerate(self.col_qs)],
            [(_KIND[kind], c) for kind, c in self.perm_columns], lookups, self.advice_queries, self.fixed_queries)


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
    num_fixed = max(1, -(-len(asg.fixed[probe.col_const]) // (1 << k)))  # `(total_fixed + (1 << k) - 1) >> k` over the distinct constants
    return FlexGateCS(lookup, num_advice, -(-looked_up // max_rows) if lookup else 0, k=k, minimum_rows=minimum_rows, num_fixed=num_fixed)


class Assignment:
    def __init__(self, cs: FlexGateCS):
        self.cs = cs
        self.advice = [dict() for _ in range(cs.n_advice)]
        self.fixed = [dict() for _ in range(cs.n_fixed)]
        self.instance = []   # public inputs (column 0)
        self.copies = []     # ((kind, column, row), (kind, column, row)) in constrain_equal order
        self.break_points = []  # rows at which a gate column ended (several gate columns only)


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
                asg.break_points.append(row)  # what scaffold::gen_key returns next to the proving key (phase 0)
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
        consts, Fc = {}, cs.num_fixed
        for r, v in self.const_cells:
            if v not in consts:
                consts[v] = len(consts)
                asg.fixed[cs.col_consts[consts[v] % Fc]][consts[v] // Fc] = v
            asg.copies.append((cell(r), (FIXED, cs.col_consts[consts[v] % Fc], consts[v] // Fc)))
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


RANGE_MANY_STEP = 0x9E3779B97F4A7C15


def range_closure(cs: FlexGateCS, x: int, lookup_bits: int, count: int = 1) -> Assignment:
    """reference examples/range.rs:10-34: make_public = [x]; range_check(x, 64); x + x.  The table column is what
    RangeConfig::load_lookup_table assigns: 0 .. 2^LOOKUP_BITS - 1 (src/scaffold.rs:462).  count > 1: the same body for
    x, x + step, x + 2 step, ... (mod 2^64) in ONE context, every value public — a circuit that fills several gate and
    lookup-advice columns while the limb bases, shared by all the checks, still fit the one constants column."""
    asg = Assignment(cs)
    ctx = Context(asg)
    public = []
    for i in range(count):
        xc = ctx.load_witness((x + i * RANGE_MANY_STEP) & ((1 << 64) - 1))
        ctx.range_check(xc, 64, lookup_bits)
        ctx.add(xc, xc)
        public.append(xc)
    ctx.finish(public)
    return load_lookup_table(asg, lookup_bits)


def load_lookup_table(asg: Assignment, lookup_bits: int) -> Assignment:
    """RangeConfig::load_lookup_table (src/scaffold.rs:462): the Range builder's synthesize assigns 0 .. 2^LOOKUP_BITS - 1 to the table
    column whatever the closure does — also for a closure that looks nothing up (the reference picks the Range builder whenever
    LOOKUP_BITS is set: src/scaffold.rs:44-48)."""
    assert asg.cs.lookup, "the Gate builder has no table column"
    asg.fixed[asg.cs.col_table] = None  # dense: filled by keygen from `table_values`
    asg.table_values = list(range(1 << lookup_bits))
    return asg


def mock(asg: Assignment, k: int = None) -> None:
    """scaffold::mock (src/scaffold.rs:205-243: MockProver::run(..).assert_satisfied()) for these constraint systems, on the
    host: every enabled row satisfies the vertical gate, every copy constraint joins equal cells, every looked-up cell is
    a table value.  Raises ValueError naming the first violation — what the reference's users run before `prove`.  With the
    DEGREE known (k, or the constraint system's own for the multi-column layouts) cells beyond the usable rows are refused as
    MockProver::run refuses them (NotEnoughRowsAvailable) — the same rule h2mi_prover_keygen applies."""
    cs = asg.cs
    k = k if k is not None else cs.k
    if k is not None:
        u = (1 << k) - (cs.blinding_factors + 1)
        rows = [(f"fixed column {c}", max(cells)) for c, cells in enumerate(asg.fixed) if cells]
        rows += [(f"advice column {c}", max(cells)) for c, cells in enumerate(asg.advice) if cells]
        rows += [("a copy constraint", max(left[2], right[2])) for left, right in asg.copies]
        if cs.lookup:
            rows.append(("the lookup table", len(asg.table_values) - 1))
        for what, row in rows:
            if row >= u:
                raise ValueError(f"NotEnoughRowsAvailable: {what} reaches row {row}, the usable rows end at {u}")
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
class FlexKeys:
    """keygen_vk + keygen_pk (src/scaffold.rs:284,287) through h2mi_prover_keygen: the fixed cells of a run of the closure (the
    lookup table dense), the copy constraints with their columns renumbered into the permutation argument's order; the library
    builds the sigma polynomials (Assembly::copy over advice, constants and instance cells alike), l_0 / l_last / l_active, the
    lookup table's sorted form and the verifying key's commitments."""

    def __init__(self, params: ParamsKZG, cs: FlexGateCS, asg: Assignment):
        self.cs = cs
        k = params.k
        self.domain = d = EvaluationDomain(cs.degree, k)
        self.u = u = d.n - (cs.blinding_factors + 1)
        fixed_cells = list(asg.fixed)
        if cs.lookup:
            tv = asg.table_values
            if len(tv) > u:
                raise ValueError(f"lookup table of {len(tv)} rows does not fit the {u} usable rows of a 2^{k} circuit (LOOKUP_BITS must be below DEGREE)")
            fixed_cells[cs.col_table] = [v % R for v in tv]  # dense: rows 0 .. len - 1
        index = {col: j for j, col in enumerate(cs.perm_columns)}
        copies = [(index[(left[0], left[1])], left[2], index[(right[0], right[1])], right[2]) for left, right in asg.copies]
        self.keys = engine.Keys(cs.abi(k), params, fixed_cells, copies)
        self.fixed_commitments, self.permutation_commitments = self.keys.fixed_commitments, self.keys.permutation_commitments
        # read-only views of the key's library-owned vectors (Lagrange / coefficient / extended-coset forms), for callers that check them
        nf, m = cs.n_fixed, len(cs.perm_columns)
        self.fixed_values, self.fixed_polys, self.fixed_cosets = (self.keys.views(kd, nf) for kd in (engine.PKBUF_FIXED, engine.PKBUF_FIXED_POLY, engine.PKBUF_FIXED_COSET))
        self.sigma_values, self.sigma_polys, self.sigma_cosets = (self.keys.views(kd, m) for kd in (engine.PKBUF_SIGMA, engine.PKBUF_SIGMA_POLY, engine.PKBUF_SIGMA_COSET))
        self._vk_bytes, self.transcript_repr = transcript_repr(k, cs.degree, self.fixed_commitments, self.permutation_commitments)

    def vk_bytes(self) -> bytes:
        return self._vk_bytes

    def release(self):
        self.keys.release()


# ---- create_proof --------------------------------------------------------------------------------------------------------
class FlexWorkspace:
    """one library prover (device buffers, streams) kept for the next proof against the same proving key (the reference's drivers
    prove repeatedly against one pk / SRS, e.g. examples/linear_regression.rs:126-195).
    combiner: a dist.PhaseCombiner with >= 8 slots when `params` is one rank's slice of the SRS (one process per GPU): every
    commitment is then this rank's partial point, combined across ranks at every transcript write."""

    def __init__(self, params: ParamsKZG, pk: "FlexKeys", combiner=None):
        self.combiner = combiner
        self.prover = engine.Prover(pk.keys, params, combiner=combiner)

    def release(self):
        self.prover.release()


def create_proof(params: ParamsKZG, pk: FlexKeys, asg: Assignment, seed: int, transcript: Blake2bWrite = None, trace: dict = None,
                 ws: FlexWorkspace = None) -> bytes:
    """create_proof for one circuit with one instance column: scaffold::prove's call (src/scaffold.rs:322-331,
    `&[&[&public_io]]`).  `params` is the whole SRS, or one rank's slice of it together with ws.combiner.  Without `ws` the
    device buffers live for this call only (released on every exit path)."""
    own = ws is None
    ws = ws or FlexWorkspace(params, pk)
    try:
        transcript = transcript or Blake2bWrite.init()
        transcript.common_scalar(_m(pk.transcript_repr))
        for v in asg.instance:  # KZG: the public inputs are hashed as scalars, not committed
            transcript.common_scalar(_m(v))
        ws.prover.drive(asg.advice, asg.instance, seed, transcript, trace)
        return transcript.finalize()
    finally:
        if own:
            ws.release()
