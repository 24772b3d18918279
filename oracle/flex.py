"""CPU oracle: keygen / create_proof / verify_proof for a small family of PLONKish constraint systems, data-driven — the
generic engine behind the halo2-lib shapes the reference proves through `scaffold::prove`
(src/scaffold.rs:246-366: GateWithInstanceCircuitBuilder / RangeWithInstanceCircuitBuilder, :379-485; closures
examples/halo2_lib.rs:14-60 and examples/range.rs:10-34), and — as a cross-check of the engine itself — the StandardPlonk
circuit, whose proofs must come out byte-identical to oracle/prover.py's.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: halo2_proofs (plonk/{keygen,prover,verifier}.rs,
plonk/lookup/*, plonk/permutation/*) and halo2-base (FlexGateConfig's vertical gate q (a + a(wX) a(w^2 X) - a(w^3 X)),
RangeConfig's lookup-advice column, the constants column, the cell layout of load_witness / mul / add / mul_add /
range_check) are restated from memory; neither crate is available here.  Pinned by `verify` accepting exactly the
consistent proofs (every challenge re-derived, gate / permutation / lookup expressions recomputed from the opened
evaluations, SHPLONK's final equation by pairing or by the known SRS secret).

A constraint system here = columns (advice / fixed / instance), gates (Python callables over a query function), the
equality-enabled columns in argument order, single-expression lookups (input = a product of columns, e.g. selector *
advice, in a table column), the explicit
query lists (their order is the order create_proof writes evaluations in), degree and blinding_factors.
rng stand-in (the reference passes OsRng): SplitMix64 streams seed+1 advice blinding (column-major, bf + 1 rows),
seed+2 permutation-product blinding (set-major, bf rows), seed+3 random polynomial, seed+4 permuted lookup columns
(A' then S', bf + 1 rows each, lookup-major), seed+5 lookup-product blinding (bf rows per lookup).
"""
from __future__ import annotations

import hashlib
import struct

from . import bn254 as o
from . import formats as fmt
from . import lookup as L
from . import plonk as P
from .prover import (ProofReader, construct_intermediate_sets, div_by_vanishing, evaluate_vanishing_polynomial, lagrange_interpolate,
                     rotate_omega)

R = o.R
ADVICE, FIXED, INSTANCE = "advice", "fixed", "instance"


class ConstraintSystem:
    def __init__(self, name, n_advice, n_fixed, n_instance, gates, perm_columns, lookups, advice_queries, fixed_queries, instance_queries, degree,
                 blinding_factors):
        self.name = name
        self.n_advice, self.n_fixed, self.n_instance = n_advice, n_fixed, n_instance
        self.gates = gates                    # callables g(q) -> value, q(kind, column, rotation)
        self.perm_columns = perm_columns      # [(kind, column)] in the order enable_equality was called
        self.lookups = lookups                # [([(kind, column) factors of the input expression], (kind, column) table)]
        self.advice_queries, self.fixed_queries, self.instance_queries = advice_queries, fixed_queries, instance_queries
        self.degree, self.blinding_factors = degree, blinding_factors

    @property
    def chunk(self):
        return self.degree - 2


def standard_plonk_cs():
    gate = lambda q: (q(FIXED, 0, 0) * q(ADVICE, 0, 0) + q(FIXED, 1, 0) * q(ADVICE, 1, 0) + q(FIXED, 2, 0) * q(ADVICE, 2, 0)
                      + q(FIXED, 3, 0) * q(ADVICE, 0, 0) * q(ADVICE, 1, 0) + q(FIXED, 4, 0)) % R
    return ConstraintSystem("standard_plonk", 3, 5, 0, [gate], [(ADVICE, 0), (ADVICE, 1), (ADVICE, 2)], [], [(0, 0), (1, 0), (2, 0)],
                            [(c, 0) for c in range(5)], [], 3, 5)


def flex_gate_cs(lookup: bool):
    """halo2-base's builders at the sizes of the reference's examples: ONE gate advice column.  Column order as configure()
    allocates them [RECALL]: RangeConfig::configure takes the lookup table column first, FlexGateConfig::configure then the
    constants column (enable_equality at once: first column of the permutation argument) and the gate advice column with
    its simple selector; the scaffold adds the instance column last (src/scaffold.rs:394-395, 449-450).  With a single
    advice column the Range builder does not add a lookup-advice column: it looks up q_lookup * a with a complex selector
    enabled on the rows of the cells to look up (the reference hands `range.q_lookup` to sub_synthesize:
    src/scaffold.rs:464-469).  keygen appends the selectors' fixed columns: complex selectors first (own column each),
    then the combined simple selectors.
      Gate builder:  fixed 0 = constants, 1 = q_enable;                          degree 3, permutation sets of 1
      Range builder: fixed 0 = table, 1 = constants, 2 = q_lookup, 3 = q_enable;  lookup degree 2 + 2 + 1 = 5: sets of 3, 4 h pieces
    Queries in creation order: enable_equality queries its column at Rotation::cur, the gate queries a at 0..3, the lookup
    its table; selector columns are queried when keygen substitutes them."""
    if lookup:
        TABLE, CONST, QL, QE = 0, 1, 2, 3
        fix_q = [(CONST, 0), (TABLE, 0), (QL, 0), (QE, 0)]
    else:
        CONST, QE = 0, 1
        fix_q = [(CONST, 0), (QE, 0)]
    gate = lambda q: q(FIXED, QE, 0) * (q(ADVICE, 0, 0) + q(ADVICE, 0, 1) * q(ADVICE, 0, 2) - q(ADVICE, 0, 3)) % R
    perm = [(FIXED, CONST), (ADVICE, 0), (INSTANCE, 0)]
    adv_q = [(0, 0), (0, 1), (0, 2), (0, 3)]
    # blinding_factors = max(3, most queries on one advice column = 4) + 2
    cs = ConstraintSystem("range" if lookup else "flex_gate", 1, 4 if lookup else 2, 1, [gate], perm,
                          [([(FIXED, QL), (ADVICE, 0)], (FIXED, TABLE))] if lookup else [], adv_q, fix_q, [(0, 0)], 5 if lookup else 3, 6)
    cs.col_const, cs.col_q = CONST, QE
    cs.col_table, cs.col_qlookup = (TABLE, QL) if lookup else (None, None)
    return cs


def flex_multi_cs(lookup: bool, num_advice: int, num_lookup_advice: int = 0, num_fixed: int = 1):
    """halo2-base's builders when the cells do NOT fit one column: `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268)
    then configures num_advice = ceil(cells / (2^k - minimum_rows)) gate columns and, for the Range builder, num_lookup_advice =
    ceil(cells_to_lookup / (2^k - minimum_rows)) lookup-advice columns [RECALL halo2-base 0.3 gates/builder.rs, flex_gate.rs, range.rs]:
      * FlexGateConfig::configure: the constants column first (enable_equality), then per gate column BasicGateConfig::configure —
        advice column (enable_equality), its own simple selector, the vertical gate on it;
      * RangeConfig::configure: the table column before everything; with more than one gate column there is no q_lookup — each
        lookup-advice column (advice, enable_equality, allocated after the gate columns) gets its own lookup argument
        (input = the column, degree 1; table = the table column): constraint-system degree 2 + 1 + 1 = 4, permutation chunks of two;
      * keygen appends one fixed column per simple selector (the gates of different columns are enabled on the same rows, so
        compress_selectors cannot merge them), after the user's fixed columns;
      * the scaffold adds the instance column last;
      * num_fixed = ceil(distinct constants / 2^k) constants columns (each enable_equality, allocated one after the other where the
        single one was); assign_constants deals the distinct constants out round-robin: constant i to column i mod num_fixed, row
        i div num_fixed.
    num_advice = 1 is flex_gate_cs (the q_lookup form): not built here."""
    assert num_advice >= 2 and (lookup or num_lookup_advice == 0) and num_fixed >= 1  # Range builder without a looked-up cell: no lookup-advice column, no lookup argument
    A, Lc = num_advice, num_lookup_advice
    TABLE, CONST = (0, 1) if lookup else (None, 0)
    CONSTS = list(range(CONST, CONST + num_fixed))
    q0 = CONST + num_fixed
    gate_of = lambda j: (lambda q: q(FIXED, q0 + j, 0) * (q(ADVICE, j, 0) + q(ADVICE, j, 1) * q(ADVICE, j, 2) - q(ADVICE, j, 3)) % R)
    gates = [gate_of(j) for j in range(A)]
    perm = [(FIXED, c) for c in CONSTS] + [(ADVICE, j) for j in range(A + Lc)] + [(INSTANCE, 0)]
    adv_q = [(j, r) for j in range(A) for r in range(4)] + [(A + l, 0) for l in range(Lc)]
    fix_q = [(c, 0) for c in CONSTS] + ([(TABLE, 0)] if Lc else []) + [(q0 + j, 0) for j in range(A)]  # the table column is queried by the lookups only
    lookups = [([(ADVICE, A + l)], (FIXED, TABLE)) for l in range(Lc)]
    cs = ConstraintSystem(f"{'range' if lookup else 'flex_gate'}_{A}x{Lc}" + (f"x{num_fixed}" if num_fixed > 1 else ""), A + Lc, q0 + A, 1, gates, perm,
                          lookups, adv_q, fix_q, [(0, 0)], 4 if Lc else 3, 6)
    cs.col_const, cs.col_table, cs.col_qlookup = CONST, TABLE, None
    cs.col_consts, cs.num_fixed = CONSTS, num_fixed
    cs.col_q = [q0 + j for j in range(A)]
    cs.num_advice, cs.num_lookup_advice = A, Lc
    return cs


def multi_column_counts(n_cells: int, n_lookup_cells: int, k: int, minimum_rows: int = 9):
    """GateThreadBuilder::config: columns needed for this many cells at 2^k rows -> (num_advice, num_lookup_advice)"""
    max_rows = (1 << k) - minimum_rows
    return -(-n_cells // max_rows), -(-n_lookup_cells // max_rows)


def num_fixed_columns(t, k: int) -> int:
    """GateThreadBuilder::config: `num_fixed = (total_fixed + (1 << k) - 1) >> k` over the DISTINCT constants — all 2^k rows counted,
    not the usable ones, so 26 .. 32 constants at DEGREE 5 get one column and keygen then fails (NotEnoughRowsAvailable) [RECALL]"""
    total_fixed = len({v for kind, v in t.rows if kind == K_})
    return max(1, -(-total_fixed // (1 << k)))


class Assignment:
    """what synthesize leaves: sparse cells per column, copy constraints in call order, the public inputs"""

    def __init__(self, cs):
        self.cs = cs
        self.advice = [dict() for _ in range(cs.n_advice)]
        self.fixed = [dict() for _ in range(cs.n_fixed)]
        self.instance = [[] for _ in range(cs.n_instance)]
        self.copies = []  # ((kind, column, row), (kind, column, row)): constrain_equal(left, right)


# ---- the halo2-lib closures of the reference as explicit cell tables [RECALL halo2-base 0.3 gates/flex_gate.rs, range.rs] ---------
# Written from halo2-base's own description of what each instruction assigns — NOT from the product's host (flex.Context appends
# cells through an imperative Context; here every closure is a literal table of rows with closed-form row indices, so the tests'
# `asg.advice == oasg.advice` compares two independently written layouts):
#   GateChip::mul(a, b)       assign_region_last([Constant(0), a, b, Witness(a b)], [0])
#   GateChip::add(a, b)       assign_region_last([a, b, Constant(1), Witness(a + b)], [0])
#   GateChip::mul_add(a,b,c)  assign_region_last([c, a, b, Witness(a b + c)], [0])
#   GateChip::inner_product(a, b) with b[0] = Constant(1): cells a_0, then (a_i, b_i, running sum) for i >= 1, gates on rows 0, 3, 6, ...
#                             otherwise Constant(0), then (a_i, b_i, running sum) for i >= 0
#   RangeChip::range_check(a, bits): k = ceil(bits / lookup_bits) limbs through inner_product(limbs, [1, 2^lb, 2^(2 lb), ...]);
#                             constrain_equal(a, acc); "the progression of indices is 0, 1, 4, ..., 4 + 3 i" for cells_to_lookup;
#                             rem = bits % lookup_bits: 1 -> assert_bit(last limb) = | 0 | x | x | x |, > 1 -> mul(last limb,
#                             Constant(2^(lookup_bits - rem))) whose output is looked up too
#   Context: an Existing(cell) input re-assigns the value and records (new cell, cell) in advice_equality_constraints, a
#   Constant(c) input records (c, new cell) in constant_equality_constraints; assign_all applies the advice equalities in
#   order, assign_constants gives every distinct constant ONE fixed cell (first-appearance order) and ties its uses to it;
#   the scaffold's wrapper then constrains the public cells to the instance column (src/scaffold.rs:411, 480).
W_, E_, K_ = "witness", "existing", "constant"


class _Table:
    """a closure as data: rows of (kind, value, source row for Existing), gate rows, extra equalities, looked-up rows"""

    def __init__(self):
        self.rows, self.gates, self.lookups = [], [], []
        self.events = []  # advice_equality_constraints in chronological order: (new row, source row)

    def put(self, cells, gate_offsets=()):
        """cells: [(kind, value or source row)]; -> row of the first cell"""
        base = len(self.rows)
        for kind, v in cells:
            if kind == E_:
                self.events.append((len(self.rows), v))
                self.rows.append((E_, self.rows[v][1]))
            else:
                self.rows.append((kind, v % R))
        self.gates += [base + g for g in gate_offsets]
        return base

    def value(self, row):
        return self.rows[row][1]

    def assignment(self, cs, public_rows):
        asg = Assignment(cs)
        asg.advice[0] = {r: v for r, (_, v) in enumerate(self.rows)}
        for g in self.gates:
            asg.fixed[cs.col_q][g] = 1
        asg.copies = [((ADVICE, 0, new), (ADVICE, 0, src)) for new, src in self.events]
        first_use = {}
        for r, (kind, v) in enumerate(self.rows):
            if kind == K_:
                first_use.setdefault(v, len(first_use))
        for v, slot in first_use.items():
            asg.fixed[cs.col_const][slot] = v
        asg.copies += [((ADVICE, 0, r), (FIXED, cs.col_const, first_use[v])) for r, (kind, v) in enumerate(self.rows) if kind == K_]
        for r in self.lookups:  # single advice column: q_lookup on the cell's own row
            asg.fixed[cs.col_qlookup][r] = 1
        for i, r in enumerate(public_rows):
            asg.instance[0].append(self.value(r))
            asg.copies.append(((ADVICE, 0, r), (INSTANCE, 0, i)))
        return asg


def break_point_rows(t: _Table, k: int, minimum_rows: int = 9):
    """the rows at which halo2-base's assign_threads leaves a gate column (`break_points[0]`, what scaffold::gen_key returns next to
    the proving key, src/scaffold.rs:95-155) [RECALL]: the row offset of every cell that is assigned a second time at row 0 of the
    next column — the same scan as multi_column_assignment's break indices, reported per column instead of per cell"""
    max_rows = (1 << k) - minimum_rows
    gate_start = set(t.gates)
    rows, first = [], 0
    for i in range(len(t.rows)):
        r = i - first
        if (i in gate_start and r + 4 > max_rows) or r >= max_rows - 1:
            rows.append(r)
            first = i
    return rows


def multi_column_assignment(t: _Table, cs, public_rows, k: int, minimum_rows: int = 9):
    """the flat cell list of a closure laid over cs.num_advice gate columns as halo2-base's assign_all does [RECALL builder.rs]:
    cells go down the current column; after placing a cell at row r the column is full when r >= max_rows - 1, or when the cell
    STARTS a gate that would not fit (r + 4 > max_rows): the cell is then assigned AGAIN at row 0 of the next column (two gates may
    overlap at it) and tied to its first copy, and a gate that starts at it is enabled on the new column.  Cells to look up are
    copied, in order, into the lookup-advice columns (max_rows cells each) and tied to their originals.  constrain_equal calls in
    the order halo2-base issues them: break copies as they happen, the lookup copies, then the closure's own equalities, the
    constants, the public cells.
    This is written differently from the product's flex.Context on purpose: positions come from a precomputed list of break
    indices (closed form below), not from a running row counter."""
    A, Lc = cs.num_advice, cs.num_lookup_advice
    max_rows = (1 << k) - minimum_rows
    gate_start = set(t.gates)
    # break indices: cell i ends a column iff its row r_i satisfies r_i >= max_rows - 1 or (i starts a gate and r_i + 4 > max_rows).
    # r_i = i - (index of the cell at row 0 of i's column); found by one scan over the cells
    breaks, first = [], 0
    for i in range(len(t.rows)):
        r = i - first
        if (i in gate_start and r + 4 > max_rows) or r >= max_rows - 1:
            breaks.append(i)
            first = i  # the copy of cell i sits at row 0 of the next column: cell i + 1 lands on row 1 = (i + 1) - i
    if len(breaks) + 1 > A:
        raise ValueError(f"NOT ENOUGH ADVICE COLUMNS: {len(breaks) + 1} needed, {A} configured")
    col_of = lambda i: sum(1 for b in breaks if b < i)       # column of the FIRST copy of cell i
    start_of = lambda c: 0 if c == 0 else breaks[c - 1]       # flat index of the cell at row 0 of column c
    row_of = lambda i: i - start_of(col_of(i))
    asg = Assignment(cs)
    copies_break = []
    for i, (_, v) in enumerate(t.rows):
        asg.advice[col_of(i)][row_of(i)] = v
        if i in breaks:
            c = col_of(i) + 1
            asg.advice[c][0] = v
            copies_break.append(((ADVICE, c, 0), (ADVICE, col_of(i), row_of(i))))
    for g in t.gates:  # a gate that starts at a break cell lives on the NEW column
        if g in breaks:
            asg.fixed[cs.col_q[col_of(g) + 1]][0] = 1
        else:
            asg.fixed[cs.col_q[col_of(g)]][row_of(g)] = 1
    where = lambda i: (ADVICE, col_of(i), row_of(i))  # assigned_advices keeps the first copy
    copies_lookup = []
    for pos, i in enumerate(t.lookups):
        lc, lr = divmod(pos, max_rows)
        if lc >= Lc:
            raise ValueError("NOT ENOUGH LOOKUP ADVICE COLUMNS")
        asg.advice[A + lc][lr] = t.value(i)
        copies_lookup.append((where(i), (ADVICE, A + lc, lr)))
    first_use = {}
    for r, (kind, v) in enumerate(t.rows):
        if kind == K_:
            first_use.setdefault(v, len(first_use))
    F_ = len(cs.col_consts)
    const_cell = lambda slot: (FIXED, cs.col_consts[slot % F_], slot // F_)  # dealt out round-robin over the constants columns
    for v, slot in first_use.items():
        asg.fixed[const_cell(slot)[1]][const_cell(slot)[2]] = v
    asg.copies = copies_break + copies_lookup
    asg.copies += [(where(new), where(src)) for new, src in t.events]
    asg.copies += [(where(r), const_cell(first_use[v])) for r, (kind, v) in enumerate(t.rows) if kind == K_]
    for i, r in enumerate(public_rows):
        asg.instance[0].append(t.value(r))
        asg.copies.append((where(r), (INSTANCE, 0, i)))
    return asg


def standard_plonk_assignment(cs, x):
    """src/circuits/standard_plonk.rs:83-108 as cells (the same rows oracle/plonk.py::StandardPlonkInstance fills)"""
    asg = Assignment(cs)
    x %= R
    asg.advice[0].update({0: x, 1: x, 2: x})
    asg.advice[1].update({1: x, 2: x})
    asg.advice[2].update({1: x * x % R, 2: (x * x + 72) % R})
    asg.fixed[2].update({1: R - 1, 2: R - 1})
    asg.fixed[3].update({1: 1, 2: 1})
    asg.fixed[4].update({2: 72})
    asg.copies = [((ADVICE, l[0], l[1]), (ADVICE, r[0], r[1])) for l, r in P.STANDARD_PLONK_COPIES]
    return asg


def halo2_lib_assignment(cs, x):
    """examples/halo2_lib.rs:14-60: x^2 + 72 three ways; public: x and out.  Seventeen rows:
         0        x                       load_witness
         1 ..  4  0, x, x, x^2            gate.mul(x, x)                          gate on row 1
         5 ..  8  x^2, 72, 1, x^2 + 72    gate.add(x_sq, Constant(72))            gate on row 5
         9 .. 12  72, x, x, x^2 + 72      assign_region_last([...], [0])          gate on row 9
        13 .. 16  72, x, x, x^2 + 72      gate.mul_add(x, x, Constant(72))        gate on row 13"""
    return _halo2_lib_table(x).assignment(cs, [0, 8])


def _halo2_lib_table(x):
    x %= R
    t = _Table()
    t.put([(W_, x)])
    t.put([(K_, 0), (E_, 0), (E_, 0), (W_, x * x)], [0])
    t.put([(E_, 4), (K_, 72), (K_, 1), (W_, x * x + 72)], [0])
    t.put([(K_, 72), (E_, 0), (E_, 0), (W_, x * x + 72)], [0])
    t.put([(K_, 72), (E_, 0), (E_, 0), (W_, x * x + 72)], [0])
    assert len(t.rows) == 17 and t.gates == [1, 5, 9, 13]
    return t


def range_assignment(cs, x, lookup_bits, n):
    """examples/range.rs:10-34: x public, range_check(x, 64) with LOOKUP_BITS limbs, then x + x.  The table column holds
    0 .. 2^LOOKUP_BITS - 1 (RangeConfig::load_lookup_table), zero elsewhere.  Rows: 0 = x; the inner product occupies rows
    1 .. 3k - 1 (k limbs): limb 0 at row 1, limb i >= 1 at row 3 i - 1, its base 2^(i lb) at row 3 i, the running sum at row
    3 i + 1, gates on rows 1, 4, 7, ...; then the remainder cells, then [x, x, 1, 2x]."""
    assert n >= 1 << lookup_bits
    asg = _range_table(x, lookup_bits).assignment(cs, [0])
    asg.fixed[cs.col_table] = {i: i for i in range(1 << lookup_bits)}
    return asg


def _range_table(x, lookup_bits):
    assert 0 <= x < 1 << 64
    k = -(-64 // lookup_bits)
    limb = lambda i: (x >> (lookup_bits * i)) & ((1 << lookup_bits) - 1)
    partial = lambda i: x & ((1 << (lookup_bits * (i + 1))) - 1)  # sum of limbs 0 .. i with their bases
    t = _Table()
    t.put([(W_, x)])
    cells = [(W_, limb(0))]
    for i in range(1, k):
        cells += [(W_, limb(i)), (K_, 1 << (lookup_bits * i)), (W_, partial(i))]
    base = t.put(cells, [3 * j for j in range(k - 1)])
    assert base == 1
    limb_row = lambda i: 1 if i == 0 else 3 * i - 1
    acc_row = 3 * (k - 1) + 1 if k > 1 else 1
    assert len(t.rows) == acc_row + 1 and t.value(acc_row) == x
    t.events.append((0, acc_row))  # ctx.constrain_equal(&a, &acc)
    t.lookups = [limb_row(i) for i in range(k)]
    rem = 64 % lookup_bits
    top = limb_row(k - 1)
    if rem == 1:
        t.put([(K_, 0), (E_, top), (E_, top), (E_, top)], [0])
    elif rem > 1:
        shift = 1 << (lookup_bits - rem)
        r0 = t.put([(K_, 0), (E_, top), (K_, shift), (W_, limb(k - 1) * shift)], [0])
        t.lookups.append(r0 + 3)
    t.put([(E_, 0), (E_, 0), (K_, 1), (W_, 2 * x)], [0])
    return t


RANGE_MANY_STEP = 0x9E3779B97F4A7C15


def range_many_values(x, count):
    """the witnesses of the `count`-fold range closure: x, x + step, x + 2 step, ... modulo 2^64"""
    return [(x + i * RANGE_MANY_STEP) & ((1 << 64) - 1) for i in range(count)]


def _range_many_table(xs, lookup_bits):
    """examples/range.rs's body once per value — load_witness(x_i); range_check(x_i, 64); x_i + x_i — in one context, every x_i public:
    the circuit that fills SEVERAL lookup-advice columns while its constants (the limb bases, shared by all the checks) still fit
    one fixed column.  Same instruction tables as _range_table, at running row offsets; -> (table, public rows)."""
    k = -(-64 // lookup_bits)
    t = _Table()
    publics = []
    for x in xs:
        assert 0 <= x < 1 << 64
        limb = lambda i: (x >> (lookup_bits * i)) & ((1 << lookup_bits) - 1)
        partial = lambda i: x & ((1 << (lookup_bits * (i + 1))) - 1)
        r0 = t.put([(W_, x)])
        publics.append(r0)
        cells = [(W_, limb(0))]
        for i in range(1, k):
            cells += [(W_, limb(i)), (K_, 1 << (lookup_bits * i)), (W_, partial(i))]
        base = t.put(cells, [3 * j for j in range(k - 1)])
        limb_row = lambda i: base + (0 if i == 0 else 3 * i - 2)
        acc_row = base + (3 * (k - 1) if k > 1 else 0)
        assert len(t.rows) == acc_row + 1 and t.value(acc_row) == x
        t.events.append((r0, acc_row))
        t.lookups += [limb_row(i) for i in range(k)]
        rem = 64 % lookup_bits
        top = limb_row(k - 1)
        if rem == 1:
            t.put([(K_, 0), (E_, top), (E_, top), (E_, top)], [0])
        elif rem > 1:
            shift = 1 << (lookup_bits - rem)
            q0 = t.put([(K_, 0), (E_, top), (K_, shift), (W_, limb(k - 1) * shift)], [0])
            t.lookups.append(q0 + 3)
        t.put([(E_, r0), (E_, r0), (K_, 1), (W_, 2 * x)], [0])
    return t, publics


def range_many_assignment_multi(cs, x, lookup_bits, k, count, minimum_rows=9):
    """`count` range checks (range_many_values) over cs.num_advice gate columns and cs.num_lookup_advice lookup-advice columns"""
    t, publics = _range_many_table(range_many_values(x, count), lookup_bits)
    asg = multi_column_assignment(t, cs, publics, k, minimum_rows)
    asg.fixed[cs.col_table] = {i: i for i in range(1 << lookup_bits)}
    return asg


def halo2_lib_assignment_multi(cs, x, k, minimum_rows=9):
    """the halo2_lib closure over cs.num_advice columns (a DEGREE at which its 17 cells overflow one)"""
    return multi_column_assignment(_halo2_lib_table(x), cs, [0, 8], k, minimum_rows)


def range_assignment_multi(cs, x, lookup_bits, k, minimum_rows=9):
    """the range closure over cs.num_advice gate columns and cs.num_lookup_advice lookup-advice columns"""
    asg = multi_column_assignment(_range_table(x, lookup_bits), cs, [0], k, minimum_rows)
    asg.fixed[cs.col_table] = {i: i for i in range(1 << lookup_bits)}
    return asg


# ---- keygen ----------------------------------------------------------------------------------------------------------
def _assembly(cs, copies):
    """permutation/keygen.rs Assembly::copy on (argument column index, row) -> {cell: next cell of its cycle}"""
    index = {col: j for j, col in enumerate(cs.perm_columns)}
    asm = {}
    aux, sizes = {}, {}
    g = lambda d, c: d.get(c, c)
    for left, right in copies:
        lc_, rc_ = (index[(left[0], left[1])], left[2]), (index[(right[0], right[1])], right[2])
        lcy, rcy = g(aux, lc_), g(aux, rc_)
        if lcy == rcy:
            continue
        if sizes.get(lcy, 1) < sizes.get(rcy, 1):
            lcy, rcy = rcy, lcy
        sizes[lcy] = sizes.get(lcy, 1) + sizes.get(rcy, 1)
        i = rcy
        while True:
            aux[i] = lcy
            i = g(asm, i)
            if i == rcy:
                break
        asm[lc_], asm[rc_] = g(asm, rc_), g(asm, lc_)
    return asm


def _vk_transcript_repr(k, degree, commitments):
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    s = bytearray(struct.pack("<II", k, degree))
    for c in commitments:
        s += fmt.g1_to_bytes(c)
    h.update(struct.pack("<Q", len(s)))
    h.update(bytes(s))
    return int.from_bytes(h.digest(), "little") % R


class VerifierKeys:
    """the verifying-key part of `Keys` in closed form, in time proportional to the assigned cells instead of n (so
    that `verify` can check device proofs at sizes where building the columns in Python takes minutes): with
    L_i(s) = (s^n - 1) / n * w^i / (s - w^i), a fixed column commits to sum cells[i] L_i(s), and the identity
    permutation column j to DELTA^j s (sum_i w^i L_i(X) = X), corrected at the cells the copy constraints move."""

    def __init__(self, cs, k, s, asg_fixed, copies):
        check_rows_available(cs, k, asg_fixed, copies)
        self.cs, self.k, self.n, self.s = cs, k, 1 << k, s
        n = self.n
        self.dom = o.Domain(k, cs.degree)
        w = self.dom.omega
        sn1 = (pow(s, n, R) - 1) * pow(n, -1, R) % R
        cache = {}

        def lag(i):
            if i not in cache:
                wi = pow(w, i, R)
                cache[i] = sn1 * wi % R * pow((s - wi) % R, -1, R) % R
            return cache[i]

        self.fixed_commitments = [o.g1_mul(sum(v * lag(i) for i, v in cells.items()) % R, o.G1_GEN) for cells in asg_fixed]
        ident = lambda j, i: pow(P.FR_DELTA, j, R) * pow(w, i, R) % R
        acc = [pow(P.FR_DELTA, j, R) * s % R for j in range(len(cs.perm_columns))]
        for (j, i), t in _assembly(cs, copies).items():
            acc[j] = (acc[j] + (ident(*t) - ident(j, i)) * lag(i)) % R
        self.permutation_commitments = [o.g1_mul(a, o.G1_GEN) for a in acc]
        self.transcript_repr = _vk_transcript_repr(k, cs.degree, self.fixed_commitments + self.permutation_commitments)


def check_rows_available(cs, k, asg_fixed, copies):
    """keygen's Assembly (plonk/keygen.rs assign_fixed / copy) [RECALL]: a fixed cell or a copy constraint on a row outside the usable
    rows is Error::NotEnoughRowsAvailable — the blinding rows belong to no column, and the permutation argument's product does not
    run over them (a cycle through such a row would simply not be enforced: the proof would fail to verify)."""
    u = (1 << k) - (cs.blinding_factors + 1)
    for c, cells in enumerate(asg_fixed):
        if cells and max(cells) >= u:
            raise ValueError(f"NotEnoughRowsAvailable: fixed column {c} is assigned on row {max(cells)}, usable rows end at {u}")
    for left, right in copies:
        if left[2] >= u or right[2] >= u:
            raise ValueError(f"NotEnoughRowsAvailable: copy constraint {left} == {right} beyond the usable rows ({u})")


class Keys:
    def __init__(self, cs, k, s, asg_fixed, copies):
        self.cs, self.k, self.n, self.s = cs, k, 1 << k, s
        check_rows_available(cs, k, asg_fixed, copies)
        n = self.n
        self.dom = o.Domain(k, cs.degree)
        self.u = n - (cs.blinding_factors + 1)
        self.pw, self.lag = o.srs_scalars(k, s)
        self.fixed = [[cells.get(r, 0) for r in range(n)] for cells in asg_fixed]
        w = self.dom.omega
        self.omega_pows = [1] * n
        for i in range(1, n):
            self.omega_pows[i] = self.omega_pows[i - 1] * w % R
        m = len(cs.perm_columns)
        self.dpow = [pow(P.FR_DELTA, j, R) for j in range(m)]
        asm = _assembly(cs, copies)
        ident = lambda j, i: self.dpow[j] * self.omega_pows[i] % R
        self.sigma = [[ident(j, i) for i in range(n)] for j in range(m)]
        for (j, i), (tj, ti) in asm.items():
            self.sigma[j][i] = ident(tj, ti)
        self.l0 = [1] + [0] * (n - 1)
        self.l_last = [0] * n
        self.l_last[self.u] = 1
        self.l_active = [1 if i < self.u else 0 for i in range(n)]
        d = self.dom
        self.fixed_polys = [d.lagrange_to_coeff(c) for c in self.fixed]
        self.sigma_polys = [d.lagrange_to_coeff(c) for c in self.sigma]
        self.fixed_commitments = [self.commit_lagrange(c) for c in self.fixed]
        self.permutation_commitments = [self.commit_lagrange(c) for c in self.sigma]
        self.transcript_repr = self._transcript_repr()

    def commit_lagrange(self, evals):
        return o.g1_mul(sum(e * l for e, l in zip(evals, self.lag)) % R, o.G1_GEN)

    def commit(self, coeffs):
        return o.g1_mul(sum(c * p for c, p in zip(coeffs, self.pw)) % R, o.G1_GEN)

    def vk_bytes(self):
        out = bytearray(struct.pack("<II", self.k, self.cs.degree))
        for c in self.fixed_commitments + self.permutation_commitments:
            out += fmt.g1_to_bytes(c)
        return bytes(out)

    def _transcript_repr(self):
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        s = self.vk_bytes()
        h.update(struct.pack("<Q", len(s)))
        h.update(s)
        return int.from_bytes(h.digest(), "little") % R


def _rand(count, seed):
    return o.unpack(o.random_field_limbs(count, seed), R)


def _column(keys, kind, col, advice, instance_cols):
    return {ADVICE: advice, FIXED: keys.fixed, INSTANCE: instance_cols}[kind][col]


def prove(keys: Keys, asg: Assignment, seed: int) -> dict:
    cs, n, u, dom = keys.cs, keys.n, keys.u, keys.dom
    bf = cs.blinding_factors
    ev = o.eval_polynomial
    tr = fmt.Blake2bTranscript()
    tr.common_scalar(keys.transcript_repr)
    instance_cols = []
    for vals in asg.instance:  # KZG: instance values are hashed, not committed
        for v in vals:
            tr.common_scalar(v)
        instance_cols.append(list(vals) + [0] * (n - len(vals)))
    blind = iter(_rand(cs.n_advice * (bf + 1), seed + 1))
    advice = []
    for cells in asg.advice:
        col = [cells.get(r, 0) for r in range(n)]
        assert all(r < u for r in cells), "assignment reaches into the blinding rows"
        for r in range(u, n):
            col[r] = next(blind)
        advice.append(col)
    for c in advice:
        tr.write_point(keys.commit_lagrange(c))
    theta = tr.squeeze_challenge()
    col_of = lambda kc: _column(keys, kc[0], kc[1], advice, instance_cols)

    def input_of(factors):  # the lookup's input expression on the rows: a product of columns (selector * advice)
        out = [1] * n
        for kc in factors:
            out = [a * b % R for a, b in zip(out, col_of(kc))]
        return out

    lk_blind = iter(_rand(2 * (bf + 1) * max(len(cs.lookups), 1), seed + 4))
    permuted = []
    for inp, tab in cs.lookups:
        a_in, t_in = input_of(inp), col_of(tab)
        ap, sp = L.permute_expression_pair(a_in, t_in, u, [next(lk_blind) for _ in range(bf + 1)], [next(lk_blind) for _ in range(bf + 1)])
        permuted.append((ap, sp))
        tr.write_point(keys.commit_lagrange(ap))
        tr.write_point(keys.commit_lagrange(sp))
    beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge()
    # permutation products
    m = len(cs.perm_columns)
    zblind = iter(_rand(-(-m // cs.chunk) * bf, seed + 2))
    zs, start = [], 1
    for s0 in range(0, m, cs.chunk):
        z = [0] * n
        z[0] = start
        for i in range(u):
            num = den = 1
            for j in range(s0, min(m, s0 + cs.chunk)):
                v = col_of(cs.perm_columns[j])[i]
                num = num * ((v + beta * keys.dpow[j] * keys.omega_pows[i] + gamma) % R) % R
                den = den * ((v + beta * keys.sigma[j][i] + gamma) % R) % R
            z[i + 1] = z[i] * num % R * pow(den, -1, R) % R
        for r in range(u + 1, n):
            z[r] = next(zblind)
        start = z[u]
        zs.append(z)
    for z in zs:
        tr.write_point(keys.commit_lagrange(z))
    lz_blind = iter(_rand(bf * max(len(cs.lookups), 1), seed + 5))
    lzs = []
    for (inp, tab), (ap, sp) in zip(cs.lookups, permuted):
        lz = L.lookup_product(input_of(inp), col_of(tab), ap, sp, beta, gamma, u, [next(lz_blind) for _ in range(bf)])
        lzs.append(lz)
        tr.write_point(keys.commit_lagrange(lz))
    random_poly = _rand(n, seed + 3)
    tr.write_point(keys.commit(random_poly))
    y = tr.squeeze_challenge()
    # ---- quotient ------------------------------------------------------------------------------------------------
    size = 1 << dom.extended_k
    rot = size // n
    E = lambda lagr: dom.coeff_to_extended(dom.lagrange_to_coeff(lagr))
    adv_c, fix_c, ins_c = [E(c) for c in advice], [E(c) for c in keys.fixed], [E(c) for c in instance_cols]
    sig_c, z_c = [E(c) for c in keys.sigma], [E(z) for z in zs]
    lk_c = [(E(ap), E(sp), E(lz)) for (ap, sp), lz in zip(permuted, lzs)]
    l0, ll, lact = E(keys.l0), E(keys.l_last), E(keys.l_active)
    cosets = {ADVICE: adv_c, FIXED: fix_c, INSTANCE: ins_c}
    h_ext = [0] * size
    tinv = [pow((pow(dom.g_coset * pow(dom.extended_omega, i, R) % R, n, R) - 1) % R, -1, R) for i in range(rot)]
    for idx in range(size):
        rn = lambda r: (idx + r * rot) % size
        q = lambda kind, col, r: cosets[kind][col][rn(r)]
        v = 0
        for gate in cs.gates:
            v = (v * y + gate(q)) % R
        v = _permutation_terms(cs, v, y, beta, gamma, lambda kc: cosets[kc[0]][kc[1]][idx], [s[idx] for s in sig_c], [z[idx] for z in z_c],
                               [z[rn(1)] for z in z_c], [z[rn(-(bf + 1))] for z in z_c], l0[idx], ll[idx], lact[idx],
                               dom.g_coset * pow(dom.extended_omega, idx, R) % R)
        for (inp, tab), (ap, sp, lz) in zip(cs.lookups, lk_c):
            a_val = 1
            for kc in inp:
                a_val = a_val * cosets[kc[0]][kc[1]][idx] % R
            v = _lookup_terms(v, y, beta, gamma, a_val, cosets[tab[0]][tab[1]][idx], ap[idx], ap[rn(-1)], sp[idx], lz[idx],
                              lz[rn(1)], l0[idx], ll[idx], lact[idx])
        h_ext[idx] = v * tinv[idx % rot] % R
    h_coeffs = dom.extended_to_coeff(h_ext)
    pieces = [h_coeffs[i * n : (i + 1) * n] for i in range(dom.quotient_poly_degree)]
    for p in pieces:
        tr.write_point(keys.commit(p))
    x = tr.squeeze_challenge()
    xn = pow(x, n, R)
    # ---- evaluations -----------------------------------------------------------------------------------------------
    advice_polys = [dom.lagrange_to_coeff(c) for c in advice]
    z_polys = [dom.lagrange_to_coeff(z) for z in zs]
    lk_polys = [(dom.lagrange_to_coeff(ap), dom.lagrange_to_coeff(sp), dom.lagrange_to_coeff(lz)) for (ap, sp), lz in zip(permuted, lzs)]
    rx = lambda r: rotate_omega(dom, x, r)
    for c, r in cs.advice_queries:
        tr.write_scalar(ev(advice_polys[c], rx(r)))
    for c, r in cs.fixed_queries:
        tr.write_scalar(ev(keys.fixed_polys[c], rx(r)))
    h_poly = [0] * n
    for piece in reversed(pieces):
        h_poly = [(a * xn + b) % R for a, b in zip(h_poly, piece)]
    tr.write_scalar(ev(random_poly, x))
    for sp_ in keys.sigma_polys:
        tr.write_scalar(ev(sp_, x))
    x_next, x_last, x_inv = rx(1), rx(-(bf + 1)), rx(-1)
    for i, zp in enumerate(z_polys):
        tr.write_scalar(ev(zp, x))
        tr.write_scalar(ev(zp, x_next))
        if i + 1 < len(z_polys):
            tr.write_scalar(ev(zp, x_last))
    for app, spp, lzp in lk_polys:  # lookup Evaluated: product, product_next, permuted_input, permuted_input_inv, permuted_table
        for poly, pt in ((lzp, x), (lzp, x_next), (app, x), (app, x_inv), (spp, x)):
            tr.write_scalar(ev(poly, pt))
    # ---- queries, SHPLONK --------------------------------------------------------------------------------------------
    polys, queries = {}, []

    def qq(key, poly, pt):
        polys[key] = poly
        queries.append((key, pt, ev(poly, pt)))

    for c, r in cs.advice_queries:
        qq(("advice", c), advice_polys[c], rx(r))
    for i, zp in enumerate(z_polys):
        qq(("z", i), zp, x)
        qq(("z", i), zp, x_next)
    for i in reversed(range(len(z_polys) - 1)):
        qq(("z", i), z_polys[i], x_last)
    for li, (app, spp, lzp) in enumerate(lk_polys):  # lookup open: product@x, input@x, table@x, input@x_inv, product@x_next
        qq(("lz", li), lzp, x)
        qq(("la", li), app, x)
        qq(("ls", li), spp, x)
        qq(("la", li), app, x_inv)
        qq(("lz", li), lzp, x_next)
    for c, r in cs.fixed_queries:
        qq(("fixed", c), keys.fixed_polys[c], rx(r))
    for i, sp_ in enumerate(keys.sigma_polys):
        qq(("sigma", i), sp_, x)
    qq(("h",), h_poly, x)
    qq(("random",), random_poly, x)
    _shplonk_prove(keys, tr, polys, queries, n)
    return {"proof": bytes(tr.proof), "theta": theta, "beta": beta, "gamma": gamma, "y": y, "x": x, "zs": zs, "permuted": permuted, "lzs": lzs,
            "h_coeffs": h_coeffs, "advice": advice}


def _permutation_terms(cs, v, y, beta, gamma, value_of, sig, z, z_next, z_last, l0, ll, lact, X):
    if not cs.perm_columns:
        return v
    v = (v * y + (1 - z[0]) * l0) % R
    v = (v * y + (z[-1] * z[-1] - z[-1]) * ll) % R
    for s in range(1, len(z)):
        v = (v * y + (z[s] - z_last[s - 1]) * l0) % R
    cur = beta * X % R
    m = len(cs.perm_columns)
    for s, s0 in enumerate(range(0, m, cs.chunk)):
        left, right = z_next[s], z[s]
        for j in range(s0, min(m, s0 + cs.chunk)):
            val = value_of(cs.perm_columns[j])
            left = left * (val + beta * sig[j] + gamma) % R
            right = right * (val + cur + gamma) % R
            cur = cur * P.FR_DELTA % R
        v = (v * y + (left - right) * lact) % R
    return v


def _lookup_terms(v, y, beta, gamma, a_in, t_in, ap, ap_prev, sp, lz, lz_next, l0, ll, lact):
    v = (v * y + (1 - lz) * l0) % R
    v = (v * y + (lz * lz - lz) * ll) % R
    v = (v * y + (lz_next * (ap + beta) % R * (sp + gamma) - lz * (a_in + beta) % R * (t_in + gamma)) * lact) % R
    v = (v * y + (ap - sp) * l0) % R
    v = (v * y + (ap - sp) * (ap - ap_prev) % R * lact) % R
    return v


def _shplonk_prove(keys, tr, polys, queries, n):
    ev = o.eval_polynomial
    y_sh = tr.squeeze_challenge()
    rotation_sets, super_points = construct_intermediate_sets(queries)
    v = tr.squeeze_challenge()
    quotients = []
    for pts, comms in rotation_sets:
        nx, yp = [0] * n, 1
        for key, evals in comms:
            r_x = lagrange_interpolate(pts, evals)
            num = list(polys[key])
            for i, c in enumerate(r_x):
                num[i] = (num[i] - c) % R
            nx = [(a + yp * b) % R for a, b in zip(nx, num)]
            yp = yp * y_sh % R
        qx = div_by_vanishing(nx, pts)
        quotients.append(qx + [0] * (n - len(qx)))
    h_x, vp = [0] * n, 1
    for qx in quotients:
        h_x = [(a + vp * b) % R for a, b in zip(h_x, qx)]
        vp = vp * v % R
    tr.write_point(keys.commit(h_x))
    u = tr.squeeze_challenge()
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x, z_diffs, vp = [0] * n, [], 1
    for pts, comms in rotation_sets:
        z_i = evaluate_vanishing_polynomial([p for p in super_points if p not in pts], u)
        z_diffs.append(z_i)
        inner, yp = [0] * n, 1
        for key, evals in comms:
            lin = list(polys[key])
            lin[0] = (lin[0] - ev(lagrange_interpolate(pts, evals), u)) % R
            inner = [(a + yp * b) % R for a, b in zip(inner, lin)]
            yp = yp * y_sh % R
        l_x = [(a + vp * z_i % R * b) % R for a, b in zip(l_x, inner)]
        vp = vp * v % R
    l_x = [(a - zt_eval * b) % R for a, b in zip(l_x, h_x)]
    assert ev(l_x, u) == 0
    z0inv = pow(z_diffs[0], -1, R)
    h2_x = [c * z0inv % R for c in div_by_vanishing(l_x, [u])] + [0]
    tr.write_point(keys.commit(h2_x))


def verify(keys, proof: bytes, instances, g2=None, s_g2=None) -> bool:
    """plonk/verifier.rs verify_proof + VerifierSHPLONK for the constraint system of `keys` (only its verifying-key part:
    cs, k, domain, commitments, transcript_repr — and s unless the SRS's G2 elements are given)."""
    cs, n, dom = keys.cs, keys.n, keys.dom
    bf = cs.blinding_factors
    n_sets = -(-len(cs.perm_columns) // cs.chunk)
    try:
        rd = ProofReader(proof)
        rd.tr.common_scalar(keys.transcript_repr)
        for vals in instances:
            for v in vals:
                rd.tr.common_scalar(v)
        advice_c = [rd.read_point() for _ in range(cs.n_advice)]
        rd.squeeze()  # theta
        lk_perm_c = [(rd.read_point(), rd.read_point()) for _ in cs.lookups]
        beta, gamma = rd.squeeze(), rd.squeeze()
        z_c = [rd.read_point() for _ in range(n_sets)]
        lz_c = [rd.read_point() for _ in cs.lookups]
        random_c = rd.read_point()
        y = rd.squeeze()
        h_c = [rd.read_point() for _ in range(dom.quotient_poly_degree)]
        x = rd.squeeze()
        advice_evals = [rd.read_scalar() for _ in cs.advice_queries]
        fixed_evals = [rd.read_scalar() for _ in cs.fixed_queries]
        random_eval = rd.read_scalar()
        sigma_evals = [rd.read_scalar() for _ in cs.perm_columns]
        z_evals = []
        for i in range(n_sets):
            e, en = rd.read_scalar(), rd.read_scalar()
            z_evals.append((e, en, rd.read_scalar() if i + 1 < n_sets else None))
        lk_evals = [tuple(rd.read_scalar() for _ in range(5)) for _ in cs.lookups]
    except ValueError:
        return False
    xn = pow(x, n, R)
    w = dom.omega
    l_at = lambda rot: (xn - 1) * pow(n, -1, R) % R * pow(w, rot % n, R) % R * pow((x - pow(w, rot % n, R)) % R, -1, R) % R
    l_evals = [l_at(rot) for rot in range(-(bf + 1), 1)]
    l_last, l_blind, l_0 = l_evals[0], sum(l_evals[1 : 1 + bf]) % R, l_evals[1 + bf]
    lact = (1 - (l_last + l_blind)) % R
    # instance evaluations: the verifier interpolates the public inputs itself (KZG does not query instance commitments)
    instance_evals = []
    for c, r in cs.instance_queries:
        pt = rotate_omega(dom, x, r)
        ptn = pow(pt, n, R)
        instance_evals.append(sum(v * ((ptn - 1) * pow(n, -1, R) % R * pow(w, i, R) % R * pow((pt - pow(w, i, R)) % R, -1, R) % R)
                                  for i, v in enumerate(instances[c])) % R)
    evals = {}
    for qi, (c, r) in enumerate(cs.advice_queries):
        evals[(ADVICE, c, r)] = advice_evals[qi]
    for qi, (c, r) in enumerate(cs.fixed_queries):
        evals[(FIXED, c, r)] = fixed_evals[qi]
    for qi, (c, r) in enumerate(cs.instance_queries):
        evals[(INSTANCE, c, r)] = instance_evals[qi]
    q = lambda kind, col, r: evals[(kind, col, r)]
    v = 0
    for gate in cs.gates:
        v = (v * y + gate(q)) % R
    v = _permutation_terms(cs, v, y, beta, gamma, lambda kc: evals[(kc[0], kc[1], 0)], sigma_evals, [e[0] for e in z_evals], [e[1] for e in z_evals],
                           [e[2] for e in z_evals], l_0, l_last, lact, x)
    for (inp, tab), (lz, lz_next, ap, ap_inv, sp) in zip(cs.lookups, lk_evals):
        a_val = 1
        for kc in inp:
            a_val = a_val * evals[(kc[0], kc[1], 0)] % R
        v = _lookup_terms(v, y, beta, gamma, a_val, evals[(tab[0], tab[1], 0)], ap, ap_inv, sp, lz, lz_next, l_0, l_last, lact)
    expected_h = v * pow((xn - 1) % R, -1, R) % R
    h_commitment = None
    for cmt in reversed(h_c):
        h_commitment = o.g1_add(o.g1_mul(xn, h_commitment) if h_commitment else None, cmt)
    rx = lambda r: rotate_omega(dom, x, r)
    x_next, x_last, x_inv = rx(1), rx(-(bf + 1)), rx(-1)
    points, queries = {}, []

    def qq(key, cmt, pt, evl):
        points[key] = cmt
        queries.append((key, pt, evl))

    for qi, (c, r) in enumerate(cs.advice_queries):
        qq(("advice", c), advice_c[c], rx(r), advice_evals[qi])
    for i in range(n_sets):
        qq(("z", i), z_c[i], x, z_evals[i][0])
        qq(("z", i), z_c[i], x_next, z_evals[i][1])
    for i in reversed(range(n_sets - 1)):
        qq(("z", i), z_c[i], x_last, z_evals[i][2])
    for li, (lz, lz_next, ap, ap_inv, sp) in enumerate(lk_evals):
        qq(("lz", li), lz_c[li], x, lz)
        qq(("la", li), lk_perm_c[li][0], x, ap)
        qq(("ls", li), lk_perm_c[li][1], x, sp)
        qq(("la", li), lk_perm_c[li][0], x_inv, ap_inv)
        qq(("lz", li), lz_c[li], x_next, lz_next)
    for qi, (c, r) in enumerate(cs.fixed_queries):
        qq(("fixed", c), keys.fixed_commitments[c], rx(r), fixed_evals[qi])
    for i in range(len(cs.perm_columns)):
        qq(("sigma", i), keys.permutation_commitments[i], x, sigma_evals[i])
    qq(("h",), h_commitment, x, expected_h)
    qq(("random",), random_c, x, random_eval)
    rotation_sets, super_points = construct_intermediate_sets(queries)
    y_sh, vv = rd.squeeze(), rd.squeeze()
    try:
        h1 = rd.read_point()
        uu = rd.squeeze()
        h2 = rd.read_point()
    except ValueError:
        return False
    if rd.pos != len(proof):
        return False
    outer, r_outer, z_0, z_0_diff_inv, vp = None, 0, 0, 0, 1
    for i, (pts, comms) in enumerate(rotation_sets):
        z_diff = evaluate_vanishing_polynomial([p for p in super_points if p not in pts], uu)
        if i == 0:
            z_0 = evaluate_vanishing_polynomial(pts, uu)
            z_0_diff_inv = pow(z_diff, -1, R)
            z_diff = 1
        else:
            z_diff = z_diff * z_0_diff_inv % R
        inner, r_inner, yp = None, 0, 1
        for key, evs in comms:
            r_inner = (r_inner + yp * o.eval_polynomial(lagrange_interpolate(pts, evs), uu)) % R
            inner = o.g1_add(inner, o.g1_mul(yp, points[key]))
            yp = yp * y_sh % R
        r_outer = (r_outer + vp * r_inner % R * z_diff) % R
        outer = o.g1_add(outer, o.g1_mul(vp * z_diff % R, inner))
        vp = vp * vv % R
    outer = o.g1_add(outer, o.g1_mul((-r_outer) % R, o.G1_GEN))
    outer = o.g1_add(outer, o.g1_mul((-z_0) % R, h1))
    outer = o.g1_add(outer, o.g1_mul(uu, h2))
    if g2 is not None and s_g2 is not None:
        from . import pairing

        return pairing.pairing_product_is_one([(h2, s_g2), (o.g1_neg(outer), g2)]) if outer is not None else h2 is None
    return o.g1_mul(keys.s, h2) == outer
