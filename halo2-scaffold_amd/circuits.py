"""The reference's StandardPlonk circuit as the prover sees it: column layout, cell assignments, copy constraints.

Mirror of reference src/circuits/standard_plonk.rs: `StandardPlonkConfig::configure` (:29-48: advice a, b, c with
equality enabled, fixed q_a, q_b, q_c, q_ab, constant, one gate q_a a + q_b b + q_c c + q_ab a b + constant queried
at Rotation::cur) and `StandardPlonk::synthesize` (:79-112: x at (a, 0); rows 1 and 2 compute x^2 and x^2 + 72 with x
copied into a and b).  Witness generation is a handful of field operations on the host — the control plane; the
columns themselves live on the device.
"""
from dataclasses import dataclass, field as dc_field

from . import field as F

R = F.FR_MODULUS


@dataclass
class Synthesis:
    """what `synthesize` leaves behind: sparse cell assignments per column and the copy constraints in call order"""
    advice: list = dc_field(default_factory=lambda: [dict() for _ in range(3)])  # column -> {row: value}
    fixed: list = dc_field(default_factory=lambda: [dict() for _ in range(5)])
    copies: list = dc_field(default_factory=list)  # ((column, row), (column, row)) over the permutation's columns


class StandardPlonk:
    """`StandardPlonk { x }`; x = None is `Value::unknown()` (keygen: examples/standard_plonk.rs:32)."""

    N_ADVICE, N_FIXED, N_INSTANCE = 3, 5, 0
    PERMUTATION_COLUMNS = [0, 1, 2]  # enable_equality(a), (b), (c): advice columns, in this order
    CS_DEGREE = 3                    # max(gate degree 3, permutation argument 3)
    # ConstraintSystem::blinding_factors(): max(3, max queries per advice column = 1) + 2
    BLINDING_FACTORS = 5
    # queries in the order configure() makes them (all Rotation::cur)
    ADVICE_QUERIES = [(0, 0), (1, 0), (2, 0)]
    FIXED_QUERIES = [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0)]
    A, B, C_ = 0, 1, 2
    Q_A, Q_B, Q_C, Q_AB, CONSTANT = 0, 1, 2, 3, 4

    def __init__(self, x=None):
        self.x = None if x is None else x % R

    def without_witnesses(self) -> "StandardPlonk":
        return StandardPlonk(None)

    def synthesize(self) -> Synthesis:
        s = Synthesis()
        x = self.x
        val = lambda f: None if x is None else f(x) % R

        def assign_advice(col, row, v):
            s.advice[col][row] = v

        def copy_advice(col, row):  # AssignedCell::copy_advice: assign, then constrain_equal(new cell, x's cell)
            assign_advice(col, row, x)
            s.copies.append(((col, row), (self.A, 0)))

        assign_advice(self.A, 0, x)
        # row 1: | x | x | x^2 | q_c = -1, q_ab = 1
        copy_advice(self.A, 1)
        copy_advice(self.B, 1)
        assign_advice(self.C_, 1, val(lambda t: t * t))
        s.fixed[self.Q_C][1] = R - 1
        s.fixed[self.Q_AB][1] = 1
        # row 2: | x | x | x^2 + 72 | q_c = -1, q_ab = 1, constant = 72
        copy_advice(self.A, 2)
        copy_advice(self.B, 2)
        assign_advice(self.C_, 2, val(lambda t: t * t + 72))
        s.fixed[self.Q_C][2] = R - 1
        s.fixed[self.Q_AB][2] = 1
        s.fixed[self.CONSTANT][2] = 72
        return s


class PermutationAssembly:
    """plonk/permutation/keygen.rs `Assembly`: the copy constraints as a permutation of the cells of the equality-
    enabled columns.  `mapping[(col, row)]` = next cell of the cycle (identity entries are not stored, so the cost is
    proportional to the number of constrained cells, not to n)."""

    def __init__(self):
        self.mapping, self.aux, self.sizes = {}, {}, {}

    def _m(self, c):
        return self.mapping.get(c, c)

    def copy(self, left, right):
        lcyc, rcyc = self.aux.get(left, left), self.aux.get(right, right)
        if lcyc == rcyc:
            return
        if self.sizes.get(lcyc, 1) < self.sizes.get(rcyc, 1):
            lcyc, rcyc = rcyc, lcyc
        self.sizes[lcyc] = self.sizes.get(lcyc, 1) + self.sizes.get(rcyc, 1)
        i = rcyc
        while True:  # merge the smaller cycle into the larger
            self.aux[i] = lcyc
            i = self._m(i)
            if i == rcyc:
                break
        self.mapping[left], self.mapping[right] = self._m(right), self._m(left)
