"""The reference's poseidon example as a FlexGate circuit: witness generation for `hash_two`
(examples/poseidon.rs:15-36: load x, y; make them public; PoseidonChip::<F, 3, 2>::new(ctx, 8, 57); update(&[x, y]);
squeeze; make the hash public), proved by flex.create_proof like the other halo2-lib shapes (BASELINE configs[4]).

Witness generation is host work in the reference too (the closure runs on the CPU and fills the Context); the device's
share is everything create_proof does with the columns afterwards.  The chip the reference links (Axiom's, adapted from
Scroll / snark-verifier: Cargo.toml:17-18) is not vendored; it computes the standard Poseidon sponge with precomputed
sparse matrices for the partial rounds.  Here the permutation is laid out directly from its definition on the same
FlexGate primitives — per round: add the round constants (mul_add cells), x^5 by three multiplications on the s-box
lanes, the MDS product as three inner products against constant cells — about 3.7 k cells per permutation, 7.4 k for
hash_two: like every halo2-lib example at DEGREE = 20 the advice column is almost empty, so the prover's work is the
same as for the optimized chip.  Parameters: the Grain LFSR of the Poseidon reference script (t, R_F, R_P as above,
alpha = 5); sponge convention [RECALL snark-verifier]: state (2^64, 0, 0), inputs added into state[1..], a short (or
the trailing empty) chunk adds 1 at the next free position, squeeze returns state[1].  Checked against
oracle/poseidon.py (itself pinned by circomlib's published constants and known answer) in tests/test_gpu_flex.py.
"""
from . import field as F
from .flex import ADVICE, Assignment, Context, FlexGateCS

R = F.FR_MODULUS
T, RATE, R_F, R_P = 3, 2, 8, 57


class _GrainLFSR:
    """80-bit shift register kept in one integer (bit 79 = oldest); x^80 + x^18 + x^29 + x^42 + x^57 + x^67 + 1 taps
    as the reference script names them: positions 0, 13, 23, 38, 51, 62 counted from the oldest bit."""

    def __init__(self, t, r_f, r_p):
        v = 0
        for value, width in ((1, 2), (0, 4), (254, 12), (t, 12), (r_f, 10), (r_p, 10), ((1 << 30) - 1, 30)):
            v = (v << width) | value
        self.s = v
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        bit = ((s >> 79) ^ (s >> 66) ^ (s >> 56) ^ (s >> 41) ^ (s >> 28) ^ (s >> 17)) & 1
        self.s = ((s << 1) | bit) & ((1 << 80) - 1)
        return bit

    def bits(self, count):
        v = 0
        while count:
            keep, bit = self._step(), self._step()
            if keep:
                v = (v << 1) | bit
                count -= 1
        return v


def spec(t=T, r_f=R_F, r_p=R_P):
    """round constants [(r_f + r_p)][t] and the Cauchy MDS matrix [t][t]"""
    g = _GrainLFSR(t, r_f, r_p)
    constants = []
    for _ in range(r_f + r_p):
        row = []
        while len(row) < t:
            v = g.bits(254)
            if v < R:
                row.append(v)
        constants.append(row)
    while True:
        xs = [g.bits(254) % R for _ in range(t)]
        ys = [g.bits(254) % R for _ in range(t)]
        if len(set(xs + ys)) == 2 * t and all((x + y) % R for x in xs for y in ys):
            return constants, [[pow(x + y, -1, R) for y in ys] for x in xs]


_SPEC = {}


class PoseidonChip:
    """update / squeeze over Context cells"""

    def __init__(self, ctx: Context, t=T, rate=RATE, r_f=R_F, r_p=R_P):
        self.ctx, self.t, self.rate, self.r_f, self.r_p = ctx, t, rate, r_f, r_p
        if (t, r_f, r_p) not in _SPEC:
            _SPEC[(t, r_f, r_p)] = spec(t, r_f, r_p)
        self.constants, self.mds = _SPEC[(t, r_f, r_p)]
        # the initial state is a constant: load it as constant cells (tied to the constants column)
        self.state = [ctx.assign_region_last([("constant", v)], []) for v in [1 << 64] + [0] * (t - 1)]
        self.buf = []

    def update(self, cells):
        self.buf += list(cells)

    def _inner_product_const(self, cells, coeffs):
        """sum_k coeffs[k] * cells[k]: [0, a_0, c_0, acc_1, a_1, c_1, acc_2, ...], a gate on every third row"""
        ctx = self.ctx
        items, gates, acc = [("constant", 0)], [], 0
        for k, (cell, c) in enumerate(zip(cells, coeffs)):
            acc = (acc + ctx.cells[cell] * c) % R
            items += [("existing", cell), ("constant", c), ("witness", acc)]
            gates.append(3 * k)
        return ctx.assign_region_last(items, gates)

    def _permute(self):
        ctx, t = self.ctx, self.t
        half = self.r_f // 2
        for rnd in range(self.r_f + self.r_p):
            s = [ctx.add_constant(cell, c) for cell, c in zip(self.state, self.constants[rnd])]
            lanes = range(t) if rnd < half or rnd >= half + self.r_p else range(1)
            for i in lanes:
                x2 = ctx.mul(s[i], s[i])
                x4 = ctx.mul(x2, x2)
                s[i] = ctx.mul(x4, s[i])
            self.state = [self._inner_product_const(s, row) for row in self.mds]

    def squeeze(self):
        ctx = self.ctx
        chunks = [self.buf[i : i + self.rate] for i in range(0, len(self.buf), self.rate)]
        if len(self.buf) % self.rate == 0:
            chunks.append([])
        self.buf = []
        for chunk in chunks:
            for i, cell in enumerate(chunk):
                self.state[1 + i] = ctx.add(self.state[1 + i], cell)
            if len(chunk) < self.rate:
                self.state[1 + len(chunk)] = ctx.add_constant(self.state[1 + len(chunk)], 1)
            self._permute()
        return self.state[1]


def hash_two_closure(cs: FlexGateCS, x: int, y: int) -> Assignment:
    """reference examples/poseidon.rs:15-36; public inputs [x, y, hash].  Under the Range builder (LOOKUP_BITS set) the caller adds the
    table with flex.load_lookup_table."""
    asg = Assignment(cs)
    ctx = Context(asg)
    xc, yc = ctx.load_witness(x), ctx.load_witness(y)
    chip = PoseidonChip(ctx)
    chip.update([xc, yc])
    out = chip.squeeze()
    ctx.finish([xc, yc, out])
    return asg
