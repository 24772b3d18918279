"""CPU oracle for the lookup argument and the range-check constraint system the reference's
RangeWithInstanceCircuitBuilder produces (src/scaffold.rs:434-485; table of 2^LOOKUP_BITS values, :44-48, :462).

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: restated from memory of halo2_proofs v2023_02_02
(plonk/lookup/prover.rs `commit_permuted` / `permute_expression_pair` / `commit_product`, plonk/evaluation.rs, the lookup
part of `evaluate_h`) and of halo2-base's FlexGate / RangeConfig (one vertical gate q (a + a(wX) a(w^2 X) - a(w^3 X)),
one lookup-enabled advice column whose every cell must lie in the fixed table); neither crate is available here.  What
pins it: for a satisfying witness the combined numerator is divisible by X^n - 1 and the quotient identity holds at a
random point (`check_quotient_identity`) — no mis-stated permuted column, product or term order survives that.

Shape ("range shape"): advice a (gate), la (lookup advice); fixed q (selector), t (table); optional extra equality-
enabled columns (the builder's instance and constants columns) that only take part in the permutation argument;
constraint-system degree 4 (the lookup product term), so the permutation chunk length is 2 and the extended domain is 4n.
Single-expression lookups only (one input column, one table column): the theta-compression is the identity.
"""
from __future__ import annotations

from . import bn254 as o
from .plonk import BLINDING_FACTORS, FR_DELTA

R = o.R
CS_DEGREE = 4


def permute_expression_pair(inputs, table, usable_rows: int, blind_in, blind_tab):
    """lookup/prover.rs permute_expression_pair: (A', S') of n entries.  A' = the usable inputs sorted (Fr's Ord =
    canonical integer order); S' holds the input value wherever A' starts a new run, and the table values not consumed
    that way — ascending, each as often as it is left over — on the repeated rows, LAST repeated row first.
    blind_*: the blinding_factors + 1 random values appended to each."""
    a_sorted = sorted(inputs[:usable_rows])
    leftover = {}
    for v in table[:usable_rows]:
        leftover[v] = leftover.get(v, 0) + 1
    s_perm = [0] * usable_rows
    repeated = []
    for row, v in enumerate(a_sorted):
        if row == 0 or v != a_sorted[row - 1]:
            s_perm[row] = v
            if leftover.get(v, 0) <= 0:
                raise ValueError("input value not in the table (ConstraintSystemFailure)")
            leftover[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(leftover):
        for _ in range(leftover[v]):
            s_perm[repeated.pop()] = v
    assert not repeated
    return a_sorted + list(blind_in), s_perm + list(blind_tab)


def lookup_product(inputs, table, a_perm, s_perm, beta: int, gamma: int, usable_rows: int, blind):
    """lookup/prover.rs commit_product: z[0] = 1, z[i+1] = z[i] (a_i + beta)(t_i + gamma) / ((a'_i + beta)(s'_i + gamma)),
    i < usable_rows; then blinding_factors random values."""
    z = [1]
    for i in range(usable_rows):
        num = (inputs[i] + beta) * (table[i] + gamma) % R
        den = (a_perm[i] + beta) * (s_perm[i] + gamma) % R
        z.append(z[-1] * num % R * pow(den, -1, R) % R)
    return z + list(blind)


class RangeInstance:
    """a satisfying assignment of the range shape at 2^k rows.

    Gate rows (q = 1 on every fourth row up to `gates`): a_r + a_{r+1} a_{r+2} = a_{r+3}.  Lookup column: values below
    2^lookup_bits (some copied from gate cells that are themselves small, so the permutation argument is non-trivial).
    extra_cols: further equality-enabled columns (dense random, copy-constrained among themselves)."""

    def __init__(self, k: int, lookup_bits: int, seed: int = 1, gates: int = None, extra_cols: int = 0):
        self.k, self.n = k, 1 << k
        n = self.n
        self.u = n - (BLINDING_FACTORS + 1)
        u = self.u
        assert (1 << lookup_bits) <= u
        self.dom = o.Domain(k, CS_DEGREE)
        rnd = iter(o.unpack(o.random_field_limbs(4 * n + 64, seed), R))
        small = lambda: next(rnd) % (1 << lookup_bits)
        gates = (u - 4) // 4 if gates is None else gates
        a, q = [0] * n, [0] * n
        for g in range(gates):
            r = 4 * g
            a[r], a[r + 1], a[r + 2] = small(), small(), next(rnd)
            a[r + 3] = (a[r] + a[r + 1] * a[r + 2]) % R
            q[r] = 1
        la = [small() for _ in range(u)] + [0] * (n - u)
        # copy constraints between the two columns: la[i] := a[4i] (a small gate operand), a few dozen of them
        copies = [((1, i), (0, 4 * i)) for i in range(min(gates, 40))]
        for (c1, r1), (c0, r0) in copies:
            la[r1] = a[r0]
        table = list(range(1 << lookup_bits)) + [0] * (n - (1 << lookup_bits))
        extras = []
        for e in range(extra_cols):
            col = [next(rnd) for _ in range(u)] + [0] * (n - u)
            extras.append(col)
        if extra_cols >= 2:  # tie two cells of the extra columns together
            extras[1][5] = extras[0][7]
            copies.append(((3, 5), (2, 7)))
        for col in [a, la] + extras:  # blinding rows
            for r in range(u, n):
                col[r] = next(rnd)
        self.a, self.la, self.q, self.table, self.extras = a, la, q, table, extras
        self.perm_cols = [a, la] + extras
        m = len(self.perm_cols)
        w = self.dom.omega
        self.omega_pows = [1] * n
        for i in range(1, n):
            self.omega_pows[i] = self.omega_pows[i - 1] * w % R
        self.dpow = [pow(FR_DELTA, j, R) for j in range(m)]
        ident = lambda j, i: self.dpow[j] * self.omega_pows[i] % R
        sigma = [[ident(j, i) for i in range(n)] for j in range(m)]
        # every copy is a 2-cycle here (each cell takes part in at most one)
        for (c1, r1), (c0, r0) in copies:
            sigma[c1][r1], sigma[c0][r0] = ident(c0, r0), ident(c1, r1)
        self.sigma = sigma
        self.copies = copies
        self._rnd = rnd
        self.l0 = [1] + [0] * (n - 1)
        self.l_last = [0] * n
        self.l_last[u] = 1
        self.l_active = [1 if i < u else 0 for i in range(n)]

    def blind(self, count):
        return [next(self._rnd) for _ in range(count)]

    # ---- prover columns ----
    def permuted(self):
        return permute_expression_pair(self.la, self.table, self.u, self.blind(BLINDING_FACTORS + 1), self.blind(BLINDING_FACTORS + 1))

    def permutation_products(self, beta, gamma, chunk=CS_DEGREE - 2):
        n, u = self.n, self.u
        m = len(self.perm_cols)
        zs, start = [], 1
        for s0 in range(0, m, chunk):
            z = [0] * n
            z[0] = start
            for i in range(u):
                num = den = 1
                for j in range(s0, min(m, s0 + chunk)):
                    v = self.perm_cols[j][i]
                    num = num * ((v + beta * self.dpow[j] * self.omega_pows[i] + gamma) % R) % R
                    den = den * ((v + beta * self.sigma[j][i] + gamma) % R) % R
                z[i + 1] = z[i] * num % R * pow(den, -1, R) % R
            for r in range(u + 1, n):
                z[r] = next(self._rnd)
            start = z[u]
            zs.append(z)
        return zs

    def to_extended(self, lagrange_vals):
        return self.dom.coeff_to_extended(self.dom.lagrange_to_coeff(lagrange_vals))

    def evaluate_h(self, zs, a_perm, s_perm, z_lk, beta, gamma, y, chunk=CS_DEGREE - 2):
        """numerator of h on the extended coset (not yet divided): gate, permutation terms, lookup terms, Horner in y"""
        d = self.dom
        size = 1 << d.extended_k
        rot = size // self.n
        E = self.to_extended
        a, la, q, t = E(self.a), E(self.la), E(self.q), E(self.table)
        cols = [E(c) for c in self.perm_cols]
        sig = [E(c) for c in self.sigma]
        zc = [E(z) for z in zs]
        ap, sp, zl = E(a_perm), E(s_perm), E(z_lk)
        l0, ll, lact = E(self.l0), E(self.l_last), E(self.l_active)
        m = len(cols)
        out = [0] * size
        for idx in range(size):
            X = d.g_coset * pow(d.extended_omega, idx, R) % R
            rn = lambda r: (idx + r * rot) % size
            v = q[idx] * (a[idx] + a[rn(1)] * a[rn(2)] - a[rn(3)]) % R
            v = (v * y + (1 - zc[0][idx]) * l0[idx]) % R
            v = (v * y + (zc[-1][idx] * zc[-1][idx] - zc[-1][idx]) * ll[idx]) % R
            for s in range(1, len(zc)):
                v = (v * y + (zc[s][idx] - zc[s - 1][rn(-(BLINDING_FACTORS + 1))]) * l0[idx]) % R
            cur = beta * X % R
            for s, s0 in enumerate(range(0, m, chunk)):
                left, right = zc[s][rn(1)], zc[s][idx]
                for j in range(s0, min(m, s0 + chunk)):
                    left = left * (cols[j][idx] + beta * sig[j][idx] + gamma) % R
                    right = right * (cols[j][idx] + cur + gamma) % R
                    cur = cur * FR_DELTA % R
                v = (v * y + (left - right) * lact[idx]) % R
            # lookup (plonk/evaluation.rs): five terms
            table_value = (la[idx] + beta) * (t[idx] + gamma) % R
            a_minus_s = (ap[idx] - sp[idx]) % R
            v = (v * y + (1 - zl[idx]) * l0[idx]) % R
            v = (v * y + (zl[idx] * zl[idx] - zl[idx]) * ll[idx]) % R
            v = (v * y + (zl[rn(1)] * (ap[idx] + beta) % R * (sp[idx] + gamma) - zl[idx] * table_value) * lact[idx]) % R
            v = (v * y + a_minus_s * l0[idx]) % R
            v = (v * y + a_minus_s * (ap[idx] - ap[rn(-1)]) % R * lact[idx]) % R
            out[idx] = v
        return out

    def divide_by_vanishing(self, h_ext):
        d = self.dom
        size = 1 << d.extended_k
        rot = size // self.n
        tinv = [pow((pow(d.g_coset * pow(d.extended_omega, i, R) % R, self.n, R) - 1) % R, -1, R) for i in range(rot)]
        return [h_ext[i] * tinv[i % rot] % R for i in range(size)]


def check_quotient_identity(inst: RangeInstance, zs, a_perm, s_perm, z_lk, h_coeffs, beta, gamma, y, x, chunk=CS_DEGREE - 2, ev=None) -> bool:
    """the verifier's equation at x from coefficient forms: expression(x) == h(x) (x^n - 1).
    `ev(lagrange_column_name_or_values, point)` may be supplied to evaluate polynomials elsewhere (e.g. on the device)."""
    d, n = inst.dom, inst.n
    if ev is None:
        ev = lambda lagr, pt: o.eval_polynomial(d.lagrange_to_coeff(lagr), pt)
    w = d.omega
    rot = lambda r: x * pow(w, r % n, R) % R
    a0, a1, a2, a3 = (ev(inst.a, rot(r)) for r in range(4))
    v = ev(inst.q, x) * (a0 + a1 * a2 - a3) % R
    cols = [ev(c, x) for c in inst.perm_cols]
    sig = [ev(c, x) for c in inst.sigma]
    z = [ev(zz, x) for zz in zs]
    z_next = [ev(zz, rot(1)) for zz in zs]
    z_last = [ev(zz, rot(-(BLINDING_FACTORS + 1))) for zz in zs]
    l0, ll, lact = ev(inst.l0, x), ev(inst.l_last, x), ev(inst.l_active, x)
    v = (v * y + (1 - z[0]) * l0) % R
    v = (v * y + (z[-1] * z[-1] - z[-1]) * ll) % R
    for s in range(1, len(z)):
        v = (v * y + (z[s] - z_last[s - 1]) * l0) % R
    cur = beta * x % R
    m = len(cols)
    for s, s0 in enumerate(range(0, m, chunk)):
        left, right = z_next[s], z[s]
        for j in range(s0, min(m, s0 + chunk)):
            left = left * (cols[j] + beta * sig[j] + gamma) % R
            right = right * (cols[j] + cur + gamma) % R
            cur = cur * FR_DELTA % R
        v = (v * y + (left - right) * lact) % R
    la, t = ev(inst.la, x), ev(inst.table, x)
    ap, sp, zl = ev(a_perm, x), ev(s_perm, x), ev(z_lk, x)
    ap_prev, zl_next = ev(a_perm, rot(-1)), ev(z_lk, rot(1))
    v = (v * y + (1 - zl) * l0) % R
    v = (v * y + (zl * zl - zl) * ll) % R
    v = (v * y + (zl_next * (ap + beta) % R * (sp + gamma) - zl * (la + beta) % R * (t + gamma)) * lact) % R
    v = (v * y + (ap - sp) * l0) % R
    v = (v * y + (ap - sp) * (ap - ap_prev) % R * lact) % R
    hx = o.eval_polynomial(h_coeffs, x)
    return v % R == hx * (pow(x, n, R) - 1) % R
