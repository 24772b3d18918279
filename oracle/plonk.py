"""CPU oracle for the quotient step of create_proof on the reference's StandardPlonk circuit.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: restated from the circuit definition in
the reference (src/circuits/standard_plonk.rs:12-112) and from memory of halo2_proofs v2023_02_02
(plonk/permutation/{keygen,prover}.rs, plonk/evaluation.rs `evaluate_h`, plonk/vanishing/prover.rs); the
crate itself is not available here.  What pins it instead is the PLONK identity itself: for a satisfying
witness the combined numerator is divisible by X^n - 1, and the quotient identity holds at a random point
(`check_quotient_identity`), which no wrong restatement of gates / permutation / blinding can satisfy.

Conventions restated:
  columns: advice a, b, c (equality enabled in that order), fixed q_a, q_b, q_c, q_ab, constant
  gate: q_a*a + q_b*b + q_c*c + q_ab*a*b + constant = 0 on every row
  blinding_factors = 5, usable rows u = n - 6, last_rotation = -6, permutation chunk length = degree - 2 = 1
  advice rows u .. n-1 (six rows: `unusable_rows_start = n - (blinding_factors + 1)`) and rows u+1 .. n-1 of every
  permutation product (five rows) hold the prover's random blinding values
  copy constraints: permutation/keygen.rs `Assembly::copy` (cycles merged smaller-into-larger, then the two
  mapping entries swapped), fed by the reference's four `copy_advice` calls in their order
  (src/circuits/standard_plonk.rs:91-92,100-101): constrain_equal(new cell, x's cell (a, 0))
  identity permutation of column j at row i: DELTA^j * omega^i, DELTA = 7^(2^28)
  terms of h(X), combined by Horner in y in this order: gate; l_0 (1 - z_0); l_last (z_2^2 - z_2);
  l_0 (z_m - z_{m-1}(omega^last X)) for m = 1, 2; for every column m:
  l_active (z_m(omega X) (v_m + beta sigma_m + gamma) - z_m(X) (v_m + beta DELTA^m X + gamma))
"""
from __future__ import annotations

from . import bn254 as o

R = o.R
FR_DELTA = pow(o.FR_GENERATOR, 1 << o.FR_S, R)
BLINDING_FACTORS = 5
CS_DEGREE = 3


class Assembly:
    """permutation/keygen.rs Assembly: mapping[col][row] = the next cell of the cycle; sigma_col(omega^row) =
    DELTA^col' omega^row' for (col', row') = mapping[col][row]."""

    def __init__(self, n_columns: int, n: int):
        self.mapping = [[(c, r) for r in range(n)] for c in range(n_columns)]
        self.aux = [[(c, r) for r in range(n)] for c in range(n_columns)]
        self.sizes = [[1] * n for _ in range(n_columns)]

    def copy(self, left, right):
        (lc, lr), (rc, rr) = left, right
        left_cycle, right_cycle = self.aux[lc][lr], self.aux[rc][rr]
        if left_cycle == right_cycle:
            return
        if self.sizes[left_cycle[0]][left_cycle[1]] < self.sizes[right_cycle[0]][right_cycle[1]]:
            left_cycle, right_cycle = right_cycle, left_cycle
        self.sizes[left_cycle[0]][left_cycle[1]] += self.sizes[right_cycle[0]][right_cycle[1]]
        i = right_cycle
        while True:
            self.aux[i[0]][i[1]] = left_cycle
            i = self.mapping[i[0]][i[1]]
            if i == right_cycle:
                break
        self.mapping[lc][lr], self.mapping[rc][rr] = self.mapping[rc][rr], self.mapping[lc][lr]


# the reference's synthesize(): x.copy_advice(.., a, 1), (.., b, 1), (.., a, 2), (.., b, 2) — each is
# constrain_equal(newly assigned cell, x's cell) with x assigned at (a, 0)
STANDARD_PLONK_COPIES = [((0, 1), (0, 0)), ((1, 1), (0, 0)), ((0, 2), (0, 0)), ((1, 2), (0, 0))]


class StandardPlonkInstance:
    """the circuit at 2^k rows with witness x (rows 0..2 as the reference assigns them) and seeded blinding."""

    def __init__(self, k: int, x: int, seed: int = 1):
        self.k, self.n = k, 1 << k
        n = self.n
        assert n >= 16
        self.u = n - (BLINDING_FACTORS + 1)  # index of the l_last row
        self.dom = o.Domain(k, CS_DEGREE)
        self.seed = seed
        # seeded stand-ins for the prover's rng (the reference passes OsRng): stream seed+1 = advice blinding
        # (column-major, six rows each), seed+2 = permutation-product blinding (set-major, five rows each),
        # seed+3 = the vanishing argument's random polynomial
        nb = BLINDING_FACTORS + 1
        rnd = o.unpack(o.random_field_limbs(3 * nb, seed + 1), R)
        # advice (reference src/circuits/standard_plonk.rs:83-108)
        a, b, c = [0] * n, [0] * n, [0] * n
        a[0] = x
        a[1], b[1], c[1] = x, x, x * x % R
        a[2], b[2], c[2] = x, x, (x * x + 72) % R
        for j, col in enumerate((a, b, c)):  # rows unusable_rows_start .. n-1
            for t in range(nb):
                col[self.u + t] = rnd[j * nb + t]
        self.advice = [a, b, c]
        q_a, q_b, q_c, q_ab, const = ([0] * n for _ in range(5))
        q_c[1], q_ab[1] = R - 1, 1
        q_c[2], q_ab[2], const[2] = R - 1, 1, 72
        self.fixed = [q_a, q_b, q_c, q_ab, const]
        # copy constraints: a0 = a1 = b1 = a2 = b2, merged as Assembly::copy merges them
        w = self.dom.omega
        self.omega_pows = [1] * n
        for i in range(1, n):
            self.omega_pows[i] = self.omega_pows[i - 1] * w % R
        dpow = [pow(FR_DELTA, j, R) for j in range(3)]
        ident = lambda j, i: dpow[j] * self.omega_pows[i] % R
        sigma = [[ident(j, i) for i in range(n)] for j in range(3)]
        asm = Assembly(3, 8)  # only the first rows take part
        for left, right in STANDARD_PLONK_COPIES:
            asm.copy(left, right)
        self.copy_mapping = {(c, r): asm.mapping[c][r] for c in range(3) for r in range(8) if asm.mapping[c][r] != (c, r)}
        for (c_, r_), nxt in self.copy_mapping.items():
            sigma[c_][r_] = ident(*nxt)
        self.sigma = sigma
        # Lagrange helpers
        self.l0 = [1] + [0] * (n - 1)
        self.l_last = [0] * n
        self.l_last[self.u] = 1
        self.l_active = [1 if i < self.u else 0 for i in range(n)]

    def permutation_products(self, beta: int, gamma: int):
        n, u = self.n, self.u
        zs = []
        start = 1
        zblind = iter(o.unpack(o.random_field_limbs(3 * BLINDING_FACTORS, self.seed + 2), R))
        for m in range(3):
            z = [0] * n
            z[0] = start
            for i in range(u):
                v = self.advice[m][i]
                num = (v + beta * pow(FR_DELTA, m, R) * self.omega_pows[i] + gamma) % R
                den = (v + beta * self.sigma[m][i] + gamma) % R
                z[i + 1] = z[i] * num % R * pow(den, -1, R) % R
            for r in range(u + 1, n):
                z[r] = next(zblind)
            start = z[u]
            zs.append(z)
        return zs

    # ---- extended-domain (coset) forms ----
    def to_extended(self, lagrange_vals):
        return self.dom.coeff_to_extended(self.dom.lagrange_to_coeff(lagrange_vals))

    def evaluate_h(self, zs, beta: int, gamma: int, y: int):
        """numerator of h on the extended coset, NOT yet divided by X^n - 1; restates evaluation.rs evaluate_h."""
        d = self.dom
        size = 1 << d.extended_k
        rot = size // self.n
        adv = [self.to_extended(c) for c in self.advice]
        fix = [self.to_extended(c) for c in self.fixed]
        sig = [self.to_extended(c) for c in self.sigma]
        zc = [self.to_extended(z) for z in zs]
        l0, ll, la = self.to_extended(self.l0), self.to_extended(self.l_last), self.to_extended(self.l_active)
        out = [0] * size
        for idx in range(size):
            X = d.g_coset * pow(d.extended_omega, idx, R) % R
            r_next = (idx + rot) % size
            r_last = (idx - (BLINDING_FACTORS + 1) * rot) % size
            a, b, c = (adv[j][idx] for j in range(3))
            v = (fix[0][idx] * a + fix[1][idx] * b + fix[2][idx] * c + fix[3][idx] * a * b + fix[4][idx]) % R
            v = (v * y + (1 - zc[0][idx]) * l0[idx]) % R
            v = (v * y + (zc[2][idx] * zc[2][idx] - zc[2][idx]) * ll[idx]) % R
            for m in (1, 2):
                v = (v * y + (zc[m][idx] - zc[m - 1][r_last]) * l0[idx]) % R
            cur = beta * X % R
            for m in range(3):
                left = zc[m][r_next] * (adv[m][idx] + beta * sig[m][idx] + gamma) % R
                right = zc[m][idx] * (adv[m][idx] + cur + gamma) % R
                cur = cur * FR_DELTA % R
                v = (v * y + (left - right) * la[idx]) % R
            out[idx] = v
        return out

    def divide_by_vanishing(self, h_ext):
        """pointwise division by X^n - 1 on the coset (its values repeat with period extended_len / n)."""
        d = self.dom
        size = 1 << d.extended_k
        rot = size // self.n
        tinv = [pow((pow(d.g_coset * pow(d.extended_omega, i, R) % R, self.n, R) - 1) % R, -1, R) for i in range(rot)]
        return [h_ext[i] * tinv[i % rot] % R for i in range(size)]


def check_quotient_identity(inst: StandardPlonkInstance, zs, h_coeffs, beta, gamma, y, x) -> bool:
    """the verifier's equation at the point x: combined gate/permutation expression == h(x) (x^n - 1),
    evaluated from the polynomials' coefficient forms (so it is independent of the coset evaluation above)."""
    d, n = inst.dom, inst.n
    ev = lambda lagr, pt: o.eval_polynomial(d.lagrange_to_coeff(lagr), pt)
    w = d.omega
    x_next = x * w % R
    x_last = x * pow(w, -(BLINDING_FACTORS + 1) % n, R) % R
    a, b, c = (ev(col, x) for col in inst.advice)
    f = [ev(col, x) for col in inst.fixed]
    s = [ev(col, x) for col in inst.sigma]
    z = [ev(zz, x) for zz in zs]
    z_next = [ev(zz, x_next) for zz in zs]
    z_last = [ev(zz, x_last) for zz in zs]
    l0, ll, la = ev(inst.l0, x), ev(inst.l_last, x), ev(inst.l_active, x)
    adv = [a, b, c]
    v = (f[0] * a + f[1] * b + f[2] * c + f[3] * a * b + f[4]) % R
    v = (v * y + (1 - z[0]) * l0) % R
    v = (v * y + (z[2] * z[2] - z[2]) * ll) % R
    for m in (1, 2):
        v = (v * y + (z[m] - z_last[m - 1]) * l0) % R
    cur = beta * x % R
    for m in range(3):
        left = z_next[m] * (adv[m] + beta * s[m] + gamma) % R
        right = z[m] * (adv[m] + cur + gamma) % R
        cur = cur * FR_DELTA % R
        v = (v * y + (left - right) * la) % R
    hx = o.eval_polynomial(h_coeffs, x)
    return v == hx * (pow(x, n, R) - 1) % R
