"""CPU oracle: the optimal ate pairing on BN254 (alt_bn128), so that `oracle/prover.py::verify_proof` can finish as the
reference's `verify_proof` does — with the pairing check e(L, [s]G2) = e(R, G2) over the SRS's two G2 elements
(reference examples/standard_plonk.rs:53-64: `params.verifier_params()`, `VerifierSHPLONK`, `SingleStrategy`) — instead
of using the toxic-waste scalar.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  Written from the published construction (Fq12 = Fq[w] / (w^12 - 18 w^6 +
82), the sextic twist y^2 = x^3 + 3 / (9 + u), Miller loop over 6t + 2 = 29793968203157093288 with the two Frobenius
line corrections, final exponentiation by (q^12 - 1) / r) in the simplest possible arithmetic: dense polynomials over Fq,
generic extended Euclid for inverses.  Slow (seconds per pairing) and meant to be: it is pinned by bilinearity and
non-degeneracy (tests/test_oracle_prover.py), which no wrong line function, twist or exponent survives.
"""
from __future__ import annotations

from . import bn254 as o

Q = o.Q
R = o.R
ATE_LOOP_COUNT = 29793968203157093288  # 6 t + 2, t = 4965661367192848881
LOG_ATE = 63
MODULUS = [82, 0, 0, 0, 0, 0, -18 % Q, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82


class F12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % Q for x in c]

    @staticmethod
    def one():
        return F12([1] + [0] * 11)

    @staticmethod
    def zero():
        return F12([0] * 12)

    def __eq__(self, other):
        return self.c == other.c

    def __add__(self, other):
        return F12([a + b for a, b in zip(self.c, other.c)])

    def __sub__(self, other):
        return F12([a - b for a, b in zip(self.c, other.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def scale(self, k: int):
        return F12([a * k for a in self.c])

    def __mul__(self, other):
        prod = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(other.c):
                    prod[i + j] += a * b
        for d in range(22, 11, -1):  # w^d = w^(d-12) (18 w^6 - 82)
            t = prod[d] % Q
            if t:
                prod[d - 6] += 18 * t
                prod[d - 12] -= 82 * t
        return F12(prod[:12])

    def inv(self):
        """extended Euclid on polynomials over Fq: self * inv = 1 mod (w^12 - 18 w^6 + 82)"""
        def deg(p):
            d = len(p) - 1
            while d > 0 and p[d] % Q == 0:
                d -= 1
            return d

        def divmod_poly(a, b):
            a = [x % Q for x in a]
            db = deg(b)
            inv_lead = pow(b[db], -1, Q)
            quo = [0] * (max(deg(a) - db, 0) + 1)
            for d in range(deg(a), db - 1, -1):
                if a[d] % Q == 0:
                    continue
                f = a[d] * inv_lead % Q
                quo[d - db] = f
                for i in range(db + 1):
                    a[d - db + i] = (a[d - db + i] - f * b[i]) % Q
            return quo, a

        lm, hm = [1], [0]
        low, high = list(self.c), MODULUS + [1]
        while deg(low) > 0 or low[0] % Q != 0:
            if deg(low) == 0:
                break
            quo, rem = divmod_poly(high, low)
            # new = hm - quo * lm
            nm = list(hm) + [0] * (len(quo) + len(lm))
            for i, a in enumerate(quo):
                for j, b in enumerate(lm):
                    nm[i + j] -= a * b
            nm = [x % Q for x in nm]
            hm, lm = lm, nm
            high, low = low, rem
        c0 = pow(low[0], -1, Q)
        out = [(x * c0) % Q for x in lm] + [0] * 12
        return F12(out[:12])

    def __truediv__(self, other):
        return self * other.inv()

    def __pow__(self, e: int):
        result, base = F12.one(), self
        while e:
            if e & 1:
                result = result * base
            base = base * base
            e >>= 1
        return result


W = F12([0, 1] + [0] * 10)


def cast_g1(p):
    return (F12([p[0]] + [0] * 11), F12([p[1]] + [0] * 11))


def twist(pt):
    """G2 point over Fq2 = Fq[u] / (u^2 + 1) -> the curve y^2 = x^3 + 3 over Fq12 (u = w^6 - 9)"""
    (x0, x1), (y0, y1) = pt
    nx = F12([x0 - 9 * x1] + [0] * 5 + [x1] + [0] * 5)
    ny = F12([y0 - 9 * y1] + [0] * 5 + [y1] + [0] * 5)
    return (nx * (W * W), ny * (W * W * W))


def _double(p):
    x, y = p
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    """the line through p1 and p2 (tangent if equal), evaluated at t"""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q_g2, p_g1) -> F12:
    """without the final exponentiation; the identity in either group gives one"""
    if q_g2 is None or p_g1 is None:
        return F12.one()
    qt, pt = twist(q_g2), cast_g1(p_g1)
    r = qt
    f = F12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * _line(r, r, pt)
        r = _double(r)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * _line(r, qt, pt)
            r = _add(r, qt)
    q1 = (qt[0] ** Q, qt[1] ** Q)
    nq2 = (q1[0] ** Q, -(q1[1] ** Q))
    f = f * _line(r, q1, pt)
    r = _add(r, q1)
    f = f * _line(r, nq2, pt)
    return f


FINAL_EXPONENT = (Q ** 12 - 1) // R


def final_exponentiate(f: F12) -> F12:
    return f ** FINAL_EXPONENT


def pairing(q_g2, p_g1) -> F12:
    return final_exponentiate(miller_loop(q_g2, p_g1))


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 with one final exponentiation; pairs = [(G1 point, G2 point)]"""
    f = F12.one()
    for p, q in pairs:
        f = f * miller_loop(q, p)
    return final_exponentiate(f) == F12.one()
