"""Large-size form of the oracle prover: oracle/flex.py's keygen and `prove`, statement for statement, over whole vectors
(oracle/vec.py: C loops of oracle/h2ref.c) instead of one Python integer at a time.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED exactly as oracle/flex.py is (same restated
create_proof order, same stand-ins for rng and vk.transcript_repr).  Why it exists: the north-star target is a
2^20-row proof (reference call sites examples/standard_plonk.rs:41-50, src/scaffold.rs:322-331) and the Python-integer
oracle needs hours there; this one needs minutes, so `tests/golden/make_big_golden.py` can commit golden proof BYTES at
k = 16 / 20 / 22 for the device provers to reproduce.  What ties it to the slow oracle: it reuses that module's
constraint systems, gate lambdas, `_permutation_terms`, `_lookup_terms`, `construct_intermediate_sets`,
`lagrange_interpolate`, `permute_expression_pair` and transcript unchanged (the vector class overloads + - * %), and
tests/test_oracle_fast.py asserts byte-identical proofs for every shape at the sizes the slow one finishes in seconds.
Commitments stay f(s) * G (inner product with s^i or L_i(s), then one scalar multiplication): no MSM is involved.
"""
from __future__ import annotations

import numpy as np

from . import bn254 as o
from . import cref
from . import flex as FX
from . import formats as fmt
from . import lookup as L
from . import plonk as P
from .prover import construct_intermediate_sets, evaluate_vanishing_polynomial, lagrange_interpolate, rotate_omega
from .vec import FV, FastDomain

R = o.R
ADVICE, FIXED, INSTANCE = FX.ADVICE, FX.FIXED, FX.INSTANCE


def ints_to_fv(vals) -> FV:
    """canonical ints -> Montgomery FV (bulk: bytes -> limbs -> one C to_mont pass)"""
    raw = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4)
    return FV(cref.field_op(1, 6, raw))


def fv_to_ints(v: FV):
    can = cref.field_op(1, 5, v.a)
    b = can.tobytes()
    return [int.from_bytes(b[i : i + 32], "little") for i in range(0, len(b), 32)]


class Keys:
    """flex.Keys on vectors: same attributes where `prove` / `verify` need them"""

    def __init__(self, cs, k, s, asg_fixed, copies):
        self.cs, self.k, self.n, self.s = cs, k, 1 << k, s
        FX.check_rows_available(cs, k, asg_fixed, copies)
        n = self.n
        self.dom = FastDomain(k, cs.degree)
        self.u = n - (cs.blinding_factors + 1)
        self.pw = FV.powers(s, n)
        self.lag = self.pw.ntt(self.dom.omega_inv, k) * pow(n, -1, R)  # L_i(s) = (1/n) sum_j s^j omega^(-ij)
        self.fixed = [FV.from_sparse(n, cells) for cells in asg_fixed]
        w = self.dom.omega
        m = len(cs.perm_columns)
        self.dpow = [pow(P.FR_DELTA, j, R) for j in range(m)]
        self.ident = [FV.powers(w, n, start=self.dpow[j]) for j in range(m)]  # DELTA^j omega^i
        self.sigma = [c.copy() for c in self.ident]
        for (j, i), (tj, ti) in FX._assembly(cs, copies).items():
            self.sigma[j].set(i, self.dpow[tj] * pow(w, ti, R) % R)
        self.l0 = FV.from_sparse(n, {0: 1})
        self.l_last = FV.from_sparse(n, {self.u: 1})
        self.l_active = FV.full(n, 1)
        self.l_active.a[self.u :] = 0
        d = self.dom
        self.fixed_polys = [d.lagrange_to_coeff(c) for c in self.fixed]
        self.sigma_polys = [d.lagrange_to_coeff(c) for c in self.sigma]
        self.fixed_commitments = [self.commit_lagrange(c) for c in self.fixed]
        self.permutation_commitments = [self.commit_lagrange(c) for c in self.sigma]
        self.transcript_repr = FX._vk_transcript_repr(k, cs.degree, self.fixed_commitments + self.permutation_commitments)

    def commit_lagrange(self, evals: FV):
        return o.g1_mul(evals.dot(self.lag), o.G1_GEN)

    def commit(self, coeffs: FV):
        return o.g1_mul(coeffs.dot(self.pw), o.G1_GEN)

    def vk_bytes(self):
        import struct

        out = bytearray(struct.pack("<II", self.k, self.cs.degree))
        for c in self.fixed_commitments + self.permutation_commitments:
            out += fmt.g1_to_bytes(c)
        return bytes(out)


def prove(keys: Keys, asg, seed: int) -> dict:
    """flex.prove, same statement order; see there for the restated create_proof sequence"""
    cs, n, u, dom = keys.cs, keys.n, keys.u, keys.dom
    bf = cs.blinding_factors
    tr = fmt.Blake2bTranscript()
    tr.common_scalar(keys.transcript_repr)
    instance_cols = []
    for vals in asg.instance:
        for v in vals:
            tr.common_scalar(v)
        instance_cols.append(FV.from_sparse(n, dict(enumerate(vals))))
    blind = iter(FX._rand(cs.n_advice * (bf + 1), seed + 1))
    advice = []
    for cells in asg.advice:
        assert all(r < u for r in cells), "assignment reaches into the blinding rows"
        col = FV.from_sparse(n, cells)
        for r in range(u, n):
            col.set(r, next(blind))
        advice.append(col)
    for c in advice:
        tr.write_point(keys.commit_lagrange(c))
    theta = tr.squeeze_challenge()
    columns = {ADVICE: advice, FIXED: keys.fixed, INSTANCE: instance_cols}
    col_of = lambda kc: columns[kc[0]][kc[1]]

    def input_of(factors):
        out = None
        for kc in factors:
            out = col_of(kc) if out is None else out * col_of(kc)
        return out

    lk_blind = iter(FX._rand(2 * (bf + 1) * max(len(cs.lookups), 1), seed + 4))
    permuted = []
    for inp, tab in cs.lookups:
        a_in, t_in = input_of(inp), col_of(tab)
        ap, sp = L.permute_expression_pair(fv_to_ints(a_in), fv_to_ints(t_in), u, [next(lk_blind) for _ in range(bf + 1)],
                                           [next(lk_blind) for _ in range(bf + 1)])
        ap, sp = ints_to_fv(ap), ints_to_fv(sp)
        permuted.append((ap, sp))
        tr.write_point(keys.commit_lagrange(ap))
        tr.write_point(keys.commit_lagrange(sp))
    beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge()
    m = len(cs.perm_columns)
    zblind = iter(FX._rand(-(-m // cs.chunk) * bf, seed + 2))
    zs, start = [], 1
    for s0 in range(0, m, cs.chunk):
        num = den = 1
        for j in range(s0, min(m, s0 + cs.chunk)):
            v = col_of(cs.perm_columns[j])
            num = num * ((v + beta * keys.ident[j] + gamma) % R) % R
            den = den * ((v + beta * keys.sigma[j] + gamma) % R) % R
        z = (num * den.batch_inv()).running_product(start, u + 1).padded(n)  # z[i+1] = z[i] num_i / den_i, i < u
        for r in range(u + 1, n):
            z.set(r, next(zblind))
        start = z.get(u)
        zs.append(z)
    for z in zs:
        tr.write_point(keys.commit_lagrange(z))
    lz_blind = iter(FX._rand(bf * max(len(cs.lookups), 1), seed + 5))
    lzs = []
    for (inp, tab), (ap, sp) in zip(cs.lookups, permuted):
        num = (input_of(inp) + beta) * (col_of(tab) + gamma)
        den = (ap + beta) * (sp + gamma)
        lz = (num * den.batch_inv()).running_product(1, u + 1).padded(n)
        for r in range(u + 1, n):
            lz.set(r, next(lz_blind))
        lzs.append(lz)
        tr.write_point(keys.commit_lagrange(lz))
    random_poly = FV(o.random_field_limbs(n, seed + 3))
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
    q = lambda kind, col, r: cosets[kind][col].roll(r * rot) if r else cosets[kind][col]
    v = 0
    for gate in cs.gates:
        v = (v * y + gate(q)) % R
    X = FV.powers(dom.extended_omega, size, start=dom.g_coset)
    v = FX._permutation_terms(cs, v, y, beta, gamma, lambda kc: cosets[kc[0]][kc[1]], sig_c, z_c, [z.roll(rot) for z in z_c],
                              [z.roll(-(bf + 1) * rot) for z in z_c], l0, ll, lact, X)
    del X
    for (inp, tab), (ap, sp, lz) in zip(cs.lookups, lk_c):
        a_val = None
        for kc in inp:
            a_val = cosets[kc[0]][kc[1]] if a_val is None else a_val * cosets[kc[0]][kc[1]] % R
        v = FX._lookup_terms(v, y, beta, gamma, a_val, cosets[tab[0]][tab[1]], ap, ap.roll(-rot), sp, lz, lz.roll(rot), l0, ll, lact)
    tinv = [pow((pow(dom.g_coset * pow(dom.extended_omega, i, R) % R, n, R) - 1) % R, -1, R) for i in range(rot)]
    tinv_v = FV(np.ascontiguousarray(np.tile(o.pack(tinv, R), (n, 1))))
    h_ext = v * tinv_v
    del adv_c, fix_c, ins_c, sig_c, z_c, lk_c, l0, ll, lact, cosets, v, tinv_v
    h_coeffs = dom.extended_to_coeff(h_ext)
    del h_ext
    pieces = [FV(np.ascontiguousarray(h_coeffs.a[i * n : (i + 1) * n])) for i in range(dom.quotient_poly_degree)]
    for p in pieces:
        tr.write_point(keys.commit(p))
    x = tr.squeeze_challenge()
    xn = pow(x, n, R)
    # ---- evaluations -----------------------------------------------------------------------------------------------
    ev = lambda poly, pt: poly.eval(pt)
    advice_polys = [dom.lagrange_to_coeff(c) for c in advice]
    z_polys = [dom.lagrange_to_coeff(z) for z in zs]
    lk_polys = [(dom.lagrange_to_coeff(ap), dom.lagrange_to_coeff(sp), dom.lagrange_to_coeff(lz)) for (ap, sp), lz in zip(permuted, lzs)]
    rx = lambda r: rotate_omega(dom, x, r)
    for c, r in cs.advice_queries:
        tr.write_scalar(ev(advice_polys[c], rx(r)))
    for c, r in cs.fixed_queries:
        tr.write_scalar(ev(keys.fixed_polys[c], rx(r)))
    h_poly = FV.zeros(n)
    for piece in reversed(pieces):
        h_poly = (h_poly * xn + piece) % R
    tr.write_scalar(ev(random_poly, x))
    for sp_ in keys.sigma_polys:
        tr.write_scalar(ev(sp_, x))
    x_next, x_last, x_inv = rx(1), rx(-(bf + 1)), rx(-1)
    for i, zp in enumerate(z_polys):
        tr.write_scalar(ev(zp, x))
        tr.write_scalar(ev(zp, x_next))
        if i + 1 < len(z_polys):
            tr.write_scalar(ev(zp, x_last))
    for app, spp, lzp in lk_polys:
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
    for li, (app, spp, lzp) in enumerate(lk_polys):
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
    sh = _shplonk_prove(keys, tr, polys, queries, n)
    out = {"proof": bytes(tr.proof), "theta": theta, "beta": beta, "gamma": gamma, "y": y, "x": x, "zs": zs, "permuted": permuted, "lzs": lzs,
           "h_coeffs": h_coeffs, "advice": advice}
    out.update(sh)
    return out


def _sub_head(poly: FV, low):
    """poly - (low-degree polynomial given by its coefficient list)"""
    out = poly.copy()
    for i, c in enumerate(low):
        out.set(i, (out.get(i) - c) % R)
    return out


def _shplonk_prove(keys, tr, polys, queries, n):
    """flex._shplonk_prove on vectors"""
    y_sh = tr.squeeze_challenge()
    rotation_sets, super_points = construct_intermediate_sets(queries)
    v = tr.squeeze_challenge()
    quotients = []
    for pts, comms in rotation_sets:
        nx, yp = FV.zeros(n), 1
        for key, evals in comms:
            num = _sub_head(polys[key], lagrange_interpolate(pts, evals))
            nx = (nx + yp * num) % R
            yp = yp * y_sh % R
        for root in pts:  # div_by_vanishing: one kate_division per point (the vector keeps its length, top coefficients zero)
            nx = nx.kate_division(root)
        quotients.append(nx)
    h_x, vp = FV.zeros(n), 1
    for qx in quotients:
        h_x = (h_x + vp * qx) % R
        vp = vp * v % R
    tr.write_point(keys.commit(h_x))
    u = tr.squeeze_challenge()
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x, z_diffs, vp = FV.zeros(n), [], 1
    for pts, comms in rotation_sets:
        z_i = evaluate_vanishing_polynomial([p for p in super_points if p not in pts], u)
        z_diffs.append(z_i)
        inner, yp = FV.zeros(n), 1
        for key, evals in comms:
            lin = _sub_head(polys[key], [o.eval_polynomial(lagrange_interpolate(pts, evals), u)])
            inner = (inner + yp * lin) % R
            yp = yp * y_sh % R
        l_x = (l_x + vp * z_i % R * inner) % R
        vp = vp * v % R
    l_x = (l_x - zt_eval * h_x) % R
    assert l_x.eval(u) == 0
    z0inv = pow(z_diffs[0], -1, R)
    h2_x = l_x.kate_division(u) * z0inv
    tr.write_point(keys.commit(h2_x))
    return {"shplonk_y": y_sh, "v": v, "u": u}
