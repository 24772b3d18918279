"""CPU oracle for create_proof / verify_proof on the reference's StandardPlonk circuit — keygen, the whole
prover transcript and the SHPLONK multi-open argument, in plain Python integers.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: restated from the reference's call sites
(examples/standard_plonk.rs:29-64: setup, keygen_vk, keygen_pk, create_proof, verify_proof; circuit
src/circuits/standard_plonk.rs:26-114) and from memory of halo2_proofs v2023_02_02 (plonk/{keygen,prover,
verifier}.rs, plonk/permutation/*, plonk/vanishing/*, poly/kzg/multiopen/shplonk/{prover,verifier}.rs,
poly/kzg/multiopen/shplonk.rs `construct_intermediate_sets`); the crate is not available here and the reference
holds no proof bytes.  What pins it instead: `verify_proof` below accepts exactly the proofs whose openings are
consistent — it re-derives every challenge from the proof bytes, recomputes the gate / permutation expressions
from the claimed evaluations, and checks the final KZG equation either as the reference's verifier does, with the
pairing e(L, [s]G2) = e(Rt, G2) over the SRS's two G2 elements (oracle/pairing.py; no secret involved), or in its G1
form s * L == Rt with the toxic-waste scalar of the synthetic SRS (cheap; bench and tests generate the SRS from a
seed; the reference uses `ParamsKZG::setup(k, OsRng)`).

Two deliberate stand-ins, both documented where they are used:
  * rng: the reference passes OsRng (examples/standard_plonk.rs:48), so its bytes are not reproducible; here every
    `Scalar::random(rng)` sweep is a seeded SplitMix64 stream (see oracle/plonk.py StandardPlonkInstance);
  * vk.transcript_repr: the crate hashes `format!("{:?}", vk.pinned())` (Rust Debug text, not reproducible
    without the crate); here the same Blake2b-512 / "Halo2-Verify-Key" construction runs over k, the degree and the
    compressed fixed / permutation commitments.

Commitments are computed as f(s) * G (one scalar multiplication) — the definition the MSM over the SRS must
reproduce, by another route.
"""
from __future__ import annotations

import hashlib
import struct

from . import bn254 as o
from . import formats as fmt
from . import plonk as P

R = o.R


# ---- small polynomial helpers (arithmetic.rs: lagrange_interpolate, eval_polynomial, kate_division) ----------
def poly_mul_linear(a, root):  # a(X) * (X - root)
    out = [0] * (len(a) + 1)
    for i, c in enumerate(a):
        out[i + 1] = (out[i + 1] + c) % R
        out[i] = (out[i] - c * root) % R
    return out


def lagrange_interpolate(points, evals):
    """coefficients (low to high) of the polynomial of degree < len(points) through (points[i], evals[i])."""
    n = len(points)
    out = [0] * n
    for j in range(n):
        num = [1]
        den = 1
        for m in range(n):
            if m != j:
                num = poly_mul_linear(num, points[m])
                den = den * (points[j] - points[m]) % R
        scale = evals[j] * pow(den, -1, R) % R
        for i, c in enumerate(num):
            out[i] = (out[i] + c * scale) % R
    return out


def evaluate_vanishing_polynomial(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R
    return acc


def div_by_vanishing(poly, roots):
    for r in roots:
        poly = o.kate_division(poly, r)
    return poly


# ---- keys --------------------------------------------------------------------------------------------------------
class ProvingKey:
    """keygen_vk + keygen_pk for the reference's circuit at 2^k rows over the SRS with secret s."""

    def __init__(self, k: int, s: int):
        self.k, self.n, self.s = k, 1 << k, s
        self.inst = P.StandardPlonkInstance(k, 0)  # fixed columns, sigma, Lagrange helpers do not depend on the witness
        self.dom = self.inst.dom
        self.pw, self.lag = o.srs_scalars(k, s)  # s^i and L_i(s)
        i = self.inst
        self.fixed_polys = [self.dom.lagrange_to_coeff(c) for c in i.fixed]
        self.sigma_polys = [self.dom.lagrange_to_coeff(c) for c in i.sigma]
        self.fixed_commitments = [self.commit_lagrange(c) for c in i.fixed]
        self.permutation_commitments = [self.commit_lagrange(c) for c in i.sigma]
        self.transcript_repr = self._transcript_repr()

    def commit_lagrange(self, evals):
        return o.g1_mul(sum(e * l for e, l in zip(evals, self.lag)) % R, o.G1_GEN)

    def commit(self, coeffs):
        return o.g1_mul(sum(c * p for c, p in zip(coeffs, self.pw)) % R, o.G1_GEN)

    def vk_bytes(self) -> bytes:
        out = bytearray(struct.pack("<II", self.k, P.CS_DEGREE))
        for c in self.fixed_commitments + self.permutation_commitments:
            out += fmt.g1_to_bytes(c)
        return bytes(out)

    def _transcript_repr(self) -> int:
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        s = self.vk_bytes()
        h.update(struct.pack("<Q", len(s)))
        h.update(s)
        return int.from_bytes(h.digest(), "little") % R


def vk_transcript_repr(k: int, fixed_commitments, permutation_commitments) -> int:
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    s = bytearray(struct.pack("<II", k, P.CS_DEGREE))
    for c in list(fixed_commitments) + list(permutation_commitments):
        s += fmt.g1_to_bytes(c)
    h.update(struct.pack("<Q", len(s)))
    h.update(bytes(s))
    return int.from_bytes(h.digest(), "little") % R


class VerifierKey:
    """what verify_proof needs: the verifying key's commitments (affine points) and, for the G1 form of the final
    pairing check, the SRS secret.  `closed_form(k, s)` derives the commitments at ANY size from the circuit
    definition without touching a length-n vector: a fixed column is zero except for its assigned cells, so its
    commitment is sum_cells v L_row(s) G; sigma_j is the identity image DELTA^j omega^i — the evaluations of the
    polynomial DELTA^j X, whose commitment is DELTA^j s G — patched at the cells of the copy cycle."""

    def __init__(self, k: int, s, fixed_commitments, permutation_commitments):
        """s may be None when verify_proof is given the SRS's G2 elements (pairing check)"""
        self.k, self.n, self.s = k, 1 << k, s
        self.dom = o.Domain(k, P.CS_DEGREE)
        self.fixed_commitments = list(fixed_commitments)
        self.permutation_commitments = list(permutation_commitments)
        self.transcript_repr = vk_transcript_repr(k, self.fixed_commitments, self.permutation_commitments)

    @classmethod
    def closed_form(cls, k: int, s: int) -> "VerifierKey":
        n = 1 << k
        w = o.omega_for(k)
        sn1 = (pow(s, n, R) - 1) * pow(n, -1, R) % R

        def lagrange_at_s(row):  # L_row(s) = (s^n - 1) / n * omega^row / (s - omega^row)
            wr = pow(w, row, R)
            return sn1 * wr % R * pow((s - wr) % R, -1, R) % R

        minus1 = R - 1
        fixed_cells = [{}, {}, {1: minus1, 2: minus1}, {1: 1, 2: 1}, {2: 72}]  # q_a, q_b, q_c, q_ab, constant
        fixed = [o.g1_mul(sum(v * lagrange_at_s(r) for r, v in cells.items()) % R, o.G1_GEN) for cells in fixed_cells]
        asm = P.Assembly(3, 8)
        for left, right in P.STANDARD_PLONK_COPIES:
            asm.copy(left, right)
        ident = lambda c, r: pow(P.FR_DELTA, c, R) * pow(w, r, R) % R
        perm = []
        for j in range(3):
            acc = pow(P.FR_DELTA, j, R) * s % R
            for r in range(8):
                t = asm.mapping[j][r]
                if t != (j, r):
                    acc = (acc + (ident(*t) - ident(j, r)) * lagrange_at_s(r)) % R
            perm.append(o.g1_mul(acc, o.G1_GEN))
        return cls(k, s, fixed, perm)


# the queries of the constraint system in the order ConstraintSystem records them (configure(): a, b, c then
# q_a, q_b, q_c, q_ab, constant, all at Rotation::cur — src/circuits/standard_plonk.rs:42-45)
ADVICE_QUERIES = [(0, 0), (1, 0), (2, 0)]
FIXED_QUERIES = [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0)]


def rotate_omega(dom, x, rot):
    return x * pow(dom.omega, rot % dom.n, R) % R


def construct_intermediate_sets(queries):
    """shplonk.rs: queries = [(commitment key, point, eval)] in order -> (rotation sets, sorted super point set).
    A rotation set is (sorted points, [(key, evals at those points)])."""
    def get_eval(key, pt):
        return next(ev for k_, p_, ev in queries if k_ == key and p_ == pt)

    super_points = sorted({p for _, p, _ in queries})
    commitment_sets = []  # (key, set of points), first-appearance order
    for key, pt, _ in queries:
        for entry in commitment_sets:
            if entry[0] == key:
                entry[1].add(pt)
                break
        else:
            commitment_sets.append((key, {pt}))
    rotation_sets = []  # (frozenset of points, [keys])
    for key, pts in commitment_sets:
        for entry in rotation_sets:
            if entry[0] == pts:
                entry[1].append(key)
                break
        else:
            rotation_sets.append((set(pts), [key]))
    out = []
    for pts, keys in rotation_sets:
        spts = sorted(pts)  # BTreeSet<Fr> iterates in increasing canonical value
        out.append((spts, [(key, [get_eval(key, p) for p in spts]) for key in keys]))
    return out, super_points


def create_proof(pk: ProvingKey, x_witness: int, seed: int) -> dict:
    """-> {"proof": bytes, plus every intermediate a test may want to compare}."""
    k, n, dom = pk.k, pk.n, pk.dom
    inst = P.StandardPlonkInstance(k, x_witness, seed)
    tr = fmt.Blake2bTranscript()
    tr.common_scalar(pk.transcript_repr)  # vk.hash_into
    # no instance columns (&[&[]]); one phase of advice
    advice_commitments = [pk.commit_lagrange(c) for c in inst.advice]
    for c in advice_commitments:
        tr.write_point(c)
    theta = tr.squeeze_challenge()  # no lookups, but the challenge is still drawn
    beta = tr.squeeze_challenge()
    gamma = tr.squeeze_challenge()
    zs = inst.permutation_products(beta, gamma)
    z_commitments = [pk.commit_lagrange(z) for z in zs]
    for c in z_commitments:
        tr.write_point(c)
    random_poly = o.unpack(o.random_field_limbs(n, seed + 3), R)  # vanishing::Argument::commit
    random_commitment = pk.commit(random_poly)
    tr.write_point(random_commitment)
    y = tr.squeeze_challenge()
    advice_polys = [dom.lagrange_to_coeff(c) for c in inst.advice]
    z_polys = [dom.lagrange_to_coeff(z) for z in zs]
    h_ext = inst.divide_by_vanishing(inst.evaluate_h(zs, beta, gamma, y))
    h_coeffs = dom.extended_to_coeff(h_ext)
    pieces = [h_coeffs[i * n : (i + 1) * n] for i in range(dom.quotient_poly_degree)]
    h_commitments = [pk.commit(p) for p in pieces]
    for c in h_commitments:
        tr.write_point(c)
    x = tr.squeeze_challenge()
    xn = pow(x, n, R)
    ev = o.eval_polynomial
    advice_evals = [ev(advice_polys[c], rotate_omega(dom, x, r)) for c, r in ADVICE_QUERIES]
    for e in advice_evals:
        tr.write_scalar(e)
    fixed_evals = [ev(pk.fixed_polys[c], rotate_omega(dom, x, r)) for c, r in FIXED_QUERIES]
    for e in fixed_evals:
        tr.write_scalar(e)
    # vanishing.evaluate: h(X) = sum_i xn^i h_i(X); only the random polynomial's evaluation is written
    h_poly = [0] * n
    for piece in reversed(pieces):
        h_poly = [(a * xn + b) % R for a, b in zip(h_poly, piece)]
    random_eval = ev(random_poly, x)
    tr.write_scalar(random_eval)
    sigma_evals = [ev(p, x) for p in pk.sigma_polys]  # pk.permutation.evaluate
    for e in sigma_evals:
        tr.write_scalar(e)
    x_next = rotate_omega(dom, x, 1)
    x_last = rotate_omega(dom, x, -(P.BLINDING_FACTORS + 1))
    z_evals = []
    for i, zp in enumerate(z_polys):  # permutation::prover::Constructed::evaluate
        e, en = ev(zp, x), ev(zp, x_next)
        tr.write_scalar(e)
        tr.write_scalar(en)
        el = None
        if i + 1 < len(z_polys):
            el = ev(zp, x_last)
            tr.write_scalar(el)
        z_evals.append((e, en, el))
    # ---- the queries, in create_proof's order; polynomials are identified by a key -------------------------
    polys = {}
    queries = []

    def q(key, poly, pt):
        polys[key] = poly
        queries.append((key, pt, ev(poly, pt)))

    for c, r in ADVICE_QUERIES:
        q(("advice", c), advice_polys[c], rotate_omega(dom, x, r))
    for i, zp in enumerate(z_polys):  # permutation.open
        q(("z", i), zp, x)
        q(("z", i), zp, x_next)
    for i in reversed(range(len(z_polys) - 1)):  # sets.iter().rev().skip(1)
        q(("z", i), z_polys[i], x_last)
    for c, r in FIXED_QUERIES:
        q(("fixed", c), pk.fixed_polys[c], rotate_omega(dom, x, r))
    for i, sp in enumerate(pk.sigma_polys):  # pk.permutation.open
        q(("sigma", i), sp, x)
    q(("h",), h_poly, x)  # vanishing.open
    q(("random",), random_poly, x)
    # ---- ProverSHPLONK::create_proof -------------------------------------------------------------------------
    y_sh = tr.squeeze_challenge()
    rotation_sets, super_points = construct_intermediate_sets(queries)
    v = tr.squeeze_challenge()

    def lincomb(items):  # [(scalar, poly)] -> poly
        out = [0] * n
        for sc, p in items:
            for i, c in enumerate(p):
                out[i] = (out[i] + sc * c) % R
        return out

    quotients = []
    for pts, comms in rotation_sets:
        nx = [0] * n
        yp = 1
        for key, evals in comms:
            r_x = lagrange_interpolate(pts, evals)
            num = list(polys[key])
            for i, c in enumerate(r_x):
                num[i] = (num[i] - c) % R
            nx = [(a + yp * b) % R for a, b in zip(nx, num)]
            yp = yp * y_sh % R
        qx = div_by_vanishing(nx, pts)
        quotients.append(qx + [0] * (n - len(qx)))
    vp = 1
    h_x = [0] * n
    for qx in quotients:
        h_x = [(a + vp * b) % R for a, b in zip(h_x, qx)]
        vp = vp * v % R
    h1 = pk.commit(h_x)
    tr.write_point(h1)
    u = tr.squeeze_challenge()
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x = [0] * n
    z_diffs = []
    vp = 1
    for pts, comms in rotation_sets:
        diffs = [p for p in super_points if p not in pts]
        z_i = evaluate_vanishing_polynomial(diffs, u)
        z_diffs.append(z_i)
        inner = [0] * n
        yp = 1
        for key, evals in comms:
            r_u = ev(lagrange_interpolate(pts, evals), u)
            lin = list(polys[key])
            lin[0] = (lin[0] - r_u) % R
            inner = [(a + yp * b) % R for a, b in zip(inner, lin)]
            yp = yp * y_sh % R
        l_x = [(a + vp * z_i % R * b) % R for a, b in zip(l_x, inner)]
        vp = vp * v % R
    l_x = [(a - zt_eval * b) % R for a, b in zip(l_x, h_x)]
    assert ev(l_x, u) == 0  # the crate's debug assertion
    h2_x = div_by_vanishing(l_x, [u])
    z0inv = pow(z_diffs[0], -1, R)
    h2_x = [c * z0inv % R for c in h2_x] + [0]
    h2 = pk.commit(h2_x)
    tr.write_point(h2)
    return {
        "proof": bytes(tr.proof),
        "challenges": {"theta": theta, "beta": beta, "gamma": gamma, "y": y, "x": x, "shplonk_y": y_sh, "v": v, "u": u},
        "advice_commitments": advice_commitments, "z_commitments": z_commitments, "random_commitment": random_commitment,
        "h_commitments": h_commitments, "shplonk_commitments": [h1, h2],
        "advice_evals": advice_evals, "fixed_evals": fixed_evals, "random_eval": random_eval, "sigma_evals": sigma_evals,
        "z_evals": z_evals, "quotients": quotients, "h_coeffs": h_coeffs, "h_x": h_x, "h2_x": h2_x, "instance": inst, "zs": zs,
    }


class ProofReader:
    def __init__(self, proof: bytes):
        self.tr = fmt.Blake2bTranscript()
        self.buf = proof
        self.pos = 0

    def _take(self, nbytes):
        b = self.buf[self.pos : self.pos + nbytes]
        if len(b) != nbytes:
            raise ValueError("proof too short")
        self.pos += nbytes
        return b

    def read_point(self):
        p = fmt.g1_from_bytes(self._take(32))
        self.tr.common_point(p)
        return p

    def read_scalar(self):
        s = fmt.fr_from_repr(self._take(32))
        if s is None:
            raise ValueError("non-canonical scalar")
        self.tr.common_scalar(s)
        return s

    def squeeze(self):
        return self.tr.squeeze_challenge()


def verify_proof(pk, proof: bytes, g2=None, s_g2=None) -> bool:
    """plonk/verifier.rs verify_proof + VerifierSHPLONK.  `pk`: a ProvingKey or a VerifierKey — only k, the domain, the
    commitments and transcript_repr are used, plus ONE of: the SRS's two G2 elements (g2, s_g2: affine points over Fq2)
    for the real final check e(h2, [s]G2) = e(outer, G2) (oracle/pairing.py; what the reference's verifier does with
    `params.verifier_params()`), or, when they are not given, pk.s for the same equation in G1 (s * h2 == outer)."""
    dom, n = pk.dom, pk.n
    try:
        rd = ProofReader(proof)
        rd.tr.common_scalar(pk.transcript_repr)
        advice_c = [rd.read_point() for _ in range(3)]
        rd.squeeze()  # theta
        beta, gamma = rd.squeeze(), rd.squeeze()
        z_c = [rd.read_point() for _ in range(3)]
        random_c = rd.read_point()
        y = rd.squeeze()
        h_c = [rd.read_point() for _ in range(dom.quotient_poly_degree)]
        x = rd.squeeze()
        advice_evals = [rd.read_scalar() for _ in ADVICE_QUERIES]
        fixed_evals = [rd.read_scalar() for _ in FIXED_QUERIES]
        random_eval = rd.read_scalar()
        sigma_evals = [rd.read_scalar() for _ in range(3)]
        z_evals = []
        for i in range(3):
            e, en = rd.read_scalar(), rd.read_scalar()
            el = rd.read_scalar() if i < 2 else None
            z_evals.append((e, en, el))
    except ValueError:
        return False
    xn = pow(x, n, R)
    bf = P.BLINDING_FACTORS
    # l_i(x) for i in -(bf+1) ..= 0:  l_i(x) = (x^n - 1) / n * omega^i / (x - omega^i)
    l_evals = []
    for rot in range(-(bf + 1), 1):
        wi = pow(dom.omega, rot % n, R)
        l_evals.append((xn - 1) * pow(n, -1, R) % R * wi % R * pow((x - wi) % R, -1, R) % R)
    l_last, l_blind, l_0 = l_evals[0], sum(l_evals[1 : 1 + bf]) % R, l_evals[1 + bf]
    a, b, c = advice_evals
    f = fixed_evals
    exprs = [(f[0] * a + f[1] * b + f[2] * c + f[3] * a * b + f[4]) % R]
    exprs.append(l_0 * (1 - z_evals[0][0]) % R)
    exprs.append((z_evals[2][0] * z_evals[2][0] - z_evals[2][0]) * l_last % R)
    for i in (1, 2):
        exprs.append((z_evals[i][0] - z_evals[i - 1][2]) * l_0 % R)
    for i in range(3):
        left = z_evals[i][1] * (advice_evals[i] + beta * sigma_evals[i] + gamma) % R
        right = z_evals[i][0] * (advice_evals[i] + beta * x % R * pow(P.FR_DELTA, i, R) + gamma) % R
        exprs.append((left - right) * (1 - (l_last + l_blind)) % R)
    expected_h = 0
    for e in exprs:
        expected_h = (expected_h * y + e) % R
    expected_h = expected_h * pow((xn - 1) % R, -1, R) % R
    h_commitment = None
    for cmt in reversed(h_c):
        h_commitment = o.g1_add(o.g1_mul(xn, h_commitment) if h_commitment else None, cmt)
    x_next = rotate_omega(dom, x, 1)
    x_last = rotate_omega(dom, x, -(bf + 1))
    points = {}
    queries = []

    def q(key, cmt, pt, evl):
        points[key] = cmt
        queries.append((key, pt, evl))

    for qi, (col, r) in enumerate(ADVICE_QUERIES):
        q(("advice", col), advice_c[col], rotate_omega(dom, x, r), advice_evals[qi])
    for i in range(3):
        q(("z", i), z_c[i], x, z_evals[i][0])
        q(("z", i), z_c[i], x_next, z_evals[i][1])
    for i in reversed(range(2)):
        q(("z", i), z_c[i], x_last, z_evals[i][2])
    for qi, (col, r) in enumerate(FIXED_QUERIES):
        q(("fixed", col), pk.fixed_commitments[col], rotate_omega(dom, x, r), fixed_evals[qi])
    for i in range(3):
        q(("sigma", i), pk.permutation_commitments[i], x, sigma_evals[i])
    q(("h",), h_commitment, x, expected_h)
    q(("random",), random_c, x, random_eval)
    rotation_sets, super_points = construct_intermediate_sets(queries)
    y_sh, v = rd.squeeze(), rd.squeeze()
    try:
        h1 = rd.read_point()
        u = rd.squeeze()
        h2 = rd.read_point()
    except ValueError:
        return False
    if rd.pos != len(proof):
        return False
    outer = None
    r_outer = 0
    z_0 = z_0_diff_inv = 0
    vp = 1
    for i, (pts, comms) in enumerate(rotation_sets):
        diffs = [p for p in super_points if p not in pts]
        z_diff = evaluate_vanishing_polynomial(diffs, u)
        if i == 0:
            z_0 = evaluate_vanishing_polynomial(pts, u)
            z_0_diff_inv = pow(z_diff, -1, R)
            z_diff = 1
        else:
            z_diff = z_diff * z_0_diff_inv % R
        inner = None
        r_inner = 0
        yp = 1
        for key, evals in comms:
            r_inner = (r_inner + yp * o.eval_polynomial(lagrange_interpolate(pts, evals), u)) % R
            inner = o.g1_add(inner, o.g1_mul(yp, points[key]))
            yp = yp * y_sh % R
        r_outer = (r_outer + vp * r_inner % R * z_diff) % R
        outer = o.g1_add(outer, o.g1_mul(vp * z_diff % R, inner))
        vp = vp * v % R
    outer = o.g1_add(outer, o.g1_mul((-r_outer) % R, o.G1_GEN))
    outer = o.g1_add(outer, o.g1_mul((-z_0) % R, h1))
    outer = o.g1_add(outer, o.g1_mul(u, h2))
    # e(h2, [s]G2) == e(outer, G2)  <=>  s * h2 == outer
    if g2 is not None and s_g2 is not None:
        from . import pairing

        if outer is None:
            return h2 is None
        return pairing.pairing_product_is_one([(h2, s_g2), (o.g1_neg(outer), g2)])
    return o.g1_mul(pk.s, h2) == outer
