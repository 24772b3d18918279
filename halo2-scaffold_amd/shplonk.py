"""ProverSHPLONK: the multi-open argument of create_proof on device-resident polynomials.

Mirror of halo2_proofs::poly::kzg::multiopen::ProverSHPLONK::create_proof as the reference selects it
(examples/standard_plonk.rs:41-49, src/scaffold.rs:191-199,322-331) — poly/kzg/multiopen/shplonk.rs
`construct_intermediate_sets` and shplonk/prover.rs [restated from memory of v2023_02_02]:

  queries (polynomial, point, eval) are grouped by polynomial into point sets and polynomials with the same point
  set form a rotation set; y, v are drawn; per set i:   N_i(X) = sum_j y^j (P_ij(X) - R_ij(X)),  Q_i = N_i / Z_i;
  h(X) = sum_i v^i Q_i is committed; u is drawn;  L(X) = sum_i v^i Z_{T\\i}(u) sum_j y^j (P_ij(X) - R_ij(u))
  - Z_T(u) h(X);  L(X) / (X - u), scaled by 1 / Z_{T\\0}(u), is committed.

The bookkeeping (sets, the low-degree remainders R_ij from <= 3 evaluations, scalar coefficients) is host work on a
handful of field elements; every operation on a polynomial is a device kernel over HBM-resident vectors:
h2mi_fr_lincomb_dev, h2mi_fr_add_head_dev, h2mi_fr_kate_division_dev and the two commitments (MSM).
"""
import ctypes as C

import numpy as np

from . import field as F
from ._lib import check, lib
from .device import DevBuf, SideStream

R = F.FR_MODULUS
_m = F.fr_to_mont_limbs


def _lagrange_basis(points):
    """coefficient lists (low to high) of the Lagrange basis polynomials of `points`: the part of an interpolation that depends
    on the points alone — computed once per rotation set, its modular inversions included, instead of once per member"""
    k = len(points)
    basis = []
    for j in range(k):
        num, den = [1], 1
        for m in range(k):
            if m == j:
                continue
            nxt = [0] * (len(num) + 1)
            for i, c in enumerate(num):  # num *= (X - points[m])
                nxt[i + 1] = (nxt[i + 1] + c) % R
                nxt[i] = (nxt[i] - c * points[m]) % R
            num = nxt
            den = den * (points[j] - points[m]) % R
        inv = pow(den, -1, R)
        basis.append([c * inv % R for c in num])
    return basis


def _interpolate(points, evals, basis=None):
    """coefficients (low to high) of the polynomial of degree < len(points) through the given values"""
    basis = basis if basis is not None else _lagrange_basis(points)
    out = [0] * len(points)
    for ev, bj in zip(evals, basis):
        for i, c in enumerate(bj):
            out[i] = (out[i] + c * ev) % R
    return out


def _horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def _vanishing_at(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R
    return acc


class RotationSet:
    def __init__(self, points):
        self.points = sorted(points)  # BTreeSet<Fr> order: increasing canonical value
        self.members = []             # (poly DevBuf, [eval at each point])


def construct_intermediate_sets(queries):
    """queries: [(DevBuf poly, point int, eval int)] in create_proof's order -> (rotation sets, super point set)"""
    by_poly = []  # (poly, {point: eval}) in first-appearance order; polynomials are identified by their buffer
    for poly, pt, ev in queries:
        for entry in by_poly:
            if entry[0] is poly:
                entry[1].setdefault(pt, ev)
                break
        else:
            by_poly.append((poly, {pt: ev}))
    sets = []
    for poly, evs in by_poly:
        key = frozenset(evs)
        rs = next((s for s in sets if frozenset(s.points) == key), None)
        if rs is None:
            rs = RotationSet(key)
            sets.append(rs)
        rs.members.append((poly, [evs[p] for p in rs.points]))
    return sets, sorted({pt for _, pt, _ in queries})


def _lincomb(polys, scalars, n, out: DevBuf, stream=None):
    assert len(polys) == len(scalars) <= 24
    ptrs = (C.c_void_p * len(polys))(*[p.ptr for p in polys])
    sc = np.ascontiguousarray(np.stack([_m(s) for s in scalars]))
    check(lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, len(polys), n, out.ptr, stream), "lincomb")


def _add_head(poly: DevBuf, coeffs, stream=None):
    hd = np.ascontiguousarray(np.stack([_m(c) for c in coeffs]))
    check(lib.h2mi_fr_add_head_dev(poly.ptr, hd.ctypes.data, len(coeffs), stream), "add_head")


def _kate_chain(src: DevBuf, n: int, roots, tmp: DevBuf, out: DevBuf, stream=None, tmp2: DevBuf = None):
    """out = src / prod (X - root).  `out` must have been zeroed (the quotient has n - len(roots) coefficients; the rest
    of the n stay zero).  Two to four roots: ONE round of independent divisions weighted by the partial-fraction
    coefficients 1 / prod_{k != i} (r_i - r_k) (h2mi_fr_kate_division_multi_dev) instead of a chain of dependent ones."""
    if len(roots) == 1:
        b, b_inv = _m(roots[0]), _m(pow(roots[0], -1, R))  # named: the arrays must outlive the call that reads their memory
        check(lib.h2mi_fr_kate_division_dev(src.ptr, n, b.ctypes.data, b_inv.ctypes.data, out.ptr, stream), "kate_division")
        return
    if len(roots) <= 4:
        weights = []
        for i, r in enumerate(roots):
            d = 1
            for k, rk in enumerate(roots):
                if k != i:
                    d = d * (r - rk) % R
            weights.append(pow(d, -1, R))
        rl = np.ascontiguousarray(np.stack([_m(r) for r in roots]))
        ri = np.ascontiguousarray(np.stack([_m(pow(r, -1, R)) for r in roots]))
        wl = np.ascontiguousarray(np.stack([_m(w) for w in weights]))
        check(lib.h2mi_fr_kate_division_multi_dev(src.ptr, n, rl.ctypes.data, ri.ctypes.data, wl.ctypes.data, len(roots), out.ptr, stream), "kate_division_multi")
        return
    cur, length = src, n
    bufs = [tmp, tmp2 if tmp2 is not None else src]
    for i, root in enumerate(roots):
        last = i == len(roots) - 1
        dst = out if last else bufs[i % 2]
        b, b_inv = _m(root), _m(pow(root, -1, R))
        check(lib.h2mi_fr_kate_division_dev(cur.ptr, length, b.ctypes.data, b_inv.ctypes.data, dst.ptr, stream), "kate_division")
        cur, length = dst, length - 1


class ProverSHPLONK:
    LANES = 3  # rotation sets worked on at once: the library stream and two side streams, each with its own scratch

    def __init__(self, n: int):
        """n: number of rows (coefficients per polynomial) — the domain size, not the length of an SRS slice"""
        self.n = n
        self._nx = [DevBuf(n * 32) for _ in range(self.LANES)]
        self._tmp = [DevBuf(n * 32) for _ in range(self.LANES)]
        self._q = [DevBuf(n * 32) for _ in range(6)]
        self._s = [DevBuf(n * 32) for _ in range(6)]  # per rotation set: sum_j y^j P_ij(X) - R_i(X), kept for the linearisation
        self._side = [None] + [SideStream() for _ in range(self.LANES - 1)]
        self.h_x, self.l_x, self.h2_x = DevBuf(n * 32), DevBuf(n * 32), DevBuf(n * 32)

    def release(self):
        for b in self._nx + self._tmp + [self.h_x, self.l_x, self.h2_x] + self._q + self._s:
            b.free()
        for st in self._side[1:]:
            st.free()

    def create_proof(self, transcript, queries, commit_and_write) -> None:
        """`commit_and_write(d_poly)` commits a coefficient vector (ParamsKZG::commit), writes the point to the
        transcript (the caller owns the join / device-to-host step) and returns nothing."""
        n = self.n
        y = F.fr_from_mont_limbs(transcript.squeeze_challenge())
        sets, super_points = construct_intermediate_sets(queries)
        assert len(sets) <= len(self._q)
        v = F.fr_from_mont_limbs(transcript.squeeze_challenge())
        # the divisions below need the power tables of every opening point and of its inverse: built now, in one launch
        if 0 < len(super_points) <= 16:
            bases = np.ascontiguousarray(np.stack([_m(p) for p in super_points] + [_m(pow(p, -1, R)) for p in super_points]))
            check(lib.h2mi_fr_powtab_prefetch_dev(bases.ctypes.data, len(bases), n, None), "powtab_prefetch")
        # quotient contributions Q_i = (sum_j y^j (P_ij - R_ij)) / Z_i.  The sets are independent chains of small
        # launches (one linear combination, one division per point of the set): set i runs on lane i mod 3, so the
        # longest chain, not their sum, is what the proof waits for.
        for i in range(len(sets)):
            check(lib.h2mi_memset_zero(self._q[i].ptr, n * 32), "zero")
        for st in self._side[1:]:
            st.after_library()
        remainders = []
        for i, rs in enumerate(sets):
            lane = i % self.LANES
            stream = self._side[lane].handle if lane else None
            nx, tmp = self._nx[lane], self._tmp[lane]
            ypow = [pow(y, j, R) for j in range(len(rs.members))]
            _lincomb([p for p, _ in rs.members], ypow, n, self._s[i], stream)
            rsum = [0] * len(rs.points)
            basis = _lagrange_basis(rs.points)
            for (_, evals), yp in zip(rs.members, ypow):
                for t, c in enumerate(_interpolate(rs.points, evals, basis)):
                    rsum[t] = (rsum[t] - yp * c) % R
            _add_head(self._s[i], rsum, stream)
            remainders.append([(-c) % R for c in rsum])  # R_i(X) = sum_j y^j R_ij(X), low to high
            _kate_chain(self._s[i], n, rs.points, tmp, self._q[i], stream, tmp2=nx)
        for st in self._side[1:]:
            st.join_library()
        _lincomb(self._q[: len(sets)], [pow(v, i, R) for i in range(len(sets))], n, self.h_x)
        commit_and_write(self.h_x)
        u = F.fr_from_mont_limbs(transcript.squeeze_challenge())
        # linearisation.  sum_j y^j (P_ij(X) - R_ij(u)) = S_i(X) + R_i(X) - R_i(u) with S_i = sum_j y^j P_ij - R_i the vector the
        # quotient step left in self._s[i]: one linear combination over the rotation sets' sums and h(X) — four or six
        # vectors instead of every opened polynomial again (22 at 2^20 rows: 0.32 -> 0.07 ms) — and a low-degree head
        zt_eval = _vanishing_at(super_points, u)
        z_diffs = [_vanishing_at([p for p in super_points if p not in rs.points], u) for rs in sets]
        norm = pow(z_diffs[0], -1, R)  # "normalize coefficients by the coefficient of the first polynomial"
        polys, scalars, head = [], [], [0] * max(len(rs.points) for rs in sets)
        for i, rs in enumerate(sets):
            w = pow(v, i, R) * z_diffs[i] % R * norm % R
            polys.append(self._s[i])
            scalars.append(w)
            for t, c in enumerate(remainders[i]):
                head[t] = (head[t] + w * c) % R
            head[0] = (head[0] - w * _horner(remainders[i], u)) % R
        polys.append(self.h_x)
        scalars.append((-zt_eval * norm) % R)
        _lincomb(polys, scalars, n, self.l_x)
        _add_head(self.l_x, head)
        check(lib.h2mi_memset_zero(self.h2_x.ptr, n * 32), "zero")
        _kate_chain(self.l_x, n, [u], self._tmp[0], self.h2_x)
        commit_and_write(self.h2_x)
