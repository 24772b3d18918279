"""GPU tests of the lookup argument on the device (BASELINE config 3: range.rs with LOOKUP_BITS): the
permuted input / table columns (a counting sort against the fixed table instead of the crate's sort + BTreeMap walk), the
lookup grand product and `evaluate_h` of the range-check constraint system (degree 4, extended domain 4n) against
oracle/lookup.py — element for element at k <= 8, through the quotient identity at k = 16."""
import numpy as np
import pytest

from oracle import bn254 as o
from oracle import lookup as L
from oracle.plonk import BLINDING_FACTORS

pytestmark = pytest.mark.gpu


def _vals(buf, count):
    return o.unpack(buf.to_numpy(shape=(count, 4), nbytes=count * 32), o.R)


def _dev(gpu, values):
    return gpu.DevBuf.from_numpy(o.pack(values, o.R))


@pytest.mark.parametrize("k,case", [(5, "mixed"), (6, "one value"), (7, "all distinct"), (8, "two values"), (8, "mixed"), (8, "zeros in table")])
def test_lookup_permuted_columns_match_oracle(gpu, k, case):
    """permute_expression_pair on the device == the crate's construction restated in the oracle, for input multisets that
    stress it: every row the same value (one run, every other row repeated), all distinct (no repeated row: S' = A'),
    two values, and a table that itself holds a repeated value (the zero padding of a range table)."""
    from halo2_scaffold_amd import plonk as gp

    n = 1 << k
    u = n - (BLINDING_FACTORS + 1)
    rng = np.random.default_rng(k + len(case))
    if case == "zeros in table":
        tbl = list(range(1 << (k - 2))) + [0] * (n - (1 << (k - 2)))  # a range table padded with zeros
        inputs = [int(rng.integers(0, 1 << (k - 2))) for _ in range(u)]
    else:
        tbl = [(i * 0x9E3779B97F4A7C15 + 12345) % o.R for i in range(n)]  # arbitrary distinct field elements
        if case == "one value":
            inputs = [tbl[7]] * u
        elif case == "all distinct":
            inputs = list(rng.permutation(tbl[:u]))
        elif case == "two values":
            inputs = [tbl[3] if rng.random() < 0.3 else tbl[u - 1] for _ in range(u)]
        else:
            inputs = [tbl[int(rng.integers(0, u // 3))] for _ in range(u)]
    inputs = [int(v) for v in inputs]
    blind = list(range(1000, 1000 + BLINDING_FACTORS + 1))
    want_a, want_s = L.permute_expression_pair(inputs + [0] * (n - u), tbl, u, blind, blind)
    table = gp.LookupTable(tbl, u)
    d_in = _dev(gpu, inputs + [0] * (n - u))
    d_a, d_s = _dev(gpu, [0] * u + blind), _dev(gpu, [0] * u + blind)  # blinding rows are the caller's
    assert gp.lookup_permute(k, d_in, table, d_a, d_s) == 0
    assert _vals(d_a, n) == want_a
    assert _vals(d_s, n) == want_s
    # an input outside the table is counted (the crate fails the proof)
    bad = list(inputs)
    bad[u // 2] = (max(tbl) + 1) % o.R
    bad[0] = (max(tbl) + 2) % o.R
    d_bad = _dev(gpu, bad + [0] * (n - u))
    assert gp.lookup_permute(k, d_bad, table, d_a, d_s) == 2
    for b in (d_in, d_a, d_s, d_bad):
        b.free()
    table.free()


@pytest.mark.parametrize("k,bits,extra", [(5, 3, 0), (6, 4, 2), (8, 6, 2)])
def test_lookup_product_and_evaluate_h_range_match_oracle(gpu, k, bits, extra):
    """lookup grand product, permutation products in chunks of two and the whole quotient numerator of the range-check
    constraint system (gate + permutation terms + five lookup terms, divided by X^n - 1) == oracle, element for element;
    then the quotient identity on the device-produced h(X)."""
    from halo2_scaffold_amd import plonk as gp

    inst = L.RangeInstance(k, bits, seed=10 + k, extra_cols=extra)
    n, u = inst.n, inst.u
    beta, gamma, y, x = 0xBE7A + k, 0x6A33A, 0x1234567, 0xFEDCBA987654321
    a_perm, s_perm = inst.permuted()
    z_lk = L.lookup_product(inst.la, inst.table, a_perm, s_perm, beta, gamma, u, inst.blind(BLINDING_FACTORS))
    zs = inst.permutation_products(beta, gamma)
    # device: permuted columns, lookup product, permutation products
    table = gp.LookupTable(inst.table, u)
    d_la, d_t = _dev(gpu, inst.la), _dev(gpu, inst.table)
    d_ap, d_sp = _dev(gpu, [0] * u + a_perm[u:]), _dev(gpu, [0] * u + s_perm[u:])
    assert gp.lookup_permute(k, d_la, table, d_ap, d_sp) == 0
    assert _vals(d_ap, n) == a_perm and _vals(d_sp, n) == s_perm
    d_zl = _dev(gpu, [0] * (u + 1) + z_lk[u + 1 :])
    gp.lookup_product(k, d_la, d_t, d_ap, d_sp, beta, gamma, u, d_zl)
    assert _vals(d_zl, n) == z_lk
    d_cols = [_dev(gpu, c) for c in inst.perm_cols]
    d_sig = [_dev(gpu, c) for c in inst.sigma]
    d_zs = [_dev(gpu, [0] * (u + 1) + z[u + 1 :]) for z in zs]
    gp.permutation_products(k, d_cols, d_sig, L.CS_DEGREE - 2, beta, gamma, u, d_zs)
    assert [_vals(z, n) for z in d_zs] == zs
    # extended cosets on the device, evaluate_h
    dom = gpu.EvaluationDomain(L.CS_DEGREE, k)
    ext = dom.extended_len()
    assert ext == 4 * n

    def to_ext(d_lagr):
        p, e = gpu.DevBuf(n * 32), gpu.DevBuf(ext * 32)
        dom.lagrange_to_coeff_oop_dev(d_lagr, p)
        dom.coeff_to_extended_oop_dev(p, e)
        p.free()
        return e

    e_a, e_la, e_q, e_t = to_ext(d_cols[0]), to_ext(d_la), to_ext(_dev(gpu, inst.q)), to_ext(d_t)
    e_cols = [to_ext(c) for c in d_cols]
    e_sig = [to_ext(c) for c in d_sig]
    e_zs = [to_ext(z) for z in d_zs]
    e_ap, e_sp, e_zl = to_ext(d_ap), to_ext(d_sp), to_ext(d_zl)
    e_l0, e_ll, e_la_ = to_ext(_dev(gpu, inst.l0)), to_ext(_dev(gpu, inst.l_last)), to_ext(_dev(gpu, inst.l_active))
    out = gpu.DevBuf(ext * 32)
    gp.evaluate_h_range(dom, e_a, e_la, e_q, e_t, e_cols, e_sig, e_zs, e_ap, e_sp, e_zl, e_l0, e_ll, e_la_, beta, gamma, y, out)
    want = inst.divide_by_vanishing(inst.evaluate_h(zs, a_perm, s_perm, z_lk, beta, gamma, y))
    assert _vals(out, ext) == want
    dom.extended_to_coeff_dev(out)
    hc = _vals(out, ext)
    assert not any(hc[3 * n :])  # degree bound: the division was exact
    assert L.check_quotient_identity(inst, zs, a_perm, s_perm, z_lk, hc[: 3 * n], beta, gamma, y, x)


def test_range_quotient_identity_at_2pow16(gpu):
    """size-independent check at DEGREE = 16 with LOOKUP_BITS = 12: everything after the witness is produced on the
    device (permuted columns, both grand products, 17 coset transforms, evaluate_h_range, the coset iNTT) and the
    verifier's equation is evaluated at a random point with the device's eval_polynomial on the device-resident
    polynomials.  A tampered lookup column (one value swapped for another table value) must break it."""
    import ctypes as C

    from halo2_scaffold_amd import field as F
    from halo2_scaffold_amd import plonk as gp

    k, bits = 16, 12
    inst = L.RangeInstance(k, bits, seed=77, gates=3000, extra_cols=2)
    n, u = inst.n, inst.u
    beta, gamma, y = 0xB0B0B0B, 0xCACACA, 0xD0D0D0D0
    dom = gpu.EvaluationDomain(L.CS_DEGREE, k)
    ext = dom.extended_len()
    blind = inst.blind(3 * (BLINDING_FACTORS + 1) + 2 * BLINDING_FACTORS)
    table = gp.LookupTable(inst.table, u)
    d = lambda vals: _dev(gpu, vals)
    polys = {}

    def to_poly_and_ext(key, d_lagr):
        p, e = gpu.DevBuf(n * 32), gpu.DevBuf(ext * 32)
        dom.lagrange_to_coeff_oop_dev(d_lagr, p)
        dom.coeff_to_extended_oop_dev(p, e)
        polys[key] = p
        return e

    def run(la_values):
        d_la, d_t = d(la_values), d(inst.table)
        d_ap, d_sp = d([0] * u + blind[:6]), d([0] * u + blind[6:12])
        missing = gp.lookup_permute(k, d_la, table, d_ap, d_sp)
        d_zl = d([0] * (u + 1) + blind[12:17])
        gp.lookup_product(k, d_la, d_t, d_ap, d_sp, beta, gamma, u, d_zl)
        cols = [inst.a, la_values] + inst.extras
        d_cols = [d(c) for c in cols]
        d_sig = [d(c) for c in inst.sigma]
        d_zs = [d([0] * (u + 1) + blind[17 + 5 * s : 22 + 5 * s]) for s in range(2)]
        gp.permutation_products(k, d_cols, d_sig, 2, beta, gamma, u, d_zs)
        e_cols = [to_poly_and_ext(("col", j), c) for j, c in enumerate(d_cols)]
        e_sig = [to_poly_and_ext(("sigma", j), c) for j, c in enumerate(d_sig)]
        e_zs = [to_poly_and_ext(("z", s), z) for s, z in enumerate(d_zs)]
        e_q, e_t = to_poly_and_ext("q", d(inst.q)), to_poly_and_ext("table", d_t)
        e_ap, e_sp, e_zl = to_poly_and_ext("a_perm", d_ap), to_poly_and_ext("s_perm", d_sp), to_poly_and_ext("z_lk", d_zl)
        e_l0, e_ll, e_lact = to_poly_and_ext("l0", d(inst.l0)), to_poly_and_ext("l_last", d(inst.l_last)), to_poly_and_ext("l_active", d(inst.l_active))
        h = gpu.DevBuf(ext * 32)
        gp.evaluate_h_range(dom, e_cols[0], e_cols[1], e_q, e_t, e_cols, e_sig, e_zs, e_ap, e_sp, e_zl, e_l0, e_ll, e_lact, beta, gamma, y, h)
        dom.extended_to_coeff_dev(h)
        top = h.to_numpy(shape=(n, 4), nbytes=n * 32, offset=3 * n * 32)
        return missing, h, not top.any(), _vals(d_zl, u + 1)[u], _vals(d_zs[1], u + 1)[u]

    out32 = gpu.DevBuf(32)

    def ev_dev(buf, point, offset=0):
        pl = F.fr_to_mont_limbs(point)
        assert gpu.lib.h2mi_fr_eval_poly_dev(buf.ptr + offset * 32, n, pl.ctypes.data, out32.ptr, None) == 0
        return _vals(out32, 1)[0]

    key_of = {id(inst.a): ("col", 0), id(inst.q): "q", id(inst.table): "table", id(inst.l0): "l0", id(inst.l_last): "l_last",
              id(inst.l_active): "l_active"}
    for j, c in enumerate(inst.sigma):
        key_of[id(c)] = ("sigma", j)
    for j, c in enumerate(inst.extras):
        key_of[id(c)] = ("col", 2 + j)

    def check(la_values, h):
        key_of[id(la_values)] = ("col", 1)
        inst.la = la_values
        inst.perm_cols = [inst.a, la_values] + inst.extras
        ev = lambda obj, pt: ev_dev(polys[obj if isinstance(obj, (str, tuple)) else key_of[id(obj)]], pt)
        pt = 0x1F2E3D4C5B6A79881726354453627180 % o.R
        ptn = pow(pt, n, o.R)
        h_at = (ev_dev(h, pt) + ptn * ev_dev(h, pt, n) + ptn * ptn % o.R * ev_dev(h, pt, 2 * n)) % o.R

        class HC:  # check_quotient_identity evaluates h_coeffs with the oracle's Horner: hand it the value instead
            pass

        orig = o.eval_polynomial
        o.eval_polynomial = lambda coeffs, point: h_at if coeffs is HC else orig(coeffs, point)
        try:
            return L.check_quotient_identity(inst, [("z", 0), ("z", 1)], "a_perm", "s_perm", "z_lk", HC, beta, gamma, y, pt, ev=ev)
        finally:
            o.eval_polynomial = orig

    la_good = list(inst.la)
    missing, h, exact, zl_end, zp_end = run(la_good)
    assert missing == 0 and exact and zl_end == 1 and zp_end == 1
    assert check(la_good, h)
    # tamper: a lookup cell that is copy-constrained to a gate cell gets another (valid) table value: the lookup still
    # closes, the permutation argument does not, the numerator is no longer divisible and the identity fails
    la_bad = list(inst.la)
    la_bad[3] = (la_bad[3] + 1) % (1 << bits)
    missing, h_bad, exact_bad, zl_end, zp_end = run(la_bad)
    assert missing == 0 and zl_end == 1 and zp_end != 1
    assert not exact_bad or not check(la_bad, h_bad)
