"""CPU tests: known-answer tests of the oracle (pure Python and C restatement) and the golden vectors.

The reference holds no vectors for this path (parity unpinned), so the oracle is pinned by
mathematical known answers: curve membership, group order, root-of-unity orders, DFT definition,
independent formulations (affine vs Jacobian; naive DFT vs radix-2; big-int vs 64-bit-limb C), and the
KZG identity commit(f; g) == commit(NTT f; g_lagrange).
"""
import json
import os

import numpy as np
import pytest

from oracle import bn254 as o
from oracle import cref

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_vectors.json")))


def _pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def test_constants():
    c = GOLD["constants"]
    assert int(c["q"], 16) == o.Q and int(c["r"], 16) == o.R
    u = 4965661367192848881  # BN parameter
    assert o.Q == 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1
    assert o.R == 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    assert o.Q % 4 == 3
    assert (o.R - 1) % (1 << 28) == 0 and (o.R - 1) % (1 << 29) != 0
    assert pow(o.FR_ROOT_OF_UNITY, 1 << 28, o.R) == 1 and pow(o.FR_ROOT_OF_UNITY, 1 << 27, o.R) == o.R - 1
    assert o.FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(o.FR_ZETA, 3, o.R) == 1 and o.FR_ZETA != 1
    assert int(c["fr_mont_R"], 16) == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    assert int(c["fq_mont_R2"], 16) == 0x06D89F71CAB8351F47AB1EFF0A417FF6B5E71911D44501FBF32CFC5B538AFA89
    assert int(c["fr_inv64"], 16) == 0xC2E1F593EFFFFFFF and int(c["fq_inv64"], 16) == 0x87D20782E4866389
    for kk in range(0, 12):
        w = o.omega_for(kk)
        assert pow(w, 1 << kk, o.R) == 1
        if kk:
            assert pow(w, 1 << (kk - 1), o.R) == o.R - 1


def test_curve_kats():
    G = o.G1_GEN
    assert o.is_on_curve(G)
    assert o.g1_mul(o.R, G) is None  # r * G = identity (cofactor 1)
    assert o.g1_mul(o.R - 1, G) == (1, o.Q - 2)
    g = GOLD["g1"]
    assert _pt(g["2G"]) == o.g1_double(G) == o.jac_to_affine(o.jac_double(o.jac_from_affine(G)))
    assert _pt(g["3G"]) == o.jac_to_affine(o.jac_add(o.jac_double(o.jac_from_affine(G)), o.jac_from_affine(G)))
    assert _pt(g["12345G"]) == o.jac_to_affine(o.jac_mul(12345, o.jac_from_affine(G)))
    for p in g.values():
        assert o.is_on_curve(_pt(p))
    # P + (-P), P + identity, doubling through add
    P = _pt(g["seedG"])
    assert o.g1_add(P, o.g1_neg(P)) is None and o.g1_add(P, None) == P
    assert o.jac_to_affine(o.jac_add(o.jac_from_affine(P), o.jac_from_affine(P))) == o.g1_double(P)


def test_ntt_kats():
    for k in range(0, 8):
        n = 1 << k
        w = o.omega_for(k)
        a = [(i * i * 31 + 7) % o.R for i in range(n)]
        assert o.ntt(a, w) == o.dft_naive(a, w)
        assert o.intt(o.ntt(a, w), w) == a
        assert o.ntt([1] + [0] * (n - 1), w) == [1] * n
        assert o.ntt([1] * n, w) == [n % o.R] + [0] * (n - 1)
    t = GOLD["ntt"]
    a = [int(x, 16) for x in t["input"]]
    assert a == o.unpack(o.random_field_limbs(64, o.SEED + 1), o.R)
    assert [int(x, 16) for x in t["forward"]] == o.dft_naive(a, int(t["omega"], 16))
    assert [int(x, 16) for x in t["coset_zeta"]] == o.ntt_ext(a, int(t["omega"], 16), pre_base=o.FR_ZETA)
    assert [int(x, 16) for x in t["inverse_scaled"]] == o.intt(a, int(t["omega"], 16))


def test_msm_and_srs_kats():
    t = GOLD["msm"]
    bs = [int(x, 16) for x in t["base_scalars"]]
    sc = [int(x, 16) for x in t["scalars"]]
    # sum s_i (b_i G) == (sum s_i b_i) G
    assert _pt(t["result"]) == o.g1_mul(sum(s * b for s, b in zip(sc, bs)) % o.R, o.G1_GEN)
    assert _pt(t["result_all_ones"]) == o.g1_mul(sum(bs) % o.R, o.G1_GEN)
    s = GOLD["srs"]
    g = [_pt(p) for p in s["g"]]
    gl = [_pt(p) for p in s["g_lagrange"]]
    assert g == [o.g1_mul(pow(s["s"], i, o.R), o.G1_GEN) for i in range(8)]
    coeffs = s["coeffs"]
    evals = o.ntt(coeffs, o.omega_for(3))
    assert o.msm_naive(coeffs, g) == o.msm_naive(evals, gl) == _pt(s["commit"])
    fs = sum(c * pow(s["s"], i, o.R) for i, c in enumerate(coeffs)) % o.R
    assert _pt(s["commit"]) == o.g1_mul(fs, o.G1_GEN)
    d = o.Domain(GOLD["domain"]["k"], GOLD["domain"]["j"])
    assert d.extended_k == GOLD["domain"]["extended_k"]
    assert [int(x, 16) for x in GOLD["domain"]["coeff_to_extended"]] == d.coeff_to_extended(coeffs)
    assert d.extended_to_coeff(d.coeff_to_extended(coeffs)) == coeffs + [0] * 8


def test_montgomery_packing_roundtrip():
    vals = [0, 1, o.R - 1, 12345678901234567890, (1 << 253) % o.R]
    arr = o.pack(vals, o.R)
    assert o.unpack(arr, o.R) == vals
    assert o.unpack(arr)[1] == o.to_mont(1, o.R)
    pts = [None, o.G1_GEN, o.g1_mul(77, o.G1_GEN)]
    assert o.unpack_points(o.pack_points(pts)) == pts
    assert (o.pack_points([None]) == 0).all()  # identity is (0, 0)


def test_synthetic_generator():
    a = o.random_field_limbs(1000, o.SEED)
    assert all(v < o.R for v in o.unpack(a))
    assert np.array_equal(a, o.random_field_limbs(1000, o.SEED))          # deterministic
    assert np.array_equal(a[100:200], o.random_field_limbs(100, o.SEED, start=100))  # counter based
    w = o.witness_like_limbs(20000, o.SEED)
    nz = w.any(axis=1).mean()
    assert 0.05 < nz < 0.10  # 5 % uniform + 2.5 % ones


# ---- C restatement (oracle/h2ref.c) against the big-integer oracle --------------------------------
def test_c_field_ops():
    for field, mod in [(0, o.Q), (1, o.R)]:
        a = o.random_field_limbs(300, 11, mod)
        b = o.random_field_limbs(300, 12, mod)
        a[0] = 0
        b[1] = o.pack([mod - 1], mod)[0]
        av, bv = o.unpack(a, mod), o.unpack(b, mod)
        assert o.unpack(cref.field_op(field, 0, a, b), mod) == [x * y % mod for x, y in zip(av, bv)]
        assert o.unpack(cref.field_op(field, 1, a, b), mod) == [(x + y) % mod for x, y in zip(av, bv)]
        assert o.unpack(cref.field_op(field, 2, a, b), mod) == [(x - y) % mod for x, y in zip(av, bv)]
        assert o.unpack(cref.field_op(field, 4, a), mod) == [pow(x, -1, mod) if x else 0 for x in av]
        assert o.unpack(cref.field_op(field, 5, a)) == av  # from_mont gives canonical limbs


def test_c_msm_golden_and_threads():
    t = GOLD["msm"]
    bs = o.pack([int(x, 16) for x in t["base_scalars"]], o.R)
    bases = cref.g1_mul_gen(bs, 2)
    sc = o.pack([int(x, 16) for x in t["scalars"]], o.R)
    for threads in (1, 2, 5, 64):
        assert o.unpack_jacobian(cref.msm(sc, bases, threads)) == _pt(t["result"])
    ones = o.pack([1] * 64, o.R)
    assert o.unpack_jacobian(cref.msm(ones, bases, 3)) == _pt(t["result_all_ones"])
    wl = o.witness_like_limbs(64, 3)
    assert o.unpack_jacobian(cref.msm(wl, bases, 1)) == _pt(t["result_witness_like"])


@pytest.mark.parametrize("n,threads", [(1, 1), (2, 1), (3, 2), (5, 1), (31, 1), (33, 4), (200, 3), (1000, 8)])
def test_c_msm_vs_naive(n, threads):
    pts = cref.g1_mul_gen(o.random_field_limbs(n, 21 + n), 4)
    s = o.random_field_limbs(n, 22 + n)
    if n >= 5:  # edge cases: zero scalar, r-1, identity base, duplicate base
        s[0] = 0
        s[1] = o.pack([o.R - 1], o.R)[0]
        pts[2] = 0
        pts[4] = pts[3]
    assert o.unpack_jacobian(cref.msm(s, pts, threads)) == o.msm_naive(o.unpack(s, o.R), o.unpack_points(pts))


@pytest.mark.parametrize("log_n,threads", [(0, 1), (1, 1), (2, 1), (6, 1), (9, 3), (12, 4)])
def test_c_ntt_vs_oracle(log_n, threads):
    a = o.random_field_limbs(1 << log_n, 31)
    w = o.omega_for(log_n)
    x = a.copy()
    cref.ntt(x, o.pack([w], o.R)[0], log_n, threads)
    assert o.unpack(x, o.R) == o.ntt(o.unpack(a, o.R), w)


def test_c_ntt_golden():
    t = GOLD["ntt"]
    a = o.pack([int(x, 16) for x in t["input"]], o.R)
    cref.ntt(a, o.pack([int(t["omega"], 16)], o.R)[0], t["log_n"], 2)
    assert o.unpack(a, o.R) == [int(x, 16) for x in t["forward"]]


def test_golden_replay_k8_self_consistency():
    """BASELINE config 0 (plumbing size, CPU only): the committed replay commitments are what the big-integer
    oracle gets for the same seeded SRS and vectors (spot-check two of the eleven: naive double-and-add MSM)."""
    r = GOLD["replay_k8"]
    k, n = r["k"], 1 << r["k"]
    secret = int(r["srs_secret"], 16)
    pw, lag = o.srs_scalars(k, secret)
    adv0 = o.unpack(o.random_field_limbs(n, o.SEED + 10), o.R)
    # commit_lagrange(advice[0]) = (sum_i adv0[i] * L_i(s)) * G
    assert _pt(r["commitments"][0]) == o.g1_mul(sum(a * l for a, l in zip(adv0, lag)) % o.R, o.G1_GEN)
    rp = o.unpack(o.random_field_limbs(n, o.SEED + 20), o.R)
    assert _pt(r["commitments"][6]) == o.g1_mul(sum(c * p for c, p in zip(rp, pw)) % o.R, o.G1_GEN)
    assert len(r["commitments"]) == 11 and all(o.is_on_curve(_pt(p)) for p in r["commitments"])


def test_plonk_quotient_oracle_identity():
    """oracle/plonk.py: for the reference circuit's satisfying witness the quotient identity holds at random
    points; for a broken witness or a broken copy constraint it does not."""
    from oracle import plonk as P

    beta, gamma, y = 0xB, 0xC, 0xD
    inst = P.StandardPlonkInstance(4, 987654321)
    zs = inst.permutation_products(beta, gamma)
    assert zs[2][inst.u] == 1 and P.FR_DELTA == pow(7, 1 << 28, o.R)
    hc = inst.dom.extended_to_coeff(inst.divide_by_vanishing(inst.evaluate_h(zs, beta, gamma, y)))
    for x in (5, 0xABCDEF, o.R - 2):
        assert P.check_quotient_identity(inst, zs, hc, beta, gamma, y, x)
    bad = P.StandardPlonkInstance(4, 987654321)
    bad.advice[1][2] = (bad.advice[1][2] + 1) % o.R  # b2 != x: breaks the gate and the copy constraint
    zb = bad.permutation_products(beta, gamma)
    hb = bad.dom.extended_to_coeff(bad.divide_by_vanishing(bad.evaluate_h(zb, beta, gamma, y)))
    assert not P.check_quotient_identity(bad, zb, hb, beta, gamma, y, 5)


# ---- third-party anchors (not derived from the oracle): EIP-196 ECADD / ECMUL vectors ---------------------------------
EIP196 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eip196_vectors.json")))


def _eip_pt(xy):
    x, y = int(xy[0], 16), int(xy[1], 16)
    return None if x == 0 and y == 0 else (x, y)


def test_eip196_vectors_python_oracle():
    """the alt_bn128 precompile vectors (independent of this repository) against oracle/bn254.py: affine
    chord-and-tangent addition, Jacobian addition, double-and-add multiplication"""
    for v in EIP196["ecadd"]:
        a, b, want = _eip_pt(v["a"]), _eip_pt(v["b"]), _eip_pt(v["sum"])
        assert o.is_on_curve(a) and o.is_on_curve(b) and o.is_on_curve(want), v["name"]
        assert o.g1_add(a, b) == want, v["name"]
        assert o.jac_to_affine(o.jac_add(o.jac_from_affine(a), o.jac_from_affine(b))) == want, v["name"]
    for v in EIP196["ecmul"]:
        p, k, want = _eip_pt(v["p"]), int(v["k"], 16), _eip_pt(v["product"])
        assert o.g1_mul(k % o.R, p) == want, v["name"]
        assert o.jac_to_affine(o.jac_mul(k % o.R, o.jac_from_affine(p))) == want, v["name"]
    assert o.FR_ROOT_OF_UNITY == int(EIP196["fr_root_of_unity"], 16) and o.FR_S == EIP196["fr_two_adicity"]
    assert pow(o.FR_ROOT_OF_UNITY, 1 << 28, o.R) == 1 and pow(o.FR_ROOT_OF_UNITY, 1 << 27, o.R) == o.R - 1
    # Fr::DELTA and Fr::ZETA as recalled from the crate (see the file's provenance): the coset generator and the
    # permutation argument's column separator the whole prover rests on
    from oracle import plonk as P

    assert o.FR_ZETA == int(EIP196["fr_zeta"], 16) and P.FR_DELTA == int(EIP196["fr_delta"], 16)
    assert o.FR_ZETA * o.FR_ZETA % o.R == int(EIP196["fr_zeta_squared_other_root"], 16) and pow(o.FR_ZETA, 3, o.R) == 1


def test_eip196_vectors_c_restatement():
    """the same vectors through oracle/h2ref.c: the serial Pippenger (n = 2 with scalars 1, 1 is an addition; n = 1 a
    scalar multiplication) and its 64-bit-limb Montgomery arithmetic"""
    from oracle import cref

    for v in EIP196["ecadd"]:
        a, b, want = _eip_pt(v["a"]), _eip_pt(v["b"]), _eip_pt(v["sum"])
        got = cref.msm(o.pack([1, 1], o.R), o.pack_points([a, b]), 1)
        assert o.unpack_jacobian(got) == want, v["name"]
    for v in EIP196["ecmul"]:
        p, k, want = _eip_pt(v["p"]), int(v["k"], 16), _eip_pt(v["product"])
        assert o.unpack_jacobian(cref.msm(o.pack([k % o.R], o.R), o.pack_points([p]), 1)) == want, v["name"]
