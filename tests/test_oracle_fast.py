"""The large-size oracle prover (oracle/fastflex.py over oracle/vec.py + the C loops of oracle/h2ref.c) against the
Python-integer oracle (oracle/flex.py, oracle/bn254.py): every vector helper element for element, and the whole proof byte
for byte for every shape at sizes the slow one finishes in seconds.  This is what lets tests/golden/big_proofs.json
(k = 16 / 20 / 22, made by the fast one) stand in for the slow oracle at sizes it cannot reach."""
import json
import os

import pytest

from oracle import bn254 as o
from oracle import fastflex as FF
from oracle import flex as FX
from oracle.vec import FV, FastDomain

R = o.R
SRS_SECRET = 0x5EC2E7 + 0x48324D49
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_vector_helpers_match_python_integers():
    n = 257
    a = o.unpack(o.random_field_limbs(n, 1), R)
    b = o.unpack(o.random_field_limbs(n, 2), R)
    a[3] = 0
    va, vb = FV.from_ints(a), FV.from_ints(b)
    s = 0x1234567890ABCDEF1234567890ABCDEF1234567 % R
    assert (va * vb).to_ints() == [x * y % R for x, y in zip(a, b)]
    assert (va + vb).to_ints() == [(x + y) % R for x, y in zip(a, b)]
    assert (va - vb).to_ints() == [(x - y) % R for x, y in zip(a, b)]
    assert (s * va).to_ints() == [s * x % R for x in a] == (va * s).to_ints()
    assert (1 - va).to_ints() == [(1 - x) % R for x in a]
    assert (va - s).to_ints() == [(x - s) % R for x in a]
    assert ((va * vb + s) % R).to_ints() == [(x * y + s) % R for x, y in zip(a, b)]
    assert va.dot(vb) == sum(x * y for x, y in zip(a, b)) % R
    assert va.eval(s) == o.eval_polynomial(a, s)
    assert va.kate_division(s).to_ints() == o.kate_division(a, s) + [0]
    assert va.batch_inv().to_ints() == [pow(x, -1, R) if x else 0 for x in a]
    run = [7]
    for x in a[: n - 1]:
        run.append(run[-1] * x % R)
    assert va.running_product(7, n).to_ints() == run
    assert FV.powers(s, 50, start=9).to_ints() == [9 * pow(s, i, R) % R for i in range(50)]
    assert va.roll(5).to_ints() == [a[(i + 5) % n] for i in range(n)] and va.roll(-2).to_ints() == [a[(i - 2) % n] for i in range(n)]
    assert FF.fv_to_ints(FF.ints_to_fv(a)) == a and FF.ints_to_fv(a).to_ints() == a
    assert FV.from_sparse(8, {1: 5, 6: R - 1}).to_ints() == [0, 5, 0, 0, 0, 0, R - 1, 0]
    assert FV.full(4, 3).to_ints() == [3] * 4


@pytest.mark.parametrize("k,degree", [(5, 3), (6, 5)])
def test_fast_domain_matches_oracle_domain(k, degree):
    d, fd = o.Domain(k, degree), FastDomain(k, degree)
    a = o.unpack(o.random_field_limbs(1 << k, 3), R)
    va = FV.from_ints(a)
    coeff = d.lagrange_to_coeff(a)
    assert fd.lagrange_to_coeff(va).to_ints() == coeff
    ext = d.coeff_to_extended(coeff)
    assert fd.coeff_to_extended(FV.from_ints(coeff)).to_ints() == ext
    e = o.unpack(o.random_field_limbs(1 << d.extended_k, 4), R)
    assert fd.extended_to_coeff(FV.from_ints(e)).to_ints() == d.extended_to_coeff(e)


def _case(shape, k, bits, x):
    if shape == "standard_plonk":
        cs = FX.standard_plonk_cs()
        return cs, FX.standard_plonk_assignment(cs, x)
    cs = FX.flex_gate_cs(shape == "range")
    return cs, (FX.range_assignment(cs, x, bits, 1 << k) if shape == "range" else FX.halo2_lib_assignment(cs, x))


@pytest.mark.parametrize("shape,k,bits,x,seed", [("standard_plonk", 5, 0, 0xDEADBEEF12345, 77), ("standard_plonk", 8, 0, 99, 3),
                                                 ("halo2_lib", 6, 0, 12, 2024), ("range", 7, 4, 0xDEADBEEFCAFE1234, 99),
                                                 ("range", 8, 7, (1 << 64) - 1, 5), ("range", 9, 6, 77, 6)])
def test_fast_prover_bytes_equal_slow_prover(shape, k, bits, x, seed):
    cs, asg = _case(shape, k, bits, x)
    slow_keys = FX.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
    fast_keys = FF.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
    assert fast_keys.vk_bytes() == slow_keys.vk_bytes() and fast_keys.transcript_repr == slow_keys.transcript_repr
    assert [p.to_ints() for p in fast_keys.sigma_polys] == slow_keys.sigma_polys
    slow, fast = FX.prove(slow_keys, asg, seed), FF.prove(fast_keys, asg, seed)
    for name in ("theta", "beta", "gamma", "y", "x"):
        assert slow[name] == fast[name], name
    assert [z.to_ints() for z in fast["zs"]] == slow["zs"]
    assert fast["h_coeffs"].to_ints() == slow["h_coeffs"]
    assert fast["proof"] == slow["proof"]


def test_fast_prover_reproduces_committed_small_goldens():
    """tests/golden/{standard_plonk,flex}_proofs.json were made by the slow oracles (oracle/prover.py, oracle/flex.py)"""
    import sys

    sys.path.insert(0, GOLDEN)
    from make_flex_golden import poseidon_assignment

    g = json.load(open(os.path.join(GOLDEN, "standard_plonk_proofs.json")))
    for case in g["cases"]:
        cs = FX.standard_plonk_cs()
        asg = FX.standard_plonk_assignment(cs, int(case["witness_x"], 16))
        keys = FF.Keys(cs, case["k"], int(g["srs_secret"], 16), asg.fixed, asg.copies)
        assert FF.prove(keys, asg, case["seed"])["proof"].hex() == case["proof"]
    g = json.load(open(os.path.join(GOLDEN, "flex_proofs.json")))
    for case in g["cases"]:
        shape, k, bits, x = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16)
        cs = FX.flex_gate_cs(shape == "range")
        asg = poseidon_assignment(cs, x, x + 1) if shape == "poseidon" else _case(shape, k, bits, x)[1]
        keys = FF.Keys(cs, k, int(g["srs_secret"], 16), asg.fixed, asg.copies)
        assert keys.vk_bytes().hex() == case["vk_bytes"]
        assert FF.prove(keys, asg, case["seed"])["proof"].hex() == case["proof"], (shape, bits)


def test_big_golden_fixture_is_self_consistent():
    """tests/golden/big_proofs.json: every committed proof hashes to its recorded digest, has the shape's length, and the
    oracle verifier accepts it against the closed-form verifying key (cells only: no length-n code), whose bytes equal the
    recorded vk bytes.  (Regenerating the proofs takes minutes: make_big_golden.py, build container only.)"""
    import hashlib
    import sys

    from oracle import formats as fm

    sys.path.insert(0, GOLDEN)
    from make_flex_golden import poseidon_assignment

    g = json.load(open(os.path.join(GOLDEN, "big_proofs.json")))
    secret = int(g["srs_secret"], 16)
    names = {c["name"] for c in g["cases"]}
    assert {"standard_plonk_k16", "standard_plonk_k20", "halo2_lib_k20", "poseidon_k20", "range_k22_bits16"} <= names
    for case in g["cases"]:
        shape, k, bits, x = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16)
        proof = bytes.fromhex(case["proof"])
        assert hashlib.sha256(proof).hexdigest() == case["proof_sha256"]
        assert len(proof) == {"standard_plonk": 992, "halo2_lib": 864, "poseidon": 864, "range": 992}[shape]
        if shape == "poseidon":
            cs = FX.flex_gate_cs(False)
            asg = poseidon_assignment(cs, x, x + 1)
        else:
            cs, asg = _case(shape, k, bits, x)
        vk = FX.VerifierKeys(cs, k, secret, asg.fixed, asg.copies)
        assert case["vk_bytes"][16:] == b"".join(fm.g1_to_bytes(c) for c in vk.fixed_commitments + vk.permutation_commitments).hex()
        assert FX.verify(vk, proof, asg.instance), case["name"]
        bad = bytearray(proof)
        bad[40] ^= 1
        assert not FX.verify(vk, bytes(bad), asg.instance)


def test_wide_golden_fixture_is_self_consistent():
    """tests/golden/flex_wide_proofs.json (round 5: dozens of gate columns, up to eight lookup-advice columns): the oracle verifier
    accepts every committed proof against the closed-form verifying key of the shape the generator records, whose bytes equal the
    recorded vk bytes; a flipped bit is refused.  The vector engine regenerates the proof bytes of the two smallest cases."""
    import sys

    from oracle import formats as fm

    sys.path.insert(0, GOLDEN)
    import make_flex_wide_golden as MW

    g = json.load(open(os.path.join(GOLDEN, "flex_wide_proofs.json")))
    secret = int(g["srs_secret"], 16)
    shapes = {(c["num_advice"], c["num_lookup_advice"]) for c in g["cases"]}
    assert max(a for a, _ in shapes) >= 31 and max(l for _, l in shapes) == 8  # the limits of include/h2mi_prover.h are exercised
    assert any(c["num_fixed"] == 2 for c in g["cases"])  # and two constants columns
    assert {c["shape"] for c in g["cases"]} >= {"halo2_lib+range_builder", "poseidon+range_builder"}  # the Range builder with nothing looked up
    for case in g["cases"]:
        shape, k, bits, x = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16)
        explicit = (case["num_advice"], case["num_lookup_advice"], case["num_fixed"]) if case["explicit"] else None
        cs, asg = MW.build(shape, k, bits, x, case["count"], explicit)
        assert (getattr(cs, "num_advice", 1), getattr(cs, "num_lookup_advice", 0), getattr(cs, "num_fixed", 1)) == (case["num_advice"], case["num_lookup_advice"], case["num_fixed"])
        assert ["0x%x" % v for v in asg.instance[0]] == case["instance"]
        proof = bytes.fromhex(case["proof"])
        vk = FX.VerifierKeys(cs, k, secret, asg.fixed, asg.copies)
        assert case["vk_bytes"][16:] == b"".join(fm.g1_to_bytes(c) for c in vk.fixed_commitments + vk.permutation_commitments).hex()
        assert FX.verify(vk, proof, asg.instance), (shape, k)
        bad = bytearray(proof)
        bad[len(bad) // 2] ^= 1
        assert not FX.verify(vk, bytes(bad), asg.instance)
        if shape == "range" and k <= 6:
            keys = FF.Keys(cs, k, secret, asg.fixed, asg.copies)
            assert FF.prove(keys, asg, case["seed"])["proof"] == proof


def test_range_many_table_and_row_budget():
    """the several-range-checks closure is _range_table repeated at running offsets (identical for one value), and keygen refuses
    what halo2's Assembly refuses: fixed cells or copy constraints beyond the usable rows (NotEnoughRowsAvailable) — LOOKUP_BITS 2 at
    DEGREE 5 needs 32 limb bases in a 25-row constants column."""
    x = 0xDEADBEEFCAFE1234
    for bits in (3, 4, 5, 8):
        a = FX._range_table(x, bits)
        b, pub = FX._range_many_table([x], bits)
        assert (a.rows, a.gates, a.lookups, a.events) == (b.rows, b.gates, b.lookups, b.events) and pub == [0]
    t, pub = FX._range_many_table(FX.range_many_values(x, 3), 4)
    assert len(pub) == 3 and len(t.lookups) == 48 and [t.value(r) for r in pub] == FX.range_many_values(x, 3)
    cs = FX.flex_multi_cs(True, 5, 2)
    asg = FX.range_assignment_multi(cs, x, 2, 5)
    with pytest.raises(ValueError, match="NotEnoughRowsAvailable"):
        FX.Keys(cs, 5, SRS_SECRET, asg.fixed, asg.copies)
    with pytest.raises(ValueError, match="NotEnoughRowsAvailable"):
        FX.VerifierKeys(cs, 5, SRS_SECRET, asg.fixed, asg.copies)
