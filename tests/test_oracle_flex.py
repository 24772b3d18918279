"""CPU tests of oracle/flex.py, the data-driven restatement of keygen / create_proof / verify_proof that checks the
device provers of the halo2-lib shapes (tests/test_gpu_flex.py).  The engine is pinned three ways: run on the
StandardPlonk constraint system it must reproduce the committed golden proofs of the dedicated oracle byte for byte;
its verifier accepts its proofs and rejects tampered ones / other public inputs (with the pairing too); and its
closed-form verifying key equals the one built from full columns."""
import json
import os

import pytest

from oracle import flex as FX
from oracle import formats as fm

R = FX.R
SECRET = 0x5EC2E7 + 0x48324D49

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "standard_plonk_proofs.json")


def test_engine_reproduces_standard_plonk_golden_proofs():
    g = json.load(open(GOLDEN))
    s = int(g["srs_secret"], 16)
    case = next(c for c in g["cases"] if c["k"] == 5)  # the reference's own size (examples/standard_plonk.rs:26)
    cs = FX.standard_plonk_cs()
    asg = FX.standard_plonk_assignment(cs, int(case["witness_x"], 16) if isinstance(case["witness_x"], str) else case["witness_x"])
    keys = FX.Keys(cs, 5, s, asg.fixed, asg.copies)
    assert keys.vk_bytes().hex() == case["vk_bytes"]
    out = FX.prove(keys, asg, case["seed"])
    assert out["proof"].hex() == case["proof"]
    assert FX.verify(keys, out["proof"], [])


def test_halo2_lib_prove_verify():
    """examples/halo2_lib.rs: public inputs [x, x^2 + 72]; 27 proof elements (1 advice, 3 z, random, 2 h pieces, 6 + 2 +
    1 + 3 + 8 evaluations... = 864 bytes)"""
    x, k, s = 12, 6, 0xABCDEF123
    cs = FX.flex_gate_cs(False)
    asg = FX.halo2_lib_assignment(cs, x)
    assert asg.instance == [[12, 216]]
    assert sorted(asg.fixed[cs.col_q]) == [1, 5, 9, 13]  # the four gates: mul, add, the raw region, mul_add
    assert asg.fixed[cs.col_const] == {0: 0, 1: 72, 2: 1}   # constants in order of first use
    keys = FX.Keys(cs, k, s, asg.fixed, asg.copies)
    proof = FX.prove(keys, asg, 11)["proof"]
    assert len(proof) == 864
    assert FX.verify(keys, proof, asg.instance)
    assert not FX.verify(keys, proof, [[12, 217]])
    assert not FX.verify(keys, proof[:-1], asg.instance) and not FX.verify(keys, proof + b"\0", asg.instance)
    for pos in (3, 40, 300, 700, 863):
        bad = bytearray(proof)
        bad[pos] ^= 4
        assert not FX.verify(keys, bytes(bad), asg.instance), pos
    # a witness that breaks the gate: no valid proof comes out
    asg.advice[0][4] = (asg.advice[0][4] + 1) % FX.R
    assert not FX.verify(keys, FX.prove(keys, asg, 11)["proof"], asg.instance)


@pytest.mark.parametrize("lookup_bits,x", [(4, 0xDEADBEEFCAFE1234), (7, (1 << 64) - 1)])
def test_range_prove_verify(lookup_bits, x):
    """examples/range.rs with LOOKUP_BITS in place of the reference's (src/scaffold.rs:44-48 reads it from the
    environment): 64 / 4 = 16 limbs, or ten 7-bit limbs whose one-bit top limb is constrained by assert_bit"""
    k, s = 8, 0x77665544
    cs = FX.flex_gate_cs(True)
    asg = FX.range_assignment(cs, x, lookup_bits, 1 << k)
    limbs = -(-64 // lookup_bits)
    looked_up = sorted(asg.fixed[cs.col_qlookup])  # single advice column: q_lookup on the cells' own rows
    assert len(looked_up) == limbs + (1 if 64 % lookup_bits > 1 else 0) and len(asg.advice) == 1  # a one-bit top limb: assert_bit instead
    assert all(asg.advice[0][r] < 1 << lookup_bits for r in looked_up)
    assert cs.degree == 5 and cs.chunk == 3
    keys = FX.Keys(cs, k, s, asg.fixed, asg.copies)
    vk = FX.VerifierKeys(cs, k, s, asg.fixed, asg.copies)
    assert vk.fixed_commitments == keys.fixed_commitments and vk.permutation_commitments == keys.permutation_commitments
    assert vk.transcript_repr == keys.transcript_repr
    proof = FX.prove(keys, asg, 5)["proof"]
    assert len(proof) == 992  # 12 commitments (1 advice, A', S', z, lookup z, random, 4 h pieces, 2 SHPLONK) + 19 evaluations
    assert FX.verify(vk, proof, asg.instance)
    assert not FX.verify(vk, proof, [[(x + 1) % (1 << 64)]])
    bad = bytearray(proof)
    bad[900] ^= 1
    assert not FX.verify(vk, bytes(bad), asg.instance)
    # a limb outside the table: permute_expression_pair refuses (the crate returns ConstraintSystemFailure)
    asg.advice[0][looked_up[0]] = 1 << lookup_bits
    with pytest.raises(Exception):
        FX.prove(keys, asg, 5)


def test_verify_with_pairing():
    """the final SHPLONK equation through the SRS's G2 elements (what the reference's verifier does) agrees with the
    known-secret form"""
    k, s = 6, 0x1234567
    cs = FX.flex_gate_cs(False)
    asg = FX.halo2_lib_assignment(cs, 3)
    keys = FX.Keys(cs, k, s, asg.fixed, asg.copies)
    proof = FX.prove(keys, asg, 1)["proof"]
    g2, s_g2 = fm.G2_GEN, fm.g2_mul(s)
    assert FX.verify(keys, proof, asg.instance, g2=g2, s_g2=s_g2)
    assert not FX.verify(keys, proof, asg.instance, g2=g2, s_g2=fm.g2_mul(s + 1))


def test_committed_flex_golden_proofs_are_what_the_oracle_produces():
    """tests/golden/flex_proofs.json (tests/golden/make_flex_golden.py): the committed proofs of the halo2-lib example
    closures still come out of the oracle engine bit for bit and still verify — the fixture both device hosts are held to"""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flex_proofs.json")))
    s = int(g["srs_secret"], 16)
    for case in g["cases"]:
        if case["shape"] == "poseidon":
            continue  # regenerated by the script (4 s); its verification below covers the committed bytes
        cs = FX.flex_gate_cs(case["shape"] == "range")
        x = int(case["x"], 16)
        asg = FX.range_assignment(cs, x, case["lookup_bits"], 1 << case["k"]) if case["shape"] == "range" else FX.halo2_lib_assignment(cs, x)
        keys = FX.Keys(cs, case["k"], s, asg.fixed, asg.copies)
        assert keys.vk_bytes().hex() == case["vk_bytes"]
        assert FX.prove(keys, asg, case["seed"])["proof"].hex() == case["proof"]
        assert [int(v, 16) for v in case["instance"]] == asg.instance[0]
        assert FX.verify(keys, bytes.fromhex(case["proof"]), asg.instance)


# ---- round 4: more than one gate column (what builder.config(k, Some(minimum_rows)) configures on overflow: src/scaffold.rs:268) ----
def test_multi_column_layout_and_proofs():
    """range_check(x, 64) with LOOKUP_BITS 4 at DEGREE 5: 51 cells over columns of 2^5 - 9 = 23 rows -> three gate columns and one
    lookup-advice column (oracle/flex.py flex_multi_cs / multi_column_assignment [RECALL halo2-base]).  The layout's own invariants
    (every gate holds on its column, every equality joins equal cells — break copies included —, every looked-up cell is in the
    table), the oracle prover / verifier on it, a tampered proof, a wrong public input, and the committed golden bytes."""
    import json
    import os

    x, bits, k = 0xDEADBEEFCAFE1234, 4, 5
    t = FX._range_table(x, bits)
    assert FX.multi_column_counts(len(t.rows), len(t.lookups), k) == (3, 1) and (len(t.rows), len(t.lookups)) == (51, 16)
    cs = FX.flex_multi_cs(True, 3, 1)
    asg = FX.range_assignment_multi(cs, x, bits, k)
    assert [len(c) for c in asg.advice] == [23, 22, 8, 16]  # two break copies: 51 + 2 cells over the gate columns
    assert cs.degree == 4 and cs.chunk == 2 and len(cs.perm_columns) == 6 and len(cs.lookups) == 1
    # the layout is a satisfying assignment
    for j in range(3):
        a = asg.advice[j]
        for r in asg.fixed[cs.col_q[j]]:
            assert r + 3 < 23 and (a[r] + a[r + 1] * a[r + 2] - a[r + 3]) % R == 0, (j, r)
    val = {FX.ADVICE: lambda c, r: asg.advice[c][r], FX.FIXED: lambda c, r: asg.fixed[c][r], FX.INSTANCE: lambda c, r: asg.instance[c][r]}
    for left, right in asg.copies:
        assert val[left[0]](left[1], left[2]) == val[right[0]](right[1], right[2]), (left, right)
    assert [(l[1], l[2], r[1]) for l, r in asg.copies[:2]] == [(1, 0, 0), (2, 0, 1)]  # the two break copies come first: row 0 of the next column
    assert all(v < 16 for v in asg.advice[3].values())
    keys = FX.Keys(cs, k, SECRET, asg.fixed, asg.copies)
    r = FX.prove(keys, asg, 99)
    assert FX.verify(keys, r["proof"], asg.instance)
    assert FX.verify(FX.VerifierKeys(cs, k, SECRET, asg.fixed, asg.copies), r["proof"], asg.instance)  # closed-form verifying key
    bad = bytearray(r["proof"])
    bad[100] ^= 1
    assert not FX.verify(keys, bytes(bad), asg.instance)
    assert not FX.verify(keys, r["proof"], [[x ^ 1]])
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flex_multi_proofs.json")))
    case = next(c for c in g["cases"] if c["shape"] == "range" and c["k"] == 5)
    assert int(g["srs_secret"], 16) == SECRET and r["proof"].hex() == case["proof"] and keys.vk_bytes().hex() == case["vk_bytes"]
    # a witness that breaks a gate on the SECOND column is not provable
    asg.advice[1][5] = (asg.advice[1][5] + 1) % R
    assert not FX.verify(keys, FX.prove(keys, asg, 99)["proof"], asg.instance)


def test_multi_column_counts_can_be_too_few_as_in_the_crate():
    """GateThreadBuilder::config takes ceil(cells / rows) columns, but a break wastes rows (a gate never straddles two columns): the
    17 cells of x^2 + 72 at DEGREE 4 (7-row columns) are configured with three columns and need four — halo2-base panics with
    'NOT ENOUGH ADVICE COLUMNS' there, and so does the layout here."""
    t = FX._halo2_lib_table(12)
    assert FX.multi_column_counts(len(t.rows), 0, 4) == (3, 0)
    with pytest.raises(ValueError, match="NOT ENOUGH ADVICE COLUMNS"):
        FX.halo2_lib_assignment_multi(FX.flex_multi_cs(False, 3, 0), 12, 4)
    asg = FX.halo2_lib_assignment_multi(FX.flex_multi_cs(False, 4, 0), 12, 4)  # with the fourth column it fits
    assert sum(len(c) for c in asg.advice) == 17 + 3


def test_fast_prover_matches_on_multi_column_shapes():
    """oracle/fastflex.py (the vector form that makes the poseidon DEGREE 11 golden) against the Python-integer engine on the
    three-column range circuit: byte-identical proofs"""
    from oracle import fastflex as FF

    x, bits, k = 0x0123456789ABCDEF, 3, 6
    t = FX._range_table(x, bits)
    A, Lc = FX.multi_column_counts(len(t.rows), len(t.lookups), k)
    assert (A, Lc) == (2, 1)
    cs = FX.flex_multi_cs(True, A, Lc)
    asg = FX.range_assignment_multi(cs, x, bits, k)
    slow = FX.prove(FX.Keys(cs, k, SECRET, asg.fixed, asg.copies), asg, 5)["proof"]
    fast = FF.prove(FF.Keys(cs, k, SECRET, asg.fixed, asg.copies), asg, 5)["proof"]
    assert slow == fast
