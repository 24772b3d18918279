"""CPU tests of oracle/poseidon.py (the hash of the reference's examples/poseidon.rs) against PUBLISHED values — round
constants and MDS entries from circomlib's poseidon_constants and circomlibjs' known answer, committed with their
provenance in tests/golden/poseidon_vectors.json — and of the product's witness generator against the oracle."""
import json
import os

import _load_pkg
from oracle import bn254 as o
from oracle import poseidon as OP

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_vectors.json")))


def test_grain_constants_and_mds_match_published_values():
    constants, mds = OP.generate(G["t"], G["r_f"], G["r_p"])
    assert len(constants) == 65 and all(len(r) == 3 for r in constants)
    assert constants[0] == [int(v, 16) for v in G["round_constants_first_row"]]
    assert mds[0] == [int(v, 16) for v in G["mds_first_row"]]
    assert all(0 <= v < o.R for row in constants + mds for v in row)


def test_permutation_known_answer():
    assert OP.circomlib_hash([1, 2]) == int(G["circomlib_hash_1_2"], 16)


def test_sponge_conventions():
    """exact multiple of RATE: a second permutation absorbs only the padding; shorter input: padded in place"""
    two = OP.sponge_hash([5, 7])
    st = OP.permute([(1 << 64), 5, 7])
    st[1] = (st[1] + 1) % o.R
    assert two == OP.permute(st)[1]
    one = OP.sponge_hash([5])
    assert one == OP.permute([(1 << 64), 5, 1])[1]
    assert len({two, one, OP.sponge_hash([7, 5])}) == 3


def test_product_witness_generator_matches_oracle():
    """halo2-scaffold_amd/poseidon.py: same parameters from its own LFSR; the circuit's public output is the oracle's
    hash; every enabled row satisfies the vertical gate and every copy constraint joins equal cells"""
    _load_pkg.load()
    from halo2_scaffold_amd import flex, poseidon

    assert poseidon.spec() == OP.params()
    cs = flex.FlexGateCS(lookup=False)
    x, y = 0x1234567890ABCDEF, o.R - 5
    asg = poseidon.hash_two_closure(cs, x, y)
    assert asg.instance == [x, y, OP.sponge_hash([x, y])]
    cells = asg.advice[0]
    assert 7000 < len(cells) < 8000
    for r in asg.fixed[cs.col_q]:
        assert (cells[r] + cells[r + 1] * cells[r + 2] - cells[r + 3]) % o.R == 0, r
    value = {"advice": lambda c, r: asg.advice[c][r], "fixed": lambda c, r: asg.fixed[c][r], "instance": lambda c, r: asg.instance[r]}
    for (k1, c1, r1), (k2, c2, r2) in asg.copies:
        assert value[k1](c1, r1) == value[k2](c2, r2)


def test_mock_accepts_the_example_closures_and_names_violations():
    """flex.mock = scaffold::mock for the halo2-lib shapes (host-side constraint check before proving)"""
    import pytest

    _load_pkg.load()
    from halo2_scaffold_amd import flex, poseidon

    gate, rng = flex.FlexGateCS(lookup=False), flex.FlexGateCS(lookup=True)
    for asg in (flex.halo2_lib_closure(gate, 7), poseidon.hash_two_closure(gate, 1, 2), flex.range_closure(rng, 0xFFFF0000FFFF, 8),
                flex.range_closure(rng, (1 << 64) - 1, 7)):
        flex.mock(asg)
    bad = flex.halo2_lib_closure(gate, 7)
    bad.advice[0][4] += 1  # the product cell of x * x
    with pytest.raises(ValueError, match="gate not satisfied at row 1"):
        flex.mock(bad)
    bad = flex.halo2_lib_closure(gate, 7)
    bad.instance[1] += 1
    with pytest.raises(ValueError, match="copy constraint"):
        flex.mock(bad)
    bad = flex.range_closure(rng, 12345, 4)
    row = sorted(bad.fixed[rng.col_qlookup])[0]
    bad.advice[0][row] = 16
    with pytest.raises(ValueError):  # the limb no longer recomposes (gate) — and is outside the table
        flex.mock(bad)
    # MockProver::run's row budget: 32 limb bases (LOOKUP_BITS 2) in the 25 usable rows of a DEGREE-5 constants column, one column by
    # config's own count; a second constants column set by hand holds them; a one-column circuit longer than its DEGREE allows
    cs = flex.configure(True, 5, lambda c: flex.range_closure(c, 0xDEADBEEFCAFE1234, 2))
    assert (cs.num_advice, cs.num_lookup_advice, cs.num_fixed) == (5, 2, 1)
    with pytest.raises(ValueError, match="NotEnoughRowsAvailable: fixed column"):
        flex.mock(flex.range_closure(cs, 0xDEADBEEFCAFE1234, 2))
    cs2 = flex.FlexGateCS(True, 5, 2, k=5, num_fixed=2)
    flex.mock(flex.range_closure(cs2, 0xDEADBEEFCAFE1234, 2))
    flex.mock(flex.range_closure(rng, 12345, 4), k=7)
    with pytest.raises(ValueError, match="NotEnoughRowsAvailable"):  # 51 cells (and their selector rows) in 25 usable rows
        flex.mock(flex.range_closure(rng, 12345, 4), k=5)


def test_scaffold_mock_runs_on_the_host(monkeypatch):
    """scaffold.mock (the reference's src/scaffold.rs:39-93, MockProver only) needs no GPU: configure from the environment, run the
    closure, check rows / gates / copies / lookups.  gen_key / prove_private / prove need the device (tests/test_gpu_flex.py)."""
    import pytest

    _load_pkg.load()
    from halo2_scaffold_amd import scaffold

    def range_example(ctx, x, make_public):  # examples/range.rs:10-34
        xc = ctx.load_witness(x)
        make_public.append(xc)
        ctx.range_check(xc, 64, ctx.lookup_bits)
        ctx.add(xc, xc)

    monkeypatch.setenv("DEGREE", "5")
    monkeypatch.setenv("LOOKUP_BITS", "4")
    monkeypatch.delenv("MINIMUM_ROWS", raising=False)
    scaffold.mock(range_example, 0xDEADBEEFCAFE1234)  # 3 gate + 1 lookup-advice columns
    monkeypatch.setenv("LOOKUP_BITS", "2")           # 32 limb bases in a 25-row constants column
    with pytest.raises(ValueError, match="NotEnoughRowsAvailable"):
        scaffold.mock(range_example, 0xDEADBEEFCAFE1234)
    monkeypatch.setenv("DEGREE", "4")
    monkeypatch.setenv("LOOKUP_BITS", "3")
    with pytest.raises(ValueError, match="NOT ENOUGH ADVICE COLUMNS"):  # the column count of config's formula is too small here, as in halo2-base
        scaffold.mock(range_example, 5)
