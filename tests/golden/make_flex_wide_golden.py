#!/usr/bin/env python3
"""Generates tests/golden/flex_wide_proofs.json from the oracle engine (oracle/flex.py, oracle/fastflex.py).

SELF-DERIVED vectors (the reference holds no proof bytes and halo2-base is not vendored).  Round 5: the halo2-lib
builders' constraint systems with MANY columns — what `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268) takes for a
circuit whose cells fill dozens of columns at the chosen DEGREE, and what a caller gets who sets the column counts by hand
(halo2-base's FlexGateConfigParams carry num_advice / num_lookup_advice explicitly):
  poseidon hash_two at DEGREE 8 (7.4 k cells over 247-row columns: 31 gate columns, 33 permutation columns, 124 advice queries),
  at DEGREE 9 (15 gate columns),
  `count` range checks in one context (oracle/flex.py _range_many_table: examples/range.rs's body once per value, the limb bases
  shared so that the constants still fit one fixed column), LOOKUP_BITS 4: 8 checks at DEGREE 6 (8 gate + 3 lookup-advice columns),
  24 checks at DEGREE 7 (11 + 4), 10 checks at DEGREE 6 with 11 + 8 columns set explicitly (eight lookup arguments, five of them
  over empty columns).  A configuration whose constants overflow the usable rows of the one constants column is refused by keygen
  (halo2's NotEnoughRowsAvailable), e.g. LOOKUP_BITS 2 at DEGREE 5: 32 limb bases for 25 usable rows, where config's
  ceil(constants / 2^k) still says one column; with num_fixed = 2 set by hand (the last case: 5 + 2 columns, two constants columns,
  the constants dealt out round-robin) it proves.
  The Range builder under closures that look nothing up (the reference takes it whenever LOOKUP_BITS is set: src/scaffold.rs:44-48):
  halo2_lib at DEGREE 6 with LOOKUP_BITS 4 (one column: the q_lookup selector stays empty) and poseidon at DEGREE 11 with LOOKUP_BITS 8
  (4 gate columns, NO lookup-advice column and no lookup argument: degree 3, the table column committed but never queried).
Usage: python tests/golden/make_flex_wide_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_flex_multi_golden as MM  # noqa: E402
from oracle import fastflex as FF  # noqa: E402
from oracle import flex as FX  # noqa: E402

SRS_SECRET = MM.SRS_SECRET
# (shape, k, lookup bits, x, seed, range checks, explicit (num_advice, num_lookup_advice[, num_fixed]) or None)
CASES = [("poseidon", 8, 0, 0xFEEDFACE, 17, 0, None), ("poseidon", 9, 0, 0xABCDEF, 23, 0, None),
         ("range", 6, 4, 0xDEADBEEFCAFE1234, 31, 8, None), ("range", 6, 4, 0x0123456789ABCDEF, 41, 10, (11, 8)),
         ("range", 7, 4, 0x0F1E2D3C4B5A6978, 43, 24, None), ("range", 5, 2, 0xDEADBEEFCAFE1234, 47, 1, (5, 2, 2)),
         ("halo2_lib+range_builder", 6, 4, 12, 5, 0, None), ("poseidon+range_builder", 11, 8, 0xFEED, 9, 0, None)]


def _closure_table(shape, x):
    """-> (table, public rows) of the halo2_lib / poseidon closure"""
    if shape == "halo2_lib":
        return FX._halo2_lib_table(x), [0, 8]
    import make_flex_golden as MG

    probe = FX.flex_gate_cs(False)
    box = {}
    orig = FX._Table.assignment
    FX._Table.assignment = lambda self, cs, pub: box.update(t=self, pub=pub) or orig(self, cs, pub)  # catch the table the generator lays out
    try:
        MG.poseidon_assignment(probe, x, x + 1)
    finally:
        FX._Table.assignment = orig
    return box["t"], box["pub"]


def build(shape, k, bits, x, count, explicit):
    if shape.endswith("+range_builder"):  # a closure without range checks under the Range builder: the table is loaded all the same
        t, pub = _closure_table(shape.split("+")[0], x)
        A, Lc = FX.multi_column_counts(len(t.rows), len(t.lookups), k)
        assert Lc == 0
        if A == 1:
            cs = FX.flex_gate_cs(True)
            asg = t.assignment(cs, pub)
        else:
            cs = FX.flex_multi_cs(True, A, 0, FX.num_fixed_columns(t, k))
            asg = FX.multi_column_assignment(t, cs, pub, k)
        asg.fixed[cs.col_table] = {i: i for i in range(1 << bits)}
        return cs, asg
    if shape != "range":
        return MM.build(shape, k, bits, x)
    t, _ = FX._range_many_table(FX.range_many_values(x, count), bits)
    A, Lc = explicit[:2] if explicit else FX.multi_column_counts(len(t.rows), len(t.lookups), k)
    Fc = explicit[2] if explicit and len(explicit) > 2 else FX.num_fixed_columns(t, k)
    cs = FX.flex_multi_cs(True, A, Lc, Fc)
    return cs, FX.range_many_assignment_multi(cs, x, bits, k, count)


def main():
    out = {"srs_secret": "0x%x" % SRS_SECRET, "cases": []}
    for shape, k, bits, x, seed, count, explicit in CASES:
        cs, asg = build(shape, k, bits, x, count, explicit)
        keys = FF.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
        r = FF.prove(keys, asg, seed)
        vk = FX.VerifierKeys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
        assert FX.verify(vk, r["proof"], asg.instance)
        if k <= 6:  # the big-integer engine, where it finishes in seconds: the same bytes
            bkeys = FX.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
            assert FX.prove(bkeys, asg, seed)["proof"] == r["proof"] and bkeys.vk_bytes() == keys.vk_bytes()
        out["cases"].append({"shape": shape, "k": k, "lookup_bits": bits, "x": "0x%x" % x, "seed": seed, "count": count, "explicit": explicit is not None,
                             "num_advice": getattr(cs, "num_advice", 1), "num_lookup_advice": getattr(cs, "num_lookup_advice", 0), "num_fixed": getattr(cs, "num_fixed", 1),
                             "instance": ["0x%x" % v for v in asg.instance[0]], "vk_bytes": keys.vk_bytes().hex(), "proof": r["proof"].hex()})
        print(shape, k, bits, getattr(cs, "num_advice", 1), getattr(cs, "num_lookup_advice", 0), len(r["proof"]), flush=True)
    with open(os.path.join(HERE, "flex_wide_proofs.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
