#!/usr/bin/env python3
"""Generates tests/golden/flex_multi_proofs.json from the oracle engine (oracle/flex.py; oracle/fastflex.py for the poseidon case).

SELF-DERIVED vectors (the reference holds no proof bytes and halo2-base is not vendored).  Round 4: the shapes
`builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268) configures when a closure's cells do NOT fit one advice column at the
chosen DEGREE — several gate columns with their own selectors, lookup-advice columns with one lookup argument each
(oracle/flex.py flex_multi_cs, multi_column_assignment [RECALL halo2-base 0.3]):
  range_check(x, 64), LOOKUP_BITS 4 at DEGREE 5 (51 cells over 23-row columns: 3 gate columns + 1 lookup-advice column; two break
  copies), LOOKUP_BITS 3 at DEGREE 6 (2 + 1), poseidon
  hash_two at DEGREE 11 (7.4 k cells: 4 gate columns).
Usage: python tests/golden/make_flex_multi_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import fastflex as FF  # noqa: E402
from oracle import flex as FX  # noqa: E402

SRS_SECRET = 0x5EC2E7 + 0x48324D49
CASES = [("range", 5, 4, 0xDEADBEEFCAFE1234, 99), ("range", 6, 3, 0x0123456789ABCDEF, 5), ("poseidon", 11, 0, 0xFEEDFACE, 17)]


def build(shape, k, bits, x):
    """-> (constraint system, assignment) with the column counts GateThreadBuilder::config takes"""
    if shape == "range":
        t = FX._range_table(x, bits)
    elif shape == "halo2_lib":
        t = FX._halo2_lib_table(x)
    else:
        import make_flex_golden as MG

        probe = FX.flex_gate_cs(False)
        box = {}
        orig = FX._Table.assignment
        FX._Table.assignment = lambda self, cs, pub: box.update(t=self, pub=pub) or orig(self, cs, pub)  # catch the table the generator lays out
        try:
            MG.poseidon_assignment(probe, x, x + 1)
        finally:
            FX._Table.assignment = orig
        t, pub = box["t"], box["pub"]
    A, Lc = FX.multi_column_counts(len(t.rows), len(t.lookups), k)
    assert A >= 2
    cs = FX.flex_multi_cs(shape == "range", A, Lc)
    if shape == "range":
        asg = FX.range_assignment_multi(cs, x, bits, k)
    elif shape == "halo2_lib":
        asg = FX.halo2_lib_assignment_multi(cs, x, k)
    else:
        asg = FX.multi_column_assignment(t, cs, pub, k)
    return cs, asg


def main():
    out = {"srs_secret": "0x%x" % SRS_SECRET, "cases": []}
    for shape, k, bits, x, seed in CASES:
        cs, asg = build(shape, k, bits, x)
        if k <= 8:
            keys = FX.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
            r = FX.prove(keys, asg, seed)
            assert FX.verify(keys, r["proof"], asg.instance)
        else:  # the vector form of the same prover (byte-identical where both finish: tests/test_oracle_fast.py)
            keys = FF.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
            r = FF.prove(keys, asg, seed)
            vk = FX.VerifierKeys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
            assert FX.verify(vk, r["proof"], asg.instance)
        out["cases"].append({"shape": shape, "k": k, "lookup_bits": bits, "x": "0x%x" % x, "seed": seed, "num_advice": cs.num_advice,
                             "num_lookup_advice": cs.num_lookup_advice, "instance": ["0x%x" % v for v in asg.instance[0]],
                             "vk_bytes": keys.vk_bytes().hex(), "proof": r["proof"].hex()})
        print(shape, k, bits, cs.num_advice, cs.num_lookup_advice, len(r["proof"]), flush=True)
    with open(os.path.join(HERE, "flex_multi_proofs.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
