#!/usr/bin/env python3
"""Generates tests/golden/flex_proofs.json from the oracle engine (oracle/flex.py, oracle/poseidon.py).

SELF-DERIVED vectors (the reference holds no proof bytes, its prover draws from OsRng and halo2-base is not vendored;
SURVEY.md 0, 8c): proofs of the reference's halo2-lib example closures — examples/halo2_lib.rs (x^2 + 72), examples/range.rs
(range_check(x, 64) with LOOKUP_BITS 4 / 7 / 6: no remainder, one-bit and four-bit top limb) and examples/poseidon.rs
(hash_two) — for a fixed SRS secret, witness and rng seed, computed in plain Python integers.  The poseidon witness is laid
out here from the oracle's own permutation (round by round: add constants, x^5 on the s-box lanes, MDS inner products),
independently of the product's chip.  They pin the oracle engine, its verifier and both device hosts (Python, C++) to each
other across rounds.
Usage: python tests/golden/make_flex_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flex as FX  # noqa: E402
from oracle import poseidon as OP  # noqa: E402

SRS_SECRET = 0x5EC2E7 + 0x48324D49
R = FX.R


def poseidon_assignment(cs, x, y):
    """examples/poseidon.rs:15-36 on the oracle's _Ctx: the same cell layout as halo2-scaffold_amd/poseidon.py, written
    against oracle/poseidon.py's parameters"""
    constants, mds = OP.params()
    asg = FX.Assignment(cs)
    ctx = FX._Ctx(asg)
    xc, yc = ctx.load_witness(x), ctx.load_witness(y)
    state = [ctx.assign_region_last([("constant", v)], []) for v in (1 << 64, 0, 0)]

    def inner(cells, coeffs):
        items, gates, acc = [("constant", 0)], [], 0
        for k, (cell, c) in enumerate(zip(cells, coeffs)):
            acc = (acc + ctx.cells[cell] * c) % R
            items += [("existing", cell), ("constant", c), ("witness", acc)]
            gates.append(3 * k)
        return ctx.assign_region_last(items, gates)

    def permute(st):
        for rnd in range(8 + 57):
            s = [ctx.add_const(cell, c) for cell, c in zip(st, constants[rnd])]
            for i in (range(3) if rnd < 4 or rnd >= 4 + 57 else range(1)):
                x2 = ctx.mul(s[i], s[i])
                x4 = ctx.mul(x2, x2)
                s[i] = ctx.mul(x4, s[i])
            st = [inner(s, row) for row in mds]
        return st

    state[1] = ctx.add(state[1], xc)
    state[2] = ctx.add(state[2], yc)
    state = permute(state)
    state[1] = ctx.add_const(state[1], 1)  # the empty chunk after an exact multiple of RATE: padding only
    state = permute(state)
    ctx.finish([xc, yc, state[1]])
    assert asg.instance[0][2] == OP.sponge_hash([x, y])
    return asg


def main():
    out = {"srs_secret": "0x%x" % SRS_SECRET, "cases": []}
    cases = [("halo2_lib", 6, 0, 12, 2024), ("range", 7, 4, 0xDEADBEEFCAFE1234, 99), ("range", 8, 7, (1 << 64) - 1, 5), ("range", 8, 6, 77, 6),
             ("poseidon", 13, 0, 0xFEEDFACE, 17)]
    for shape, k, bits, x, seed in cases:
        cs = FX.flex_gate_cs(shape == "range")
        if shape == "range":
            asg = FX.range_assignment(cs, x, bits, 1 << k)
        elif shape == "poseidon":
            asg = poseidon_assignment(cs, x, x + 1)
        else:
            asg = FX.halo2_lib_assignment(cs, x)
        keys = FX.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
        r = FX.prove(keys, asg, seed)
        assert FX.verify(keys, r["proof"], asg.instance)
        out["cases"].append({"shape": shape, "k": k, "lookup_bits": bits, "x": "0x%x" % x, "seed": seed, "instance": ["0x%x" % v for v in asg.instance[0]],
                             "vk_bytes": keys.vk_bytes().hex(), "proof": r["proof"].hex()})
        print(shape, k, bits, len(r["proof"]))
    with open(os.path.join(HERE, "flex_proofs.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
