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
    """examples/poseidon.rs:15-36 on the oracle's cell tables (oracle/flex.py _Table): the sponge of T = 3, RATE = 2 absorbing
    [x, y] laid out from the oracle's own permutation, gate by gate [RECALL snark-verifier's halo2-base Poseidon chip]:
    add-round-constants by gate.add(cell, Constant(c)), x^5 as three gate.mul, the MDS rows as inner products whose first
    coefficient is not one ([0, (cell, Constant(m), sum) ...])"""
    constants, mds = OP.params()
    W, E, K = FX.W_, FX.E_, FX.K_
    t = FX._Table()
    last = lambda base, count: base + count - 1
    xc, yc = t.put([(W, x)]), t.put([(W, y)])
    state = [t.put([(K, v)]) for v in (1 << 64, 0, 0)]

    def add_const(cell, c):
        return last(t.put([(E, cell), (K, c), (K, 1), (W, t.value(cell) + c)], [0]), 4)

    def add(a, b):
        return last(t.put([(E, a), (E, b), (K, 1), (W, t.value(a) + t.value(b))], [0]), 4)

    def mul(a, b):
        return last(t.put([(K, 0), (E, a), (E, b), (W, t.value(a) * t.value(b))], [0]), 4)

    def inner(cells, coeffs):
        items, acc = [(K, 0)], 0
        for cell, c in zip(cells, coeffs):
            acc = (acc + t.value(cell) * c) % R
            items += [(E, cell), (K, c), (W, acc)]
        return last(t.put(items, [3 * j for j in range(len(cells))]), len(items))

    def permute(st):
        for rnd in range(8 + 57):
            s = [add_const(cell, c) for cell, c in zip(st, constants[rnd])]
            for i in (range(3) if rnd < 4 or rnd >= 4 + 57 else range(1)):
                x2 = mul(s[i], s[i])
                x4 = mul(x2, x2)
                s[i] = mul(x4, s[i])
            st = [inner(s, row) for row in mds]
        return st

    state[1] = add(state[1], xc)
    state[2] = add(state[2], yc)
    state = permute(state)
    state[1] = add_const(state[1], 1)  # the empty chunk after an exact multiple of RATE: padding only
    state = permute(state)
    asg = t.assignment(cs, [xc, yc, state[1]])
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
