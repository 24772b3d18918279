#!/usr/bin/env python3
"""Generates tests/golden/big_proofs.json: golden proof BYTES at the sizes BASELINE.json names, from the large-size oracle
prover (oracle/fastflex.py — oracle/flex.py's prover statement for statement over C vector loops; byte-identical to the
Python-integer oracle wherever that one finishes: tests/test_oracle_fast.py).

SELF-DERIVED vectors (the reference holds no proof bytes, its prover draws from OsRng and the crates are not vendored;
SURVEY.md 0, 8c).  Cases (reference call sites examples/standard_plonk.rs:41-50, src/scaffold.rs:322-331):
  standard_plonk   DEGREE 16 (BASELINE configs[1]) and 20 (the headline / north-star size)
  halo2_lib        DEGREE 20 (configs[2]; examples/halo2_lib.rs:14-60)
  poseidon         DEGREE 20 (configs[4]; examples/poseidon.rs:15-36), cells laid out by make_flex_golden.poseidon_assignment
  range            LOOKUP_BITS 12 at DEGREE 16 and LOOKUP_BITS 16 at DEGREE 22 (configs[3]; examples/range.rs:10-34)
Every proof is accepted by the oracle verifier against the CLOSED-FORM verifying key (oracle/flex.py VerifierKeys: no
length-n code involved) before it is written.  Runs in the build container only (minutes; ~20 GB of memory at DEGREE 22).
Usage: python tests/golden/make_big_golden.py [case-name ...]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import fastflex as FF  # noqa: E402
from oracle import flex as FX  # noqa: E402
from make_flex_golden import poseidon_assignment  # noqa: E402

SRS_SECRET = 0x5EC2E7 + 0x48324D49
OUT = os.path.join(HERE, "big_proofs.json")
# name, shape, k, lookup_bits, x, seed
CASES = [
    ("standard_plonk_k16", "standard_plonk", 16, 0, 0x1234567890ABCDEF1234567890ABCDEF, 2024),
    ("standard_plonk_k20", "standard_plonk", 20, 0, 0xDEADBEEF12345, 77),
    ("halo2_lib_k20", "halo2_lib", 20, 0, 0xC0FFEE, 4242),
    ("poseidon_k20", "poseidon", 20, 0, 0xFEEDFACE, 17),
    ("range_k16_bits12", "range", 16, 12, 0x0123456789ABCDEF, 31337),
    ("range_k22_bits16", "range", 22, 16, 0xFEDCBA9876543210, 99),
]


def assignment(shape, k, bits, x):
    if shape == "standard_plonk":
        cs = FX.standard_plonk_cs()
        return cs, FX.standard_plonk_assignment(cs, x)
    cs = FX.flex_gate_cs(shape == "range")
    if shape == "range":
        return cs, FX.range_assignment(cs, x, bits, 1 << k)
    if shape == "poseidon":
        return cs, poseidon_assignment(cs, x, x + 1)
    return cs, FX.halo2_lib_assignment(cs, x)


def main():
    want = set(sys.argv[1:])
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {"srs_secret": "0x%x" % SRS_SECRET, "cases": []}
    done = {c["name"]: c for c in doc["cases"]}
    for name, shape, k, bits, x, seed in CASES:
        if want and name not in want:
            continue
        t0 = time.time()
        cs, asg = assignment(shape, k, bits, x)
        keys = FF.Keys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
        r = FF.prove(keys, asg, seed)
        vk = FX.VerifierKeys(cs, k, SRS_SECRET, asg.fixed, asg.copies)
        assert vk.fixed_commitments + vk.permutation_commitments == keys.fixed_commitments + keys.permutation_commitments, "closed-form vk differs"
        assert FX.verify(vk, r["proof"], asg.instance), "oracle verifier rejects the oracle proof"
        done[name] = {
            "name": name, "shape": shape, "k": k, "lookup_bits": bits, "x": "0x%x" % x, "seed": seed,
            "instance": ["0x%x" % v for v in (asg.instance[0] if asg.instance else [])],
            "vk_bytes": keys.vk_bytes().hex(), "challenges": {c: "0x%064x" % r[c] for c in ("theta", "beta", "gamma", "y", "x", "shplonk_y", "v", "u")},
            "proof_sha256": hashlib.sha256(r["proof"]).hexdigest(), "proof": r["proof"].hex(),
        }
        print("%s: %d proof bytes, %.0f s" % (name, len(r["proof"]), time.time() - t0), flush=True)
        del keys, r
        doc["cases"] = [done[n] for n, *_ in CASES if n in done]
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
