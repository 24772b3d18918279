#!/usr/bin/env python3
"""Generates tests/golden/standard_plonk_proofs.json from the oracle prover (oracle/prover.py).

SELF-DERIVED vectors (the reference holds no proof bytes and its prover uses OsRng): the proof of the
reference's StandardPlonk circuit at its own k = 5 (examples/standard_plonk.rs:26) and at DEGREE = 8 (BASELINE
configs[0]) for a fixed SRS secret, witness and rng seed, computed in plain Python integers.  They pin the oracle
prover, its verifier and the device prover to each other across rounds.
Usage: python tests/golden/make_proof_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import prover as OP  # noqa: E402

SRS_SECRET = 0x5EC2E7 + 0x48324D49
CASES = [(5, 0xDEADBEEF12345, 77), (8, 0x1234567890ABCDEF1234567890ABCDEF, 2024)]


def hx(v):
    return "0x%064x" % v


def main():
    out = {"srs_secret": hx(SRS_SECRET), "cases": []}
    for k, x, seed in CASES:
        pk = OP.ProvingKey(k, SRS_SECRET)
        r = OP.create_proof(pk, x, seed)
        assert OP.verify_proof(pk, r["proof"])
        out["cases"].append({
            "k": k, "witness_x": hx(x), "seed": seed,
            "vk_bytes": pk.vk_bytes().hex(), "vk_transcript_repr": hx(pk.transcript_repr),
            "challenges": {n: hx(v) for n, v in r["challenges"].items()},
            "proof": r["proof"].hex(),
        })
    with open(os.path.join(HERE, "standard_plonk_proofs.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
