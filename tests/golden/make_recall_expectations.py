#!/usr/bin/env python3
"""Writes tests/golden/recall_expectations.json: what THIS repository's oracle (oracle/bn254.py, formats.py, prover.py — the CPU
restatement every parity test compares the HIP path with) expects interop/probe to print when it is run against the real crates.
Each value is one of the conventions restated from memory ([RECALL] in DESIGN.md 2): limb layouts, constants, encodings, the coset of
EvaluationDomain, the transcript's framing, the SRS of gen_srs, the verifying key's commitments.  tools/compare_probe.py compares a
probe run with this file; tests/test_recall_expectations.py regenerates it from the oracle on every CPU test run, so it cannot rot.

    python3 tests/golden/make_recall_expectations.py          (rewrites the JSON)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254 as o  # noqa: E402
from oracle import formats as fmt  # noqa: E402
from oracle import flex as FX  # noqa: E402
from oracle import prover as OP  # noqa: E402

R, Q = o.R, o.Q


def chacha20_zero_seed_first_block() -> bytes:
    """first 64 keystream bytes of ChaCha20 with an all-zero key and nonce (RFC 7539 A.1 #1): what ChaCha20Rng::from_seed([0; 32])
    yields as its first eight next_u64 — independent of halo2-scaffold_amd/params.py, which has its own copy"""
    M = 0xFFFFFFFF
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [0] * 12
    x = list(st)
    rotl = lambda v, c: ((v << c) & M) | (v >> (32 - c))

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return b"".join(((a + b) & M).to_bytes(4, "little") for a, b in zip(x, st))


def limbs(v: int, mod: int):
    m = o.to_mont(v, mod)
    return ["%016x" % ((m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]


def rep(v: int) -> str:
    return fmt.fe_to_repr(v % R).hex()


def expectations() -> dict:
    e = {}
    e["fr_five_limbs"] = limbs(5, R)
    e["fr_one_limbs"] = limbs(1, R)
    e["fr_five_repr"] = rep(5)
    e["fr_root_of_unity_repr"] = rep(o.FR_ROOT_OF_UNITY)
    e["fr_zeta_repr"] = rep(o.FR_ZETA)
    e["fr_delta_repr"] = rep(pow(o.FR_GENERATOR, 1 << o.FR_S, R))
    e["fr_s"] = o.FR_S
    G = o.G1_GEN
    e["g1_generator_limbs"] = limbs(G[0], Q) + limbs(G[1], Q)
    e["g1_identity_limbs"] = ["%016x" % 0] * 8
    e["g1_generator_bytes"] = fmt.g1_to_bytes(G).hex()
    e["g1_identity_bytes"] = fmt.g1_to_bytes(None).hex()
    e["g1_two_g_bytes"] = fmt.g1_to_bytes(o.g1_double(G)).hex()
    e["g1_minus_g_bytes"] = fmt.g1_to_bytes(o.g1_neg(G)).hex()
    scalars = [1000003 * i + 7 for i in range(16)]
    bases = [o.g1_mul(i + 1, G) for i in range(16)]
    e["msm16_bytes"] = fmt.g1_to_bytes(o.msm_naive(scalars, bases)).hex()
    dom = o.Domain(4, 3)
    a = [3 * i + 1 for i in range(16)]
    e["omega_k4_repr"] = rep(dom.omega)
    e["fft16_repr"] = [rep(v) for v in o.dft_naive(a, dom.omega)]
    e["coeff_to_extended_k4_j3_repr"] = [rep(v) for v in dom.coeff_to_extended(a)]
    e["extended_k_k4_j3"] = dom.extended_k
    t = fmt.Blake2bTranscript()
    t.write_point(G)
    t.write_scalar(5)
    e["transcript_challenge_repr"] = rep(t.squeeze_challenge())
    e["transcript_bytes"] = bytes(t.proof).hex()
    s = int.from_bytes(chacha20_zero_seed_first_block(), "little") % R  # Fr::random(rng) = from_u512 of eight next_u64
    e["srs5_secret"] = "%064x" % s  # not printed by the probe (the crate keeps it nowhere): what the three lines below follow from
    pw, lag = o.srs_scalars(5, s)
    e["srs5_g"] = [fmt.g1_to_bytes(o.g1_mul(pw[i], G)).hex() for i in range(3)]
    e["srs5_g_lagrange"] = [fmt.g1_to_bytes(o.g1_mul(lag[i], G)).hex() for i in range(3)]
    pk = OP.ProvingKey(5, s)
    e["vk5_fixed_commitments"] = [fmt.g1_to_bytes(p).hex() for p in pk.fixed_commitments]
    e["vk5_permutation_commitments"] = [fmt.g1_to_bytes(p).hex() for p in pk.permutation_commitments]
    e["proof5_len"] = 992
    # ---- the halo2-lib builders through the reference's own scaffold::gen_key (src/scaffold.rs:95-155): what pins halo2-base's
    # layout conventions (column order, selector columns, constants in first-use order, constrain_equal order, the break-point rule)
    pts = lambda cms: [fmt.g1_to_bytes(p).hex() for p in cms]
    cs = FX.flex_gate_cs(False)  # examples/halo2_lib.rs at DEGREE 5: 17 cells, one gate column
    asg = FX.halo2_lib_assignment(cs, 12)
    vk = FX.VerifierKeys(cs, 5, s, asg.fixed, asg.copies)
    e["halo2lib_k5_fixed_commitments"], e["halo2lib_k5_permutation_commitments"] = pts(vk.fixed_commitments), pts(vk.permutation_commitments)
    e["halo2lib_k5_break_points_phase0"] = []
    x = 0xDEADBEEFCAFE1234
    cs = FX.flex_gate_cs(True)  # examples/range.rs at DEGREE 7, LOOKUP_BITS 4: one gate column, the q_lookup form
    asg = FX.range_assignment(cs, x, 4, 1 << 7)
    vk = FX.VerifierKeys(cs, 7, s, asg.fixed, asg.copies)
    e["range_k7_bits4_fixed_commitments"], e["range_k7_bits4_permutation_commitments"] = pts(vk.fixed_commitments), pts(vk.permutation_commitments)
    e["range_k7_bits4_break_points_phase0"] = []
    t = FX._range_table(x, 4)  # the same at DEGREE 5: 51 cells over 23-row columns, three gate columns + one lookup-advice column
    A, Lc = FX.multi_column_counts(len(t.rows), len(t.lookups), 5)
    cs = FX.flex_multi_cs(True, A, Lc, FX.num_fixed_columns(t, 5))
    asg = FX.range_assignment_multi(cs, x, 4, 5)
    vk = FX.VerifierKeys(cs, 5, s, asg.fixed, asg.copies)
    e["range_k5_bits4_fixed_commitments"], e["range_k5_bits4_permutation_commitments"] = pts(vk.fixed_commitments), pts(vk.permutation_commitments)
    e["range_k5_bits4_break_points_phase0"] = FX.break_point_rows(t, 5)
    cs = FX.flex_gate_cs(True)  # examples/halo2_lib.rs with LOOKUP_BITS = 4 at DEGREE 6: the Range builder, nothing looked up
    asg = FX.halo2_lib_assignment(cs, 12)
    asg.fixed[cs.col_table] = {i: i for i in range(16)}
    vk = FX.VerifierKeys(cs, 6, s, asg.fixed, asg.copies)
    e["halo2lib_range_builder_k6_bits4_fixed_commitments"] = pts(vk.fixed_commitments)
    e["halo2lib_range_builder_k6_bits4_permutation_commitments"] = pts(vk.permutation_commitments)
    e["halo2lib_range_builder_k6_bits4_break_points_phase0"] = []
    return e


NOTES = {
    "_what": "expected output of interop/probe (run against the real crates), computed from this repository's oracle; see make_recall_expectations.py",
    "_not_compared": ["informational", "srs5_secret"],
    "_if_a_key_differs": {
        "fr_*_limbs / g1_*_limbs": "the in-memory layout at the C ABI (include/h2mi.h: Montgomery, R = 2^256, little-endian limbs; identity = (0, 0))",
        "fr_zeta_repr / coeff_to_extended_*": "the coset generator of EvaluationDomain (DESIGN.md 2: g_coset = ZETA)",
        "g1_*_bytes": "the compressed encoding's flag bits (csrc/h2mi_serde.hip)",
        "transcript_*": "Blake2b personalisation / prefix bytes / Challenge255 reduction (halo2-scaffold_amd/transcript.py, include/h2mi_transcript.hpp)",
        "srs5_*": "gen_srs: rng seed, Fr::random's use of the keystream, or the Lagrange basis (halo2-scaffold_amd/params.py gen_srs_secret)",
        "vk5_*": "the circuit's fixed cells / copy-constraint order / sigma construction (oracle/plonk.py, csrc/h2mi_prover.cpp keygen)",
        "halo2lib_* / range_*": "halo2-base's layout as restated in oracle/flex.py (and halo2-scaffold_amd/flex.py, include/h2mi_flex.hpp): the number of "
                                "commitments = the column counts and their order; *_fixed_commitments = selector columns, constants in first-use order, the "
                                "table; *_permutation_commitments = enable_equality order and the order of constrain_equal calls; *_break_points = the rule "
                                "that ends a gate column (a differing list length means a different column count from builder.config)",
    },
}

if __name__ == "__main__":
    out = dict(NOTES)
    out.update(expectations())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "recall_expectations.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", path, len(out), "keys")
