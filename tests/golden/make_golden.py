#!/usr/bin/env python3
"""Generates tests/golden/bn254_vectors.json from the pure-Python oracle (oracle/bn254.py).

The reference (/root/reference) ships no vectors for this path and cannot be run here,
so these are SELF-DERIVED known answers: every value below is computed by big-integer arithmetic from
the published definitions (sum s_i P_i; out[i] = sum a[j] w^(ij); g[i] = s^i G), with inputs taken
from the seeded generator.  They pin the oracle, the C restatement and the HIP path to each other.
Usage: python tests/golden/make_golden.py   (rewrites the JSON next to this script)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bn254 as o  # noqa: E402


def hx(v):
    return "0x%064x" % v


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def formats_vectors():
    """wire formats (oracle/formats.py): encodings of a few points, a transcript run, the SRS file of (k=3, s=5)."""
    import hashlib

    from oracle import formats as fm

    pts = [o.g1_mul(k, o.G1_GEN) for k in (1, 2, 3, 0x1234567, o.R - 1)]
    scal = [0, 1, 0xDEADBEEF, o.R - 1]
    t = fm.Blake2bTranscript()
    ch = []
    for p in pts:
        t.write_point(p)
    ch.append(t.squeeze_challenge())
    for v in scal:
        t.write_scalar(v)
    ch.append(t.squeeze_challenge())
    ch.append(t.squeeze_challenge())
    return {
        "points": [pt(p) for p in pts],
        "compressed": [fm.g1_to_bytes(p).hex() for p in pts] + [fm.g1_to_bytes(None).hex()],
        "scalars": [hx(v) for v in scal],
        "transcript_proof": bytes(t.proof).hex(),
        "transcript_challenges": [hx(c) for c in ch],
        "g2_generator": fm.g2_to_bytes(fm.G2_GEN).hex(),
        "srs_k3_s5_sha256": hashlib.sha256(fm.srs_bytes(3, 5)).hexdigest(),
        "srs_k3_s5_tail": fm.srs_bytes(3, 5)[-64:].hex(),
    }


def replay_commitments(k):
    """BASELINE config 0/1 (standard_plonk, plumbing size): the 11 commitments of the hot-path replay at 2^k rows,
    computed entirely on the CPU (big-integer SRS scalars, C restatement of best_multiexp for the sums).
    Inputs: SRS secret and column vectors exactly as halo2-scaffold_amd/replay.py seeds them."""
    import numpy as np

    from oracle import cref

    n = 1 << k
    secret = 0x5EC2E7 + 0x48324D49
    pw, lag = o.srs_scalars(k, secret)
    g = cref.g1_mul_gen(o.pack(pw, o.R), 4)
    gl = cref.g1_mul_gen(o.pack(lag, o.R), 4)
    dom = o.Domain(k, 3)
    seed = o.SEED
    advice = [o.random_field_limbs(n, seed + 10 + i) for i in range(3)]
    perm_z = [o.random_field_limbs(n, seed + 60 + i) for i in range(3)]
    random_poly = o.random_field_limbs(n, seed + 20)
    hvals = o.unpack(o.random_field_limbs(1 << dom.extended_k, seed + 30), o.R)
    hcoef = dom.extended_to_coeff(hvals)
    commits = [cref.msm(c, gl, 2) for c in advice + perm_z]
    commits.append(cref.msm(random_poly, g, 2))
    for piece in range(2):
        commits.append(cref.msm(o.pack(hcoef[piece * n : (piece + 1) * n], o.R), g, 2))
    adv0 = o.pack(dom.lagrange_to_coeff(o.unpack(advice[0], o.R)), o.R)
    z0 = o.pack(dom.lagrange_to_coeff(o.unpack(perm_z[0], o.R)), o.R)
    commits.append(cref.msm(adv0, g, 2))
    commits.append(cref.msm(z0, g, 2))
    aff = cref.normalize(np.stack(commits))
    return {"k": k, "srs_secret": hx(secret), "commitments": [pt(p) for p in o.unpack_points(aff)]}


def main():
    out = {"comment": "self-derived known answers (oracle/bn254.py); canonical (non-Montgomery) integers", "seed": o.SEED}
    out["constants"] = {
        "q": hx(o.Q), "r": hx(o.R), "fr_root_of_unity": hx(o.FR_ROOT_OF_UNITY), "fr_zeta": hx(o.FR_ZETA),
        "fr_mont_R": hx(o.to_mont(1, o.R)), "fr_mont_R2": hx(o.to_mont(o.to_mont(1, o.R), o.R)),
        "fq_mont_R": hx(o.to_mont(1, o.Q)), "fq_mont_R2": hx(o.to_mont(o.to_mont(1, o.Q), o.Q)),
        "fr_inv64": hx((-pow(o.R, -1, 1 << 64)) % (1 << 64)), "fq_inv64": hx((-pow(o.Q, -1, 1 << 64)) % (1 << 64)),
    }
    G = o.G1_GEN
    out["g1"] = {
        "G": pt(G), "2G": pt(o.g1_double(G)), "3G": pt(o.g1_add(o.g1_double(G), G)), "(r-1)G": pt(o.g1_mul(o.R - 1, G)),
        "12345G": pt(o.g1_mul(12345, G)), "seedG": pt(o.g1_mul(o.SEED, G)),
    }
    # NTT: n = 64, seeded input (canonical values), forward / coset / inverse
    k = 6
    n = 1 << k
    a = o.unpack(o.random_field_limbs(n, o.SEED + 1), o.R)
    w = o.omega_for(k)
    out["ntt"] = {
        "log_n": k, "omega": hx(w), "input": [hx(x) for x in a], "forward": [hx(x) for x in o.ntt(a, w)],
        "coset_zeta": [hx(x) for x in o.ntt_ext(a, w, pre_base=o.FR_ZETA)],
        "inverse_scaled": [hx(x) for x in o.intt(a, w)],
    }
    # MSM: n = 64 bases s_i * G with seeded s_i, seeded scalars
    m = 64
    bs = o.unpack(o.random_field_limbs(m, o.SEED + 2), o.R)
    bases = [o.g1_mul(s, G) for s in bs]
    sc = o.unpack(o.random_field_limbs(m, o.SEED), o.R)
    out["msm"] = {
        "n": m, "base_scalars": [hx(x) for x in bs], "scalars": [hx(x) for x in sc], "result": pt(o.msm_naive(sc, bases)),
        "result_all_ones": pt(o.msm_naive([1] * m, bases)),
        "result_witness_like": pt(o.msm_naive(o.unpack(o.witness_like_limbs(m, 3), o.R), bases)),
    }
    # SRS k = 3, s = 5 and the commitment of a fixed polynomial both ways
    g, gl = o.srs(3, 5)
    coeffs = [3, 1, 4, 1, 5, 9, 2, 6]
    out["srs"] = {"k": 3, "s": 5, "g": [pt(p) for p in g], "g_lagrange": [pt(p) for p in gl], "coeffs": coeffs,
                  "commit": pt(o.msm_naive(coeffs, g))}
    # extended domain (k = 3, cs degree 3)
    d = o.Domain(3, 3)
    out["domain"] = {"k": 3, "j": 3, "extended_k": d.extended_k, "coeff_to_extended": [hx(x) for x in d.coeff_to_extended(coeffs)]}
    out["replay_k8"] = replay_commitments(8)
    out["formats"] = formats_vectors()
    with open(os.path.join(HERE, "bn254_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "bn254_vectors.json"))


if __name__ == "__main__":
    main()
