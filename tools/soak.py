#!/usr/bin/env python3
"""Soak run: many proofs back to back through reused workspaces (side streams, scratch pool, table caches under eviction),
every proof checked by the oracle's verifier — a race or a stale buffer shows up as a rejected proof.  Prints device
memory before / after.  Usage: python tools/soak.py [n_standard] [n_range] [n_poseidon]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from halo2_scaffold_amd import circuits, flex, keygen, poseidon, prover  # noqa: E402
from oracle import flex as FX  # noqa: E402
from oracle import prover as OP  # noqa: E402

S = 0x5EC2E7 + 0x48324D49
n_std, n_rng, n_pos = (int(v) for v in (sys.argv[1:] + ["200", "100", "30"])[:3])


def free_bytes():
    f = torch.cuda.mem_get_info()[0]
    return f


t_all = time.time()
# StandardPlonk at k = 14
k = 14
params = h2.ParamsKZG.setup(k, S)
c = circuits.StandardPlonk(None)
vk = keygen.keygen_vk(params, c)
pk = keygen.keygen_pk(params, vk, c)
ws = prover.ProverWorkspace(params, pk)
ovk = OP.VerifierKey.closed_form(k, S)
prover.create_proof(params, pk, circuits.StandardPlonk(1), 1, ws=ws)
h2._lib.check(h2.lib.h2mi_sync(), "sync")
m0 = free_bytes()
bad = 0
for i in range(n_std):
    proof = prover.create_proof(params, pk, circuits.StandardPlonk(1000 + i), 7000 + i, ws=ws)
    if not OP.verify_proof(ovk, proof):
        bad += 1
        print("REJECTED standard proof", i, flush=True)
print(f"standard_plonk k={k}: {n_std} proofs, {bad} rejected, free memory delta {m0 - free_bytes()} B", flush=True)
ws.release()
pk.release()
# range at k = 12, poseidon at k = 13 against the same SRS size class
for name, kk, lookup, closure, count in (("range", 12, True, None, n_rng), ("poseidon", 13, False, None, n_pos)):
    p2 = h2.ParamsKZG.setup(kk, S)
    cs = flex.FlexGateCS(lookup=lookup)
    mk = (lambda x: flex.range_closure(cs, x, 9)) if lookup else (lambda x: poseidon.hash_two_closure(cs, x, x * 3 + 1))
    keys = flex.FlexKeys(p2, cs, mk(5))
    w2 = flex.FlexWorkspace(p2, keys)
    ocs = FX.flex_gate_cs(lookup)
    a0 = mk(5)
    oa = FX.Assignment(ocs)
    oa.fixed = [dict(f) if f is not None else {i: v for i, v in enumerate(a0.table_values)} for f in a0.fixed]
    oa.copies = list(a0.copies)
    ovk2 = FX.VerifierKeys(ocs, kk, S, oa.fixed, oa.copies)
    assert ovk2.transcript_repr == keys.transcript_repr
    flex.create_proof(p2, keys, mk(5), 1, ws=w2)
    h2._lib.check(h2.lib.h2mi_sync(), "sync")
    m0 = free_bytes()
    bad = 0
    for i in range(count):
        asg = mk(0xABCDEF00 + 977 * i)
        proof = flex.create_proof(p2, keys, asg, 100 + i, ws=w2)
        if not FX.verify(ovk2, proof, [asg.instance]):
            bad += 1
            print("REJECTED", name, i, flush=True)
    print(f"{name} k={kk}: {count} proofs, {bad} rejected, free memory delta {m0 - free_bytes()} B", flush=True)
    w2.release()
    keys.release()
    p2.release()
# many columns (round 5): 24 range checks at DEGREE 7 (11 gate + 4 lookup-advice columns), keys and workspace created and released
# every ten proofs — the per-prover buffers (three forms per column, result slots sized by the largest phase) must come back
kk, bits, cnt = 7, 4, 24
p3 = h2.ParamsKZG.setup(kk, S)
mk = lambda cs, x: flex.range_closure(cs, x, bits, cnt)
cs = flex.configure(True, kk, lambda c: mk(c, 5))
ocs = FX.flex_multi_cs(True, cs.num_advice, cs.num_lookup_advice, cs.num_fixed)
oasg = FX.range_many_assignment_multi(ocs, 5, bits, kk, cnt)
ovk3 = FX.VerifierKeys(ocs, kk, S, oasg.fixed, oasg.copies)
keys = flex.FlexKeys(p3, cs, mk(cs, 5))
w3 = flex.FlexWorkspace(p3, keys)
flex.create_proof(p3, keys, mk(cs, 5), 1, ws=w3)
w3.release()
keys.release()
h2._lib.check(h2.lib.h2mi_sync(), "sync")
m0 = free_bytes()
bad = 0
n_wide = max(n_pos, 10)
for i in range(n_wide):
    if i % 10 == 0:
        if i:
            w3.release()
            keys.release()
        keys = flex.FlexKeys(p3, cs, mk(cs, 5))
        w3 = flex.FlexWorkspace(p3, keys)
    asg = mk(cs, 0x1234500 + 31 * i)
    proof = flex.create_proof(p3, keys, asg, 900 + i, ws=w3)
    if not FX.verify(ovk3, proof, [asg.instance]):
        bad += 1
        print("REJECTED wide", i, flush=True)
w3.release()
keys.release()
h2._lib.check(h2.lib.h2mi_sync(), "sync")
print(f"range x{cnt} k={kk} ({cs.num_advice} + {cs.num_lookup_advice} columns): {n_wide} proofs, {bad} rejected, free memory delta {m0 - free_bytes()} B", flush=True)
p3.release()
print(f"soak done in {time.time() - t_all:.1f} s")
