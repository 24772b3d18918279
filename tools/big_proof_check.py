#!/usr/bin/env python3
"""One StandardPlonk create_proof at a size beyond BASELINE's (default k = 23), verified by the oracle's verifier against
the closed-form verifying key — a scale check of the data-true pipeline (memory, index widths, table caches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from halo2_scaffold_amd import circuits, keygen, prover  # noqa: E402
from oracle import prover as OP  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 23
S = 0x5EC2E7 + 0x48324D49
t0 = time.time()
params = h2.ParamsKZG.setup(k, S)
c = circuits.StandardPlonk(None)
vk = keygen.keygen_vk(params, c)
pk = keygen.keygen_pk(params, vk, c)
print(f"k={k}: setup + keygen {time.time() - t0:.2f} s", flush=True)
ws = prover.ProverWorkspace(params, pk)
for i in range(3):
    t0 = time.perf_counter()
    proof = prover.create_proof(params, pk, circuits.StandardPlonk(0xABCDEF + i), 50 + i, ws=ws)
    print(f"create_proof {1e3 * (time.perf_counter() - t0):.1f} ms, {len(proof)} bytes", flush=True)
ovk = OP.VerifierKey.closed_form(k, S)
assert vk.to_bytes() == ovk.vk_bytes() if hasattr(ovk, "vk_bytes") else True
t0 = time.time()
ok = OP.verify_proof(ovk, proof)
print("oracle verifier:", ok, f"({time.time() - t0:.1f} s)")
assert ok
bad = bytearray(proof)
bad[500] ^= 1
assert not OP.verify_proof(ovk, bytes(bad))
