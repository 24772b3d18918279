"""Host-side phase boundaries of the Python create_proof at 2^20 rows, averaged over a loop (each phase ends in a device sync)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import circuits, keygen, prover
k = 20
params = h2.ParamsKZG.setup(k, 0x5EC2E7)
c = circuits.StandardPlonk(None)
pk = keygen.keygen_pk(params, keygen.keygen_vk(params, c), c)
ws = prover.ProverWorkspace(params, pk)
for i in range(3): prover.create_proof(params, pk, circuits.StandardPlonk(5 + i), 10 + i, ws=ws)
acc = {}; R = 15; t0 = time.perf_counter()
for i in range(R):
    tr = {}
    prover.create_proof(params, pk, circuits.StandardPlonk(50 + i), 20 + i, ws=ws, trace=tr)
    for name, ms in tr["phase_ms"]: acc[name] = acc.get(name, 0) + ms
print("loop ms/proof", (time.perf_counter() - t0) / R * 1e3)
for k_, v in acc.items(): print(f"{k_:28s} {v / R:8.3f}")
