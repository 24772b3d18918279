#!/usr/bin/env python3
"""Per-launch timeline of one data-true create_proof (h2mi_profile_dump): kernel, start, duration — where the proof's
wall clock goes between the host-side phase boundaries."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from halo2_scaffold_amd import circuits, keygen, prover  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = h2.lib
params = h2.ParamsKZG.setup(k, 0x5EC2E7)
c = circuits.StandardPlonk(None)
pk = keygen.keygen_pk(params, keygen.keygen_vk(params, c), c)
ws = prover.ProverWorkspace(params, pk)
for i in range(3):
    prover.create_proof(params, pk, circuits.StandardPlonk(5 + i), 10 + i, ws=ws)
lib.h2mi_profile_reset()
lib.h2mi_profile_filter(b"")
lib.h2mi_profile_enable(1)
tr = {}
prover.create_proof(params, pk, circuits.StandardPlonk(99), 99, ws=ws, trace=tr)
lib.h2mi_profile_enable(0)
need = C.c_size_t()
lib.h2mi_profile_dump(None, 0, C.byref(need))
buf = C.create_string_buffer(need.value + 16)
lib.h2mi_profile_dump(buf, need.value + 16, None)
print("phases (host wall clock, ms):", tr["phase_ms"])
for line in buf.value.decode().splitlines():
    name, t0, ms = line.split()
    print(f"{float(t0):9.3f} {float(ms):8.4f}  {name}")
