#!/usr/bin/env python3
"""N data-true create_proof() calls at 2^k rows after warm-up — the program to put behind `rocprofv3 --kernel-trace --`
(tools/trace_timeline.py turns the trace of the LAST proof into a timeline with gaps and queue ids).
Usage: proof_loop.py K [PROOFS] [standard_plonk|halo2_lib|poseidon|range] [LOOKUP_BITS]; prints wall clock per proof and
the number of kernel launches of one proof."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from halo2_scaffold_amd import circuits, flex, keygen, poseidon, prover  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = sys.argv[3] if len(sys.argv) > 3 else "standard_plonk"
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 8
lib = h2.lib
params = h2.ParamsKZG.setup(k, 0x5EC2E7)
if shape == "standard_plonk":
    c = circuits.StandardPlonk(None)
    pk = keygen.keygen_pk(params, keygen.keygen_vk(params, c), c)
    ws = prover.ProverWorkspace(params, pk)
    run = lambda i: prover.create_proof(params, pk, circuits.StandardPlonk(5 + i), 10 + i, ws=ws)
else:
    cs = flex.FlexGateCS(lookup=shape == "range")
    mk = {"halo2_lib": lambda x: flex.halo2_lib_closure(cs, x), "poseidon": lambda x: poseidon.hash_two_closure(cs, x, x + 1),
          "range": lambda x: flex.range_closure(cs, x, bits)}[shape]
    keys = flex.FlexKeys(params, cs, mk(3))
    ws = flex.FlexWorkspace(params, keys)
    asgs = [mk(100 + i) for i in range(proofs + 3)]
    run = lambda i: flex.create_proof(params, keys, asgs[i % len(asgs)], 10 + i, ws=ws)
for i in range(3):
    run(i)
h2._lib.check(lib.h2mi_sync(), "sync")
lib.h2mi_profile_reset()
lib.h2mi_profile_filter(b"")
times = []
prof = None
if os.environ.get("PROOF_LOOP_CPROFILE"):  # where the HOST spends a proof (Python host only): top functions by own time
    import cProfile

    prof = cProfile.Profile()
    prof.enable()
for i in range(proofs):
    t0 = time.perf_counter()
    run(3 + i)
    times.append(time.perf_counter() - t0)
if prof is not None:
    import pstats

    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(22)
h2._lib.check(lib.h2mi_sync(), "sync")
# launches of one proof, counted by an extra (untimed, event-bracketed) proof at the very end: the trace's last
# `launches` kernel dispatches belong to it, the `launches` before them to the last timed proof
lib.h2mi_profile_enable(1)
run(99)
lib.h2mi_profile_enable(0)
tot, cnt = C.c_double(), C.c_uint64()
lib.h2mi_profile_query(b"", C.byref(tot), C.byref(cnt))
print(json.dumps({"k": k, "shape": shape, "ms_per_proof": round(1e3 * sum(times) / len(times), 3), "min_ms": round(1e3 * min(times), 3),
                  "launches_per_proof": cnt.value, "device_kernel_ms_sum": round(tot.value, 3)}))
