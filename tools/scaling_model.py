#!/usr/bin/env python3
"""Per-rank work of the N-GPU proof replay, measured on ONE GPU: builds rank r's share of the step (its MSM
slices, the consumed transforms every rank replays, its round-robin share of the leaf transforms) for
N = 1, 2, 4, 8 and times it without the 11 x 96-byte all-gather.  The slowest rank's time is what
`bench.py --gpus N` would report on a node (plus one small collective per step); no multi-GPU box is available
to the builder, so this is the evidence behind DESIGN.md's scaling expectation.  Prints one JSON document."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from halo2_scaffold_amd import replay as rp  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = rp.SHAPES[sys.argv[2]] if len(sys.argv) > 2 else rp.SHAPES["standard_plonk"]
steps, warm = 8, 2
rows = []
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    per_rank = []
    for rank in sorted({0, world - 1, world // 2}):
        R = rp.ProofReplay(shape, k, rank=rank, world=world, dist="uniform", combine=None)
        for _ in range(warm):
            R.step(); R.finish()
        t0 = time.perf_counter()
        for _ in range(steps):
            R.step(); R.finish()
        ms = (time.perf_counter() - t0) / steps * 1e3
        per_rank.append({"rank": rank, "ms": round(ms, 3), "counts_per_step": {a: b // (steps + warm) for a, b in R.counts.items()}})
        worst = max(worst, ms)
        R.release()
    base = base or worst
    rows.append({"world": world, "slowest_rank_ms": round(worst, 3), "speedup_vs_1": round(base / worst, 2), "ranks": per_rank})
print(json.dumps({"k": k, "shape": shape.name if hasattr(shape, "name") else str(shape), "what": "rank-local step time measured on one MI355X, collective excluded",
                  "rows": rows}, indent=1))
