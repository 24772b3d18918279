#!/usr/bin/env python3
"""Per-rank work of the N-GPU proof replay, measured on ONE GPU: builds rank r's share of the step (its MSM
slices, the consumed transforms every rank replays, its round-robin share of the leaf transforms) for
N = 1, 2, 4, 8 and times it; the per-phase combine (five RCCL all-gathers of <= 4 x 96 B + device folds per step) is
timed separately with a single-rank RCCL group on this GPU — launch and fold cost, not the xGMI hop — and added.
The slowest rank's time + the combines is what `bench.py --gpus N` is expected to report on a node; no multi-GPU box
is available to the builder, so this is the evidence behind DESIGN.md's scaling expectation.  Prints one JSON document."""
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
        R = rp.ProofReplay(shape, k, rank=rank, world=world, dist="uniform")
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
# the collective: a single-rank RCCL process group on this GPU (what H2MI_FORCE_DIST=1 rehearses)
coll_ms = None
try:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from halo2_scaffold_amd.dist import PhaseCombiner

    pc = PhaseCombiner(shape.msm_per_proof, "nccl", torch.device("cuda", 0))
    phases = [3, 4, 2, 1, 1] if shape.name == "standard_plonk" else [shape.msm_per_proof]
    for it in range(3 + 20):
        if it == 3:
            h2.lib.h2mi_sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        first = 0
        for cnt in phases:
            h2.lib.h2mi_join()
            pc.combine(first, cnt)
            first += cnt
    h2.lib.h2mi_sync(); torch.cuda.synchronize()
    coll_ms = (time.perf_counter() - t0) / 20 * 1e3
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    coll_ms = f"not measured: {e}"
# the xGMI hop itself cannot be measured on one GPU.  ESTIMATE (an assumption, stated so that the first hardware run can
# replace it): an RCCL ring all-gather of <= 4 x 96 B per rank is latency-bound — its single-rank launch + fold cost is in
# coll_ms above; each of the N - 1 ring steps adds one xGMI store + flag round trip, taken as HOP_US = 2.5 us (LL protocol,
# a few hundred bytes; override with H2MI_HOP_US), so one combine costs (N - 1) * HOP_US more and a step has `len(phases)` of them
HOP_US = float(os.environ.get("H2MI_HOP_US", "2.5"))
n_combines = 5 if shape.name == "standard_plonk" else 1
for r in rows:
    if isinstance(coll_ms, float) and r["world"] > 1:
        r["with_combines_ms"] = round(r["slowest_rank_ms"] + coll_ms, 3)
        r["speedup_vs_1_with_combines"] = round(base / r["with_combines_ms"], 2)
        hop_ms = n_combines * (r["world"] - 1) * HOP_US * 1e-3
        r["xgmi_hop_estimate_ms"] = round(hop_ms, 4)
        r["with_combines_and_hop_estimate_ms"] = round(r["with_combines_ms"] + hop_ms, 3)
        r["speedup_vs_1_with_hop_estimate"] = round(base / (r["with_combines_ms"] + hop_ms), 2)
print(json.dumps({"k": k, "combines_per_step_ms_single_rank_rccl": coll_ms if not isinstance(coll_ms, float) else round(coll_ms, 3), "shape": shape.name if hasattr(shape, "name") else str(shape), "what": "rank-local step time measured on one MI355X; the five per-phase combines timed with a 1-rank RCCL group and added; "
                  "xgmi_hop_estimate_ms = combines x (N - 1) ring steps x %.1f us per step: an ASSUMPTION (no multi-GPU node was available), not a measurement" % HOP_US,
                  "rows": rows}, indent=1))
