#!/usr/bin/env python3
"""Prove one of the halo2-lib example circuits on the device and time it — BASELINE configs[2] (halo2_lib at
DEGREE 20) and configs[3] (range, LOOKUP_BITS 16, DEGREE 22; with --gpus N under torch.distributed.run the commitments
are slice MSMs combined across ranks at every transcript write, as in bench.py's N > 1 path).

    python tools/flex_proof.py --shape range --k 22 --lookup-bits 16 --proofs 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/flex_proof.py --shape range --k 13 --lookup-bits 8 --gpus 2      (H2MI_DIST_BACKEND=gloo H2MI_DEVICE=0 on one GPU)

Prints one JSON line (rank 0): ms per proof (steady state: buffers from the workspace), the first call's time, proof
length and sha256 (every rank holds the same bytes; asserted), keygen time."""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SRS_SECRET = 0x5EC2E7 + 0x48324D49


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", choices=["halo2_lib", "range", "poseidon"], default="halo2_lib")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--lookup-bits", type=int, default=16)
    ap.add_argument("--x", type=lambda v: int(v, 0), default=0x0123456789ABCDEF)
    ap.add_argument("--seed", type=int, default=31337)
    ap.add_argument("--count", type=int, default=1, help="range checks in one context (range shape): fills several columns at a small --k")
    ap.add_argument("--configure", action="store_true", help="take the column counts builder.config(k, Some(9)) would (flex.configure)")
    ap.add_argument("--proofs", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--kernels", action="store_true", help="one more proof with events around every launch: per-kernel device time")
    args = ap.parse_args()

    import torch

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("H2MI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    backend = os.environ.get("H2MI_DIST_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import _load_pkg

    h2 = _load_pkg.load()
    from halo2_scaffold_amd import flex
    from halo2_scaffold_amd._lib import check, lib
    from halo2_scaffold_amd.dist import PhaseCombiner, slice_bounds
    from halo2_scaffold_amd.params import ParamsKZG

    h2.init(local_rank)
    lookup = args.shape == "range"
    if args.shape == "poseidon":
        from halo2_scaffold_amd import poseidon

        build = lambda c, x: poseidon.hash_two_closure(c, x, x ^ 0x5A5A5A5A)
    elif lookup:
        build = lambda c, x: flex.range_closure(c, x, args.lookup_bits, args.count)
    else:
        build = lambda c, x: flex.halo2_lib_closure(c, x)
    cs = flex.configure(lookup, args.k, lambda c: build(c, args.x)) if args.configure else flex.FlexGateCS(lookup=lookup)
    closure = lambda x: build(cs, x)
    t0 = time.perf_counter()
    full = ParamsKZG.setup(args.k, SRS_SECRET)
    check(lib.h2mi_sync(), "sync")
    setup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    keys = flex.FlexKeys(full, cs, closure(args.x))  # keygen against the whole SRS: the vk holds full commitments
    check(lib.h2mi_sync(), "sync")
    keygen_s = time.perf_counter() - t0
    combiner, params = None, full
    if dist is not None:
        lo, hi = slice_bounds(1 << args.k, rank, world)
        params = full.register_slice(lo, hi)
        combiner = PhaseCombiner(80, backend, torch.device("cuda", local_rank))
    ws = flex.FlexWorkspace(params, keys, combiner=combiner)

    def barrier():
        check(lib.h2mi_sync(), "sync")
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    t0 = time.perf_counter()
    proof = flex.create_proof(params, keys, closure(args.x), args.seed, ws=ws)
    first_ms = 1e3 * (time.perf_counter() - t0)
    barrier()
    times = []
    witness_ms = []
    for i in range(args.proofs):
        t0 = time.perf_counter()
        asg = closure(args.x + 1 + i if not lookup else (args.x + 1 + i) % (1 << 64))
        witness_ms.append(1e3 * (time.perf_counter() - t0))  # the closure itself: host work in the reference too, not timed below
        t0 = time.perf_counter()
        flex.create_proof(params, keys, asg, args.seed + 1 + i, ws=ws)
        times.append(1e3 * (time.perf_counter() - t0))
    barrier()
    digest = hashlib.sha256(proof).hexdigest()
    kernels = None
    if args.kernels:
        import ctypes as C

        lib.h2mi_profile_reset()
        lib.h2mi_profile_filter(b"")
        lib.h2mi_profile_enable(1)
        flex.create_proof(params, keys, closure(args.x), args.seed, ws=ws)
        lib.h2mi_profile_enable(0)
        need = C.c_size_t()
        lib.h2mi_profile_dump(None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value + 16)
        lib.h2mi_profile_dump(buf, need.value + 16, None)
        kernels = {}
        for line in buf.value.decode().splitlines():
            name, _, dur = line.split()
            e = kernels.setdefault(name, {"ms": 0.0, "launches": 0})
            e["ms"] += float(dur)
            e["launches"] += 1
        kernels = {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
        lib.h2mi_profile_reset()
    if dist is not None:
        got = [None] * world
        dist.all_gather_object(got, digest)
        assert len(set(got)) == 1, "ranks hold different proofs"
        t = torch.tensor([sum(times) / max(len(times), 1)], dtype=torch.float64)
        if backend == "nccl":
            t = t.cuda(local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mean_ms = float(t.item())
    else:
        mean_ms = sum(times) / max(len(times), 1)
    if rank == 0:
        print(json.dumps({
            "circuit": args.shape, "k": args.k, "lookup_bits": args.lookup_bits if lookup else None, "n_gpus": world,
            "ms_per_proof": round(mean_ms, 3), "min_ms": round(min(times), 3) if times else None, "first_call_ms": round(first_ms, 3),
            "proofs_timed": len(times), "proof_bytes": len(proof), "proof_sha256": digest, "columns": [cs.num_advice, cs.num_lookup_advice, cs.num_fixed], "srs_setup_seconds": round(setup_s, 3),
            "keygen_vk_pk_seconds": round(keygen_s, 3), "advice_cells": len(closure(args.x).advice[0]),
            "witness_generation_ms_host": round(sum(witness_ms) / max(len(witness_ms), 1), 3),
            "combine": (("RCCL all_gather_into_tensor + device fold" if backend == "nccl" else "gloo all-gather + device fold")
                        + " at every transcript write") if dist is not None else "none (single GPU)",
            "combines_per_proof": combiner.combines // (1 + len(times) + (1 if args.kernels else 0)) if combiner is not None else 0,
            "kernels": kernels,
        }))
    ws.release()
    keys.release()
    if combiner is not None:
        combiner.release()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
