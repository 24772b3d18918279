#!/usr/bin/env python3
"""bench.py — prover hot-path replay of a 2^k-row StandardPlonk proof on N MI355X GPUs.

A "step" is one proof's worth of hot-path work: 11 MSM(n) over BN254 G1 +
6 iNTT(n) + 6 coset-NTT(2n) + 1 coset-iNTT(2n) over the BN254 scalar field, on synthetic vectors that
are already resident in HBM when the timed region starts.  It is NOT create_proof(): no Rust prover is
linked (none can be built here), so gate evaluation / transcript / witness generation are not timed.

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N --steps K --warmup W            (plain: starts its own torch.distributed.run child, see spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W               (a pre-set WORLD_SIZE is honoured: no second launcher)
  python bench.py --gpus N --single-process ...            (ONE process drives N devices: h2mi_init_devices(N))

N > 1: one process per GPU; the total work is fixed ("strong" scaling): every rank owns a contiguous
1/N slice of both base sets, runs each MSM on its slice and the 96-byte partial points are combined at every
transcript join (five per StandardPlonk proof) by an RCCL all-gather + device fold, all in HBM; every NTT runs on one GPU (by design): a transform whose output feeds a later
commitment is replayed by every rank, the others are spread round-robin over the ranks.
Rank 0 prints ONE JSON line (the contract's fields plus `roofline`, `issue_roofline`, `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` launched plainly (no WORLD_SIZE in the environment): start the N ranks as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` in a CHILD process, relay
    rank 0's JSON line on stdout (everything else the ranks print goes to stderr) and return the child's exit code.
    The parent imports neither torch nor the library and makes no HIP call: on this pool a process that has touched the GPU
    must not exec, and N ranks + an idle parent that holds a GPU context would also count against the box's process limit."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: no WORLD_SIZE in the environment, starting " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 without printing a result line", file=sys.stderr)
        rc = 1
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=20, help="log2 rows (BASELINE.json metric is quoted at k=20)")
    ap.add_argument("--dist", choices=["uniform", "witness", "circuit"], default="uniform",
                    help="advice/instance columns: uniform (dense worst case, default), witness (90 %% zeros), circuit (the 3 used rows + "
                         "blinding rows of the reference's StandardPlonk; every other vector stays dense)")
    ap.add_argument("--shape", choices=["standard_plonk", "halo2_lib_gate", "range_lookup"], default="standard_plonk",
                    help="proof shape to replay (BASELINE.json metric is quoted on standard_plonk)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-evaluate-h", action="store_true",
                    help="standard_plonk, N = 1: also time the device quotient evaluation (outside BASELINE's MSM+NTT metric)")
    ap.add_argument("--replicate-all-ntts", action="store_true",
                    help="N > 1: every rank replays every NTT (default: leaf transforms are spread round-robin)")
    ap.add_argument("--cpu-sample-log", type=int, default=None, help="log2 size of the CPU baseline sample MSM/NTT")
    ap.add_argument("--no-create-proof", action="store_true",
                    help="skip the data-true create_proof() timing (standard_plonk, N = 1) reported beside the MSM+NTT headline")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: ONE process drives the N devices (h2mi_init_devices(N): what the reference's single create_proof call "
                         "is) instead of one process per GPU over RCCL — the other deployment of the same slice partition")
    ap.add_argument("--no-msm-only", action="store_true", help="N > 1: skip the MSM-only speed-up measurement")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ:
        # plain launch: become the launcher BEFORE torch / the library / any HIP call exists in this process
        raise SystemExit(spawn_ranks(args.gpus, argv))

    import torch  # first: the HIP runtime torch loads is the one libh2mi.so then shares

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_devices = args.gpus if args.single_process else 1  # devices THIS process drives
    if args.single_process:
        if world != 1:
            raise SystemExit("--single-process is one process driving N devices: do not launch it under torch.distributed.run")
    elif world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch plainly (bench.py starts its own ranks) or with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}")
    dist = None
    # rehearsal knobs (not used by the driver): H2MI_DIST_BACKEND=gloo runs the N > 1 path with CPU
    # collectives, H2MI_DEVICE=<i> pins every rank to one GPU so a 1-GPU box can exercise world_size 2
    backend = os.environ.get("H2MI_DIST_BACKEND", "nccl")
    if "H2MI_DEVICE" in os.environ:
        local_rank = int(os.environ["H2MI_DEVICE"])
    # H2MI_FORCE_DIST=1: take the N > 1 code path (process group, all-gather + fold, barriers) with a single
    # rank, so that a 1-GPU box exercises the RCCL plumbing the driver's multi-GPU run depends on
    force_dist = world == 1 and os.environ.get("H2MI_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    ranks_info = rank_device_report(torch, dist, rank, local_rank, world)
    if n_devices > 1:  # one process, N devices: entry i of the library is device i (i % visible with H2MI_VIRTUAL_DEVICES)
        vis = torch.cuda.device_count()
        if vis < n_devices and "H2MI_VIRTUAL_DEVICES" not in os.environ:
            raise SystemExit(f"--single-process --gpus {n_devices} but {vis} device(s) visible (H2MI_VIRTUAL_DEVICES=1 rehearses on fewer)")
        ranks_info = [dict(rank_device_report(torch, None, 0, i % max(vis, 1), 1)[0], entry=i) for i in range(n_devices)]

    import _load_pkg

    h2 = _load_pkg.load()
    from halo2_scaffold_amd import replay as rp

    if args.single_process and n_devices > 1:
        h2._lib.check(h2.lib.h2mi_init_devices(n_devices), "h2mi_init_devices")
    else:
        h2.init(local_rank)
    lib = h2.lib
    shape = rp.SHAPES[args.shape]
    R = rp.ProofReplay(shape, args.k, rank=rank, world=world, dist=args.dist, combine_backend=backend if dist is not None else None,
                       torch_device=torch.device("cuda", local_rank), spread_leaf_ntts=not args.replicate_all_ntts,
                       with_evaluate_h=args.with_evaluate_h)
    n = R.n

    def sync_all():
        h2._lib.check(lib.h2mi_sync(), "sync")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    commitments = None
    for _ in range(args.warmup):
        R.step()
        commitments = R.finish()

    # G1 additions one step performs (bucket insertions + reduction adds), summed over the 11 MSMs
    import ctypes as C

    adds_local = 0
    n_coeff = 1 + (shape.cs_degree - 1) + 2  # random poly, h pieces, SHPLONK: dense coefficient vectors
    probes = [(True, R.advice[0], shape.n_advice), (True, R.perm_z[0], shape.n_perm_z), (False, R.random_poly, n_coeff)]
    if shape.n_lookups:
        probes.append((True, R.lookup[0], 3 * shape.n_lookups))
    for lagrange, buf, mult in probes:
        handle = R.params.g_lagrange_handle if lagrange else R.params.g_handle
        h2._lib.check(lib.h2mi_msm_bn254_g1_dev(handle, buf.ptr + R.lo * 32, R.n_local, R.probe_out.ptr, None), "msm")
        ba, ra = C.c_uint64(), C.c_uint64()
        h2._lib.check(lib.h2mi_msm_last_stats(handle, C.byref(ba), C.byref(ra)), "stats")
        adds_local += (ba.value + ra.value) * mult  # vectors of one kind share a distribution
    adds = adds_local
    if dist is not None:
        t = torch.tensor([adds_local], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t)
        adds = int(t.item())

    lib.h2mi_profile_reset()
    lib.h2mi_profile_filter(b"k_msm_accum")
    lib.h2mi_profile_enable(1)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        R.step()
        commitments = R.finish()
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.h2mi_profile_enable(0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tot_ms, cnt = C.c_double(), C.c_uint64()
    lib.h2mi_profile_query(b"k_msm_accum", C.byref(tot_ms), C.byref(cnt))
    accum_ms = tot_ms.value / max(cnt.value, 1)

    # per-phase device time (untimed extra pass, all kernels bracketed by events)
    lib.h2mi_profile_reset()
    lib.h2mi_profile_filter(b"")
    lib.h2mi_profile_enable(1)
    R.step()
    R.finish()
    lib.h2mi_profile_enable(0)
    phases = {}
    for name in ["k_msm_bin_count", "binscan_hipcub", "k_scan_seg", "k_msm_bin_scatter", "k_msm_bin_sort", "k_msm_accum", "k_msm_fold", "k_msm_finish", "k_msm_rowcol", "k_msm_weighted", "k_msm_final",
                 "k_ntt_pass_col", "k_ntt_pass_row", "k_scale_powers"]:
        lib.h2mi_profile_query(name.encode(), C.byref(tot_ms), C.byref(cnt))
        phases[name] = {"ms": round(tot_ms.value, 4), "launches": cnt.value}
    lib.h2mi_profile_reset()
    msm_ms = sum(v["ms"] for k, v in phases.items() if k.startswith("k_msm") or k.startswith("binscan") or k == "k_scan_seg")
    ntt_ms = sum(v["ms"] for k, v in phases.items() if k.startswith("k_ntt") or k.startswith("k_scale"))

    # ---- the first half of BASELINE's metric: create_proof() wall-clock, data-true (real witness, keygen'd pk,
    # Blake2b-derived challenges, device permutation products / evaluate_h / openings / SHPLONK).  Host-inclusive:
    # the transcript and the control flow run on the host between device phases, as they do in the reference.
    create_proof_stats = None
    if shape.name == "standard_plonk" and not args.no_create_proof:
        create_proof_stats = time_create_proof(h2, R, args, dist, backend, torch.device("cuda", local_rank), coll_dev)

    # ---- north_star's ">= 6x MSM speed-up at 8 GPUs" is quoted on the MSM alone: the commitments of one proof queued back to
    # back with ONE join (and one combine), no transforms between them
    msm_only = None
    if not args.no_msm_only:
        msm_only = time_msm_only(h2, R, args, dist, coll_dev, n_devices, torch, sync_all)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # digest of the proof's commitments in canonical affine form: identical for every N (same inputs, same SRS)
    import hashlib

    import numpy as np

    aff = np.zeros((len(commitments), 8), dtype=np.uint64)
    h2._lib.check(lib.h2mi_g1_batch_normalize(np.ascontiguousarray(commitments).ctypes.data, len(commitments), aff.ctypes.data), "normalize")
    digest = hashlib.sha256(aff.tobytes()).hexdigest()
    ms_per_step = elapsed / args.steps * 1e3
    value = adds / (ms_per_step * 1e-3)
    c, W, nb, nreg = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    lib.h2mi_bases_info(R.params.g_handle, C.byref(c), C.byref(W), C.byref(nb), C.byref(nreg))

    # 32 B scalar + 64 B affine base per pair, read once; one k_msm_accum launch sees one slice (a rank's, or
    # in --single-process mode one device's)
    algo_bytes = 96 * (R.n_local // n_devices)
    achieved = algo_bytes / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
    # HBM traffic of the kernel comes from rocprofv3 PMC counters, which cannot be collected from inside this
    # process: the field carries the figure of the latest committed counter run of this same workload and says so
    traffic, traffic_source = None, None
    for tname in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath) and world == 1 and args.k == 20 and args.shape == "standard_plonk":
            try:
                traffic = json.load(open(tpath)).get("k_msm_accum_hbm_bytes_per_launch")
                traffic_source = f"profiles/{tname} (separate rocprofv3 --pmc run of this command, not measured in this run)"
                break
            except Exception:
                traffic = None
    roofline = {
        "kernel": "k_msm_accum",
        "bound": "hbm",
        "achieved": round(achieved, 3),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": algo_bytes,
        "avg_launch_ms": round(accum_ms, 4),
        "note": "MSM is 254-bit integer-multiply bound (v_mad_u64_u32), not HBM bound; see DESIGN.md",
    }

    # the binding resource, stated next to the (contractual) HBM roofline: 32-bit integer multiply issue.
    # One mixed addition = 6 multiplications + 2 squarings + one two-product multiplication with a shared reduction (Y3) in
    # 9 x 29-bit limbs = 6 * 162 + 2 * 126 + 243 = 1467 v_mad_u64_u32 (1548 before round 3's f29_mul2); the peak
    # is the chip's measured v_mad_u64_u32 rate (tools/ubench.hip: 4.82 cycles per wavefront instruction per
    # SIMD at 2.4 GHz = 32.6 T lane-multiplies/s).  Other instructions of the loop (masks, shifts, the m = t * p'
    # products) share the same issue port, so ~0.7 is the practical ceiling of this fraction.
    MADS_PER_MIXED_ADD, MAD_PEAK = 1467, 32.6e12
    entries_per_launch = adds_local / max(shape.msm_per_proof, 1)  # bucket insertions (+ reduction adds) per MSM, this rank
    issue = {
        "kernel": "k_msm_accum",
        "bound": "valu (v_mad_u64_u32 issue)",
        "achieved": round(entries_per_launch * MADS_PER_MIXED_ADD / (accum_ms * 1e-3) / 1e12, 3) if accum_ms else None,
        "peak": MAD_PEAK / 1e12,
        "unit": "T mad/s",
    }
    if issue["achieved"]:
        issue["frac"] = round(issue["achieved"] / issue["peak"], 4)

    out = {
        "metric": "create_proof() wall-clock + MSM G1-adds/s at k=20 standard_plonk, 1/2/4/8 GPU",
        "value": round(value, 1),
        "unit": "G1-adds/s",
        "n_gpus": world * n_devices,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32 (254-bit Montgomery integers as 9 x 29-bit limbs, 64-bit multiply-accumulate)",
        "data": "synthetic",
        "config": {
            "workload": f"{shape.name} hot-path replay k={args.k}: {shape.msm_per_proof} MSM(2^{args.k}) + {shape.ntt_per_proof['intt_n']} iNTT(2^{args.k}) + "
                        f"{shape.ntt_per_proof['coset_ntt_ext']} coset-NTT(2^{R.domain.extended_k}) + 1 coset-iNTT(2^{R.domain.extended_k})",
            "k": args.k,
            "scalar_distribution": args.dist,
            "msm_window_bits": c.value,
            "msm_windows": W.value,
            "msm_buckets": nb.value,
            "parallelism": (f"msm-slice{world}" if world > 1 else f"single-process msm-slice{n_devices}" if n_devices > 1 else "single-gpu"),
            "deployment": ("one process per GPU (torch.distributed)" if world > 1 else
                           f"one process, {n_devices} devices (h2mi_init_devices)" if n_devices > 1 else "one process, one GPU"),
            "combines_per_step": (R.combiner.combines // max(args.warmup + args.steps + 1, 1)) if R.combiner is not None else 0,
            "combine": ((("RCCL all-gather + device fold at every transcript join, device-resident" if R.combiner.mode == "rccl" else
                          "RCCL all-gather through host-synchronised copies (H2MI_COMBINE=host) + device fold at every transcript join")
                         if backend == "nccl" else "gloo all-gather (host) + device fold at every transcript join") if R.combiner is not None
                        else "peer copies of the 96-byte partial points to the primary device + device fold at every join (inside the library)"
                        if n_devices > 1 else "none (single GPU)"),
            "what_is_timed": ("MSM + NTT + evaluate_h kernels on HBM-resident vectors; not transcript / witness generation" if R.with_evaluate_h else
                              "MSM + NTT kernels on HBM-resident vectors; not gate evaluation / transcript / witness generation"),
        },
        "rccl_world_seen": (dist.get_world_size() if dist is not None else 1),
        "collective_backend": (backend if dist is not None else None),
        "ranks": ranks_info,
        "commitments_sha256": digest,
        "ntt_placement": ("single GPU (the primary device)" if world == 1 else "leaf transforms spread round-robin, consumed transforms on every rank"
                          if R.spread else "every rank replays every transform"),
        "g1_adds_per_step": adds,
        "msm_pairs_per_s": round(shape.msm_per_proof * n / (ms_per_step * 1e-3), 1),
        "device_ms_per_step": {"msm": round(msm_ms, 3), "ntt": round(ntt_ms, 3)},
        "kernels": phases,
        "roofline": roofline,
        "issue_roofline": issue,
    }

    if msm_only is not None:
        out.update(msm_only)
    if world == 1 and n_devices == 1 and args.k == 20 and shape.name == "standard_plonk":
        try:
            out["roofline_ntt"] = time_ntt_roofline(h2)
        except Exception as e:  # never lose the headline line to an auxiliary measurement
            out["roofline_ntt"] = {"error": repr(e)}
    if create_proof_stats is not None:
        out["create_proof"] = create_proof_stats
        out["pipeline_ms_per_step"] = create_proof_stats["ms_per_proof"]
        if world == 1 and n_devices == 1 and args.k == 20:
            try:
                small = time_small_proofs(h2, ROOT)
                out["create_proof_k16"], out["create_proof_k8"], out["create_proof_k5"] = small["k16"], small["k8"], small["k5"]
                out["create_proof"]["cpp_host_ms"] = small["k20"].get("cpp_host_ms")
                out["create_proof_hosts_what"] = small["what"]
                if small["k20"].get("cpp_host_ms"):  # the compiled host is the mirror of the reference's (compiled) example: the headline wall clock
                    out["pipeline_ms_per_step"] = small["k20"]["cpp_host_ms"]
                    out["pipeline_ms_per_step_what"] = ("create_proof() wall clock per 2^20-row proof on the C++ host (examples/standard_plonk.cpp, a child process); "
                                                        "create_proof.ms_per_proof is the same prover driven from Python")
            except Exception as e:
                out["create_proof_k16"] = {"error": repr(e)}
            # BASELINE configs[2], [4], [3] data-true on this GPU: create_proof through the halo2-lib builders
            try:
                out["halo2_lib_create_proof"] = time_halo2_lib_examples(h2, R)
            except Exception as e:  # never lose the headline line to an auxiliary measurement
                out["halo2_lib_create_proof"] = {"error": repr(e)}
    parity_ok = True
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(R, args, n)
        par = out["cpu_baseline"].pop("parity")
        out["parity_in_run"] = bool(par["msm_commitment_equal"] and par["ntt_limbs_equal"])
        out["parity_in_run_detail"] = par
        parity_ok = out["parity_in_run"]
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if not parity_ok:  # a fast kernel whose results differ from the oracle's is not a measurement
        sys.stderr.write("bench.py: parity_in_run is FALSE — the GPU's results differ from the CPU restatement's on the same vectors\n")
        sys.exit(3)


def time_msm_only(h2, R, args, dist, coll_dev, n_devices, torch, sync_all):
    """One proof's commitments as MSMs ALONE (no transforms): `msm_per_proof` MSMs over this rank's slice queued back to back,
    one join, one combine of the partial points (N > 1), barrier-bracketed, slowest rank -> `msm_only_ms` per MSM.
    One process per GPU with N > 1 additionally times the SAME MSMs over the WHOLE base set on one GPU inside the same run (every
    rank on its own device at once; rank 0's figure is reported) -> `msm_only_speedup_vs_1` = that / the sliced figure.
    The single-process mode cannot hold an unsharded registration next to the sharded one, so its speed-up is formed against the
    N = 1 line's `msm_only_ms` by the reader (null here)."""
    import ctypes as C

    lib = h2.lib
    world = R.world
    M = R.shape.msm_per_proof
    reps = max(2, min(args.steps, 10))
    src = R.random_poly

    def run_sliced():
        R._slot, R._phase_start = 0, 0
        for _ in range(M):
            R._msm(src, lagrange=True)
        h2._lib.check(lib.h2mi_join(), "join")
        if R.combiner is not None:
            R.combiner.combine(0, M)

    def timed(fn):
        fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / (reps * M) * 1e3

    combines_before = R.combiner.combines if R.combiner is not None else 0
    sliced_ms = timed(run_sliced)
    if R.combiner is not None:
        R.combiner.combines = combines_before  # keep combines_per_step a count of the replay's own joins
    out = {"msm_only_ms": round(sliced_ms, 4), "msm_only_speedup_vs_1": None,
           "msm_only_what": f"{M} MSMs over 2^{args.k} points (each rank / device its 1/{world * n_devices} slice), queued back to back, one join"
                            + (" + one combine" if world * n_devices > 1 else "") + ", per MSM, slowest rank"}
    if world > 1:
        from halo2_scaffold_amd.params import ParamsKZG

        full = ParamsKZG.setup(args.k, 0x5EC2E7 + 0x48324D49, register=False)
        hfull = C.c_uint64()
        h2._lib.check(lib.h2mi_bases_register_dev(full._gl_dev.ptr, full.n, C.byref(hfull)), "register full g_lagrange")

        def run_full():
            for i in range(M):
                h2._lib.check(lib.h2mi_msm_bn254_g1_dev(hfull.value, src.ptr, R.n, R.out_ptr + 96 * i, None), "msm")
            h2._lib.check(lib.h2mi_join(), "join")

        one_ms = timed(run_full)
        h2._lib.check(lib.h2mi_sync(), "sync")
        h2._lib.check(lib.h2mi_bases_release(hfull.value), "release")
        full.release()
        out["msm_only_1gpu_ms"] = round(one_ms, 4)
        out["msm_only_speedup_vs_1"] = round(one_ms / sliced_ms, 3)
        out["msm_only_what"] += "; msm_only_1gpu_ms: the same MSMs over all 2^%d points, every rank on its own GPU at once, in this run" % args.k
    return out


def rank_device_report(torch, dist, rank, local_rank, world):
    """every rank's device ordinal and PCI bus id, gathered to all ranks; two ranks on one device without the H2MI_DEVICE
    rehearsal knob is a launch mistake (each rank would time half a GPU): fail loudly on every rank."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        name = pr.name
    except Exception as e:  # no GPU visible: the library's init reports that properly a moment later
        bus, name = "unknown", repr(e)
    mine = {"rank": rank, "device": local_rank, "pci_bus_id": bus, "name": name, "visible_devices": torch.cuda.device_count()}
    if dist is None:
        return [mine]
    info = [None] * dist.get_world_size()
    dist.all_gather_object(info, mine)
    seen = {}
    for r in info:
        seen.setdefault((r["device"], r["pci_bus_id"]), []).append(r["rank"])
    shared = {k: v for k, v in seen.items() if len(v) > 1}
    if shared and "H2MI_DEVICE" not in os.environ:
        raise SystemExit(f"ranks share a GPU {shared}: launch one rank per device (LOCAL_RANK), or set H2MI_DEVICE=<i> for a one-GPU rehearsal")
    return info


def time_ntt_roofline(h2):
    """the NTT kernels alone: per-transform launch time of k_ntt_pass_* from HIP events on the library's
    launch stream, against both bounds — HBM on the algorithmic 64 n bytes per transform and 32-bit multiply
    issue on the transform's field multiplications (162 v_mad_u64_u32 each; tools/ubench.hip: 32.6 T mad/s)."""
    import ctypes as C

    import numpy as np

    from halo2_scaffold_amd import field as F
    from halo2_scaffold_amd.device import DevBuf

    lib = h2.lib

    def mults_per_element(log_n, coset):  # csrc/h2mi_ntt.hip choose_split + local_ntt: radix-4 rounds, the first one nearly free
        if log_n <= 10:
            ms = [log_n]
        elif log_n <= 20:
            ms = [(log_n + 1) // 2, log_n - (log_n + 1) // 2]
        else:
            m0 = max(8, (log_n + 2) // 3)
            rem = log_n - m0
            m1 = max(min(8, rem - 4), (rem + 1) // 2)
            ms = [m0, m1, rem - m1]
        per = sum((m // 2 - 1) + 0.25 + (0.5 if m & 1 else 0) for m in ms)
        return per + (len(ms) - 1) + (1 if coset else 0), ms

    out = {}
    for label, log_n, coset in (("ntt_2^20", 20, False), ("coset_ntt_2^21", 21, True), ("ntt_2^24", 24, False)):
        n = 1 << log_n
        buf = DevBuf(n * 32)
        h2._lib.check(lib.h2mi_fr_random_dev(buf.ptr, n, 0x4E5454, 0, None), "random")
        omega = F.fr_to_mont_limbs(F.omega_for(log_n))
        pre = F.fr_to_mont_limbs(F.FR_ZETA) if coset else None
        run = lambda: h2._lib.check(lib.h2mi_ntt_bn254_fr_dev(buf.ptr, log_n, omega.ctypes.data, pre.ctypes.data if coset else None, None, None), "ntt")
        for _ in range(3):
            run()
        h2._lib.check(lib.h2mi_sync(), "sync")
        reps = 10 if log_n <= 21 else 4
        lib.h2mi_profile_reset()
        lib.h2mi_profile_filter(b"k_ntt_pass")
        lib.h2mi_profile_enable(1)
        for _ in range(reps):
            run()
        h2._lib.check(lib.h2mi_sync(), "sync")
        lib.h2mi_profile_enable(0)
        tot, cnt = C.c_double(), C.c_uint64()
        lib.h2mi_profile_query(b"k_ntt_pass", C.byref(tot), C.byref(cnt))
        lib.h2mi_profile_reset()
        buf.free()
        us = tot.value / reps * 1e3
        mpe, ms = mults_per_element(log_n, coset)
        gbs = 64 * n / (us * 1e-6) / 1e9
        mads = mpe * 162 * n / (us * 1e-6) / 1e12
        out[label] = {"us_per_transform": round(us, 1), "launches_per_transform": cnt.value // reps, "passes": ms,
                      "algorithmic_bytes": 64 * n, "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
                      "mad_issue": {"field_mults_per_element": mpe, "achieved": round(mads, 2), "peak": 32.6, "unit": "T mad/s", "frac": round(mads / 32.6, 4)}}
    out["bound"] = "valu (v_mad_u64_u32 issue): see DESIGN.md 4.2; the HBM fraction is the contractual figure on 64 n algorithmic bytes"
    return out


def time_small_proofs(h2, root):
    """create_proof() at the other BASELINE sizes (configs[1]: DEGREE 16; configs[0]: DEGREE 8 — CPU-only in BASELINE, here as
    the launch-latency floor), steady state, host-inclusive wall clock: the Python host in this process and the C++ host
    (examples/standard_plonk, the compiled mirror of the reference's Rust example) as a child process."""
    import subprocess

    from halo2_scaffold_amd import circuits, keygen, prover
    from halo2_scaffold_amd.params import ParamsKZG

    lib = h2.lib
    out = {}
    exe = os.path.join(root, "examples", "standard_plonk")
    for k in (16, 8, 5, 20):  # 5: the reference's own example run (examples/standard_plonk.rs:26)
        entry = {}
        if k != 20:  # the k = 20 Python-host figure is the create_proof block itself
            params = ParamsKZG.setup(k, 0x5EC2E7 + 0x48324D49)
            c = circuits.StandardPlonk(None)
            pk = keygen.keygen_pk(params, keygen.keygen_vk(params, c), c)
            ws = prover.ProverWorkspace(params, pk)
            for i in range(3):
                prover.create_proof(params, pk, circuits.StandardPlonk(5 + i), 10 + i, ws=ws)
            h2._lib.check(lib.h2mi_sync(), "sync")
            times = []
            for i in range(20):
                t0 = time.perf_counter()
                prover.create_proof(params, pk, circuits.StandardPlonk(50 + i), 100 + i, ws=ws)
                times.append(time.perf_counter() - t0)
            ws.release()
            pk.release()
            params.release()
            entry["python_host_ms"] = round(1e3 * sum(times) / len(times), 3)
            entry["python_host_min_ms"] = round(1e3 * min(times), 3)
        try:
            r = subprocess.run([exe, str(k)], capture_output=True, text=True, timeout=300, env=dict(os.environ, H2MI_PROOFS="20" if k < 20 else "10"))
            line = next(l for l in r.stdout.splitlines() if l.startswith("steady_ms_per_proof"))
            entry["cpp_host_ms"] = round(float(line.split()[1]), 3)
        except Exception as e:  # the C++ example is optional evidence: never lose the headline line to it
            entry["cpp_host_error"] = repr(e)[:200]
        out[f"k{k}"] = entry
    out["what"] = ("steady-state create_proof() wall clock per proof, host-inclusive; cpp_host = examples/standard_plonk.cpp over include/h2mi_plonk.hpp "
                   "(same C ABI, same proof bytes)")
    return out


def time_halo2_lib_examples(h2, R):
    """scaffold::prove's create_proof (src/scaffold.rs:322-331) for the reference's halo2-lib example closures, data-true on
    one GPU: halo2_lib (x^2 + 72) and poseidon hash_two at DEGREE = 20 against the replay's SRS, range_check with
    LOOKUP_BITS = 16 at DEGREE = 22 against its own.  Steady-state wall clock per proof (workspace reused, fresh witness and
    seed per proof), host-inclusive; witness generation (the closure) is outside the timed span, as keygen is."""
    import hashlib

    from halo2_scaffold_amd import flex, poseidon
    from halo2_scaffold_amd.params import ParamsKZG

    lib = h2.lib

    def run(params, cs, closure, proofs):
        keys = flex.FlexKeys(params, cs, closure(3))
        ws = flex.FlexWorkspace(params, keys)
        proof = flex.create_proof(params, keys, closure(3), 1, ws=ws)
        h2._lib.check(lib.h2mi_sync(), "sync")
        times = []
        for i in range(proofs):
            asg = closure(1000 + i)
            t0 = time.perf_counter()
            proof = flex.create_proof(params, keys, asg, 100 + i, ws=ws)
            times.append(time.perf_counter() - t0)
        ws.release()
        keys.release()
        return {"ms_per_proof": round(1e3 * sum(times) / len(times), 3), "min_ms": round(1e3 * min(times), 3), "proofs": proofs,
                "proof_bytes": len(proof), "last_proof_sha256": hashlib.sha256(proof).hexdigest()}

    gate, rng = flex.FlexGateCS(lookup=False), flex.FlexGateCS(lookup=True)
    out = {"halo2_lib_k20": run(R.params, gate, lambda x: flex.halo2_lib_closure(gate, x), 5),
           "poseidon_k20": run(R.params, gate, lambda x: poseidon.hash_two_closure(gate, x, x + 1), 5)}
    big = ParamsKZG.setup(22, 0x5EC2E7 + 0x48324D49)
    out["range_lookup16_k22"] = run(big, rng, lambda x: flex.range_closure(rng, x, 16), 3)
    big.release()
    # the same three proofs through the C++ host (examples/halo2_lib.cpp over include/h2mi_flex.hpp: same C ABI, same proof bytes —
    # tests/test_gpu_big_golden.py), as a child process; optional evidence, never allowed to lose the headline line
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples", "halo2_lib")
    for name, shape, k, bits, proofs in (("halo2_lib_k20", "halo2_lib", 20, 0, 8), ("poseidon_k20", "poseidon", 20, 0, 8), ("range_lookup16_k22", "range", 22, 16, 4)):
        try:
            r = subprocess.run([exe, shape, str(k), str(bits), "77", hex(0x5EC2E7 + 0x48324D49), "1"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, H2MI_PROOFS=str(proofs)))
            line = next(l for l in r.stdout.splitlines() if l.startswith("steady_ms_per_proof"))
            out[name]["cpp_host_ms"] = round(float(line.split()[1]), 3)
        except Exception as e:
            out[name]["cpp_host_error"] = repr(e)[:200]
    return out


def time_create_proof(h2, R, args, dist, backend, torch_device, coll_dev):
    """keygen once, then `steps` proofs of the reference's circuit (examples/standard_plonk.rs:33-50) through one
    workspace; every proof uses a fresh witness and rng seed.  Wall-clock around create_proof(), proof bytes out.
    N > 1 (one process per GPU): every rank runs the prover; each commitment is the rank's slice MSM and the partial
    points are combined at every transcript join (RCCL all-gather + device fold), so all ranks draw the same challenges
    and hold the same proof; transforms, quotient and openings are replicated (NTT is single-GPU by design)."""
    import hashlib

    import torch

    from halo2_scaffold_amd import circuits, keygen, prover
    from halo2_scaffold_amd.params import ParamsKZG

    lib = h2.lib
    circuit = circuits.StandardPlonk(None)
    combiner = None
    params = R.params
    t0 = time.perf_counter()
    if dist is not None:  # keygen against the whole SRS (the vk must be the full commitments), then commit by slice
        from halo2_scaffold_amd.dist import PhaseCombiner

        full = ParamsKZG.setup(args.k, 0x5EC2E7 + 0x48324D49)
        vk = keygen.keygen_vk(full, circuit)
        pk = keygen.keygen_pk(full, vk, circuit)
        h2._lib.check(lib.h2mi_sync(), "sync")
        full.release()
        combiner = PhaseCombiner(4, backend, torch_device)
    else:
        vk = keygen.keygen_vk(params, circuit)
        pk = keygen.keygen_pk(params, vk, circuit)
    h2._lib.check(lib.h2mi_sync(), "sync")
    keygen_s = time.perf_counter() - t0
    ws = prover.ProverWorkspace(params, pk, combiner=combiner)
    proofs_run = 0

    def barrier():
        h2._lib.check(lib.h2mi_sync(), "sync")
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    proof = b""
    for i in range(max(args.warmup, 1)):
        proof = prover.create_proof(params, pk, circuits.StandardPlonk(0x1234 + i), 1000 + i, ws=ws)
    barrier()
    times = []
    for i in range(args.steps):
        t0 = time.perf_counter()
        proof = prover.create_proof(params, pk, circuits.StandardPlonk(0xABCDEF + i), 2000 + i, ws=ws)
        times.append(time.perf_counter() - t0)
    barrier()
    if dist is not None:  # the slowest rank's mean
        t = torch.tensor([sum(times) / len(times)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(t.item())] * len(times)
    # device-only share of one proof (all kernels bracketed by events; untimed extra pass)
    import ctypes as C

    lib.h2mi_profile_reset()
    lib.h2mi_profile_filter(b"")
    lib.h2mi_profile_enable(1)
    prover.create_proof(params, pk, circuits.StandardPlonk(7), 7, ws=ws)
    lib.h2mi_profile_enable(0)
    tot, cnt = C.c_double(), C.c_uint64()
    lib.h2mi_profile_query(b"", C.byref(tot), C.byref(cnt))
    kernels = {}
    for name in ["k_msm", "k_scan", "k_ntt", "k_scale_powers", "k_evaluate_h", "k_perm", "k_mulscan", "k_fr_inv_one", "k_eval_poly", "k_sum_fe",
                 "k_kate", "k_lincomb", "k_fr_random", "k_fr_add_head", "k_g1_normalize", "k_pow_table"]:
        t_, c_ = C.c_double(), C.c_uint64()
        lib.h2mi_profile_query(name.encode(), C.byref(t_), C.byref(c_))
        kernels[name] = {"ms": round(t_.value, 4), "launches": c_.value}
    lib.h2mi_profile_reset()
    tr = {}
    prover.create_proof(params, pk, circuits.StandardPlonk(8), 8, ws=ws, trace=tr)  # host-side phase boundaries (each ends in a device sync)
    phase_ms = tr.get("phase_ms")
    ws.release()
    pk.release()
    combines_per_proof = 0
    if combiner is not None:  # counted, not assumed: every create_proof above went through this combiner
        proofs_run = max(args.warmup, 1) + args.steps + 2
        combines_per_proof = combiner.combines // proofs_run
        combiner.release()
    times.sort()
    return {
        "ms_per_proof": round(sum(times) / len(times) * 1e3, 3),
        "min_ms": round(times[0] * 1e3, 3),
        "proofs": len(times),
        "combines_per_proof": combines_per_proof,
        "proof_bytes": len(proof),
        "last_proof_sha256": hashlib.sha256(proof).hexdigest(),
        "keygen_vk_pk_seconds": round(keygen_s, 3),
        "device_kernel_ms_sum": round(tot.value, 3),
        "kernel_launches": cnt.value,
        "kernels_ms": kernels,
        "phase_ms": phase_ms,
        "what": ("create_proof() of the reference's StandardPlonk circuit at 2^%d rows: real witness and copy constraints, keygen'd proving key, "
                 "Blake2b transcript on the host, 11 MSM + 13 NTT + permutation products + evaluate_h + 21 evaluations + SHPLONK on the device; "
                 "wall-clock, host-inclusive; rng = seeded SplitMix64 (the reference uses OsRng); byte-identical to the in-repo CPU oracle (golden proof at this "
                 "size: tests/test_gpu_big_golden.py), NOT interoperable with the reference's verify_proof: vk.transcript_repr is a stand-in for the "
                 "crate's hash of the pinned key's Debug text, so every challenge differs from the Rust prover's" % args.k),
    }


def host_threads():
    """threads this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box hands
    each job a share of the host, e.g. 16 of 256 hardware threads)."""
    t = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            t = min(t, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(t, 64))


def cpu_baseline(R, args, n):
    """the C restatement of best_multiexp / best_fft (oracle/h2ref.c) timed on this host's cores."""
    import math

    import numpy as np

    from oracle import cref

    try:
        cref.use_native()  # compile the restatement for this host's CPU
    except Exception:
        pass  # fall back to the portable build
    threads = host_threads()
    ls = args.cpu_sample_log if args.cpu_sample_log is not None else min(args.k, 20)
    ns = 1 << ls
    scal = R.cols[0].to_numpy(shape=(n, 4))[:ns].copy()
    bases = R.params.get_g_lagrange()[:ns].copy()
    cref.msm(scal[:1024], bases[:1024], threads)  # warm the thread pool / page in
    t0 = time.perf_counter()
    cpu_point = cref.msm(scal, bases, threads)
    t_msm = time.perf_counter() - t0
    # G1 additions of the reference algorithm on this sample (per thread chunk: window c = ceil(ln chunk))
    chunk = max(ns // threads, 1) if ns > threads else ns
    cc = 1 if chunk < 4 else 3 if chunk < 32 else math.ceil(math.log(chunk))
    segments = 256 // cc + 1
    nchunks = math.ceil(ns / chunk)
    used_segments = math.ceil(254 / cc)
    cpu_adds = ns * used_segments * (1 - 2.0 ** -cc) + nchunks * segments * 2 * ((1 << cc) - 1)
    from halo2_scaffold_amd import field as F

    w = F.fr_to_mont_limbs(F.omega_for(ls))
    a = scal.copy()
    t0 = time.perf_counter()
    cref.ntt(a, w, ls, threads)
    t_ntt = time.perf_counter() - t0
    # parity in the run (the reference's only check is in-run too: verify_proof after create_proof, src/scaffold.rs:354-361): the
    # GPU's commitment and transform of the SAME vectors the CPU restatement just processed must be the same group element / the
    # same limbs.  The oracle is the checker here, never the thing measured.
    import ctypes as C

    from halo2_scaffold_amd.device import DevBuf

    import _load_pkg

    h2 = _load_pkg.load()
    d_out = DevBuf(96)
    h2._lib.check(h2.lib.h2mi_msm_bn254_g1_dev(R.params.g_lagrange_handle, R.cols[0].ptr, ns, d_out.ptr, None), "parity msm")
    gpu_point = d_out.to_numpy(shape=(12,))
    d_out.free()
    msm_equal = bool(np.array_equal(cref.normalize(gpu_point), cref.normalize(cpu_point)))
    g = scal.copy()
    h2.best_fft(g, w, ls)
    ntt_equal = bool(np.array_equal(g, a))
    parity = {"msm_commitment_equal": msm_equal, "ntt_limbs_equal": ntt_equal,
              "what": f"GPU MSM(2^{ls}) and NTT(2^{ls}) of the vectors the CPU baseline just processed, against oracle/h2ref.c's results"}
    with open("/proc/cpuinfo") as f:
        model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    return {
        "value": round(cpu_adds / t_msm, 1),
        "unit": "G1-adds/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 MSM(2^{ls}) + 1 NTT(2^{ls}) of the same synthetic vectors, C restatement of best_multiexp/best_fft "
                  f"(oracle/h2ref.c), {threads} threads, window c={cc}",
        "cpu_model": model,
        "msm_seconds": round(t_msm, 4),
        "msm_pairs_per_s": round(ns / t_msm, 1),
        "ntt_seconds": round(t_ntt, 4),
        "projected_step_seconds": round(R.shape.msm_per_proof * t_msm * (n / ns)
                                        + (R.shape.ntt_per_proof["intt_n"] + (R.domain.extended_len() // n) * (R.shape.ntt_per_proof["coset_ntt_ext"] + 1)) * t_ntt * (n / ns), 3),
        "label": "restated CPU baseline (C), not the Rust binary",
        "parity": parity,
    }


if __name__ == "__main__":
    main()
