#!/usr/bin/env python3
"""Per-operation benchmark: MSM and NTT at the BASELINE sizes, median of >= 10 runs after 3
warm-ups; device-resident time (HIP-event-free: synchronised wall clock around the async call), wall
clock around the host-pointer C-ABI call (PCIe-inclusive), and the C restatement of the reference's CPU
algorithms with T = 1 and T = all usable host threads.  Prints one JSON document."""
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import _load_pkg

h2 = _load_pkg.load()
h2.init(0)
from bench import host_threads  # noqa: E402
from halo2_scaffold_amd import field as F  # noqa: E402
from halo2_scaffold_amd import synth  # noqa: E402
from oracle import cref  # noqa: E402

lib = h2.lib
REPS = 10


def med(fn, reps=REPS, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    ks = [int(x) for x in sys.argv[1:]] or [8, 16, 20, 22]
    T = host_threads()
    with open("/proc/cpuinfo") as f:
        model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    out = {"cpu_model": model, "cpu_threads_usable": T, "reps": REPS, "msm": [], "ntt": [],
           "note": "gpu_dev_us: vectors resident in HBM; gpu_host_us: host pointers through the C ABI (H2D + D2H included); "
                   "cpu_*: oracle/h2ref.c (C restatement of best_multiexp / best_fft), not the Rust binary"}
    kmax = max(ks)
    full = h2.ParamsKZG.setup(kmax, 0x48324D49)
    g_all = full.get_g()
    full.release()
    for k in ks:
        n = 1 << k
        p = h2.ParamsKZG.from_bases(k, g_all[:n])
        for dist in ("uniform", "witness"):
            sc = synth.uniform_fr(n, synth.SEED) if dist == "uniform" else synth.witness_like_fr(n, synth.SEED)
            d_sc = h2.DevBuf.from_numpy(sc)
            d_out = h2.DevBuf(96)

            def dev():
                p.commit_dev(d_sc, d_out)
                lib.h2mi_sync()

            t_dev = med(dev)
            t_host = med(lambda: p.commit(sc))
            row = {"k": k, "scalars": dist, "gpu_dev_us": round(t_dev * 1e6, 1), "gpu_host_us": round(t_host * 1e6, 1),
                   "gpu_pairs_per_s": round(n / t_dev, 1)}
            if dist == "uniform":
                ns = min(n, 1 << 18)  # bounded CPU sample, scaled linearly
                t1 = med(lambda: cref.msm(sc[:ns], g_all[:ns], 1), reps=1, warm=0) * (n / ns)
                tT = med(lambda: cref.msm(sc[:n], g_all[:n], T), reps=3, warm=1) if n <= (1 << 20) else med(
                    lambda: cref.msm(sc[: 1 << 20], g_all[: 1 << 20], T), reps=3, warm=1) * (n / (1 << 20))
                row.update({"cpu_1thread_us": round(t1 * 1e6, 1), f"cpu_{T}threads_us": round(tT * 1e6, 1),
                            "cpu_sample": f"T=1 on 2^{ns.bit_length() - 1} pairs scaled; T={T} on min(n, 2^20) pairs scaled"})
            out["msm"].append(row)
            d_sc.free()
            d_out.free()
        p.release()
    for log_n in sorted(set(ks + [21, 24] if 20 in ks or 22 in ks else ks)):
        n = 1 << log_n
        a = synth.uniform_fr(n, synth.SEED + 1)
        w = F.fr_to_mont_limbs(F.omega_for(log_n))
        winv = F.fr_to_mont_limbs(F.fr_inv(F.omega_for(log_n)))
        ninv = F.fr_to_mont_limbs(F.fr_inv(n))
        zeta = F.fr_to_mont_limbs(F.FR_ZETA)
        d = h2.DevBuf.from_numpy(a)

        def fwd():
            lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, None, None, None)
            lib.h2mi_sync()

        def inv():
            lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, winv.ctypes.data, None, ninv.ctypes.data, None)
            lib.h2mi_sync()

        def coset():
            lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, zeta.ctypes.data, None, None)
            lib.h2mi_sync()

        t_f, t_i, t_c = med(fwd), med(inv), med(coset)
        host = a.copy()
        t_h = med(lambda: lib.h2mi_ntt_bn254_fr(host.ctypes.data, w.ctypes.data, log_n), reps=5)
        row = {"log_n": log_n, "gpu_forward_us": round(t_f * 1e6, 1), "gpu_inverse_scaled_us": round(t_i * 1e6, 1),
               "gpu_coset_us": round(t_c * 1e6, 1), "gpu_host_us": round(t_h * 1e6, 1),
               "algorithmic_GBps": round(64 * n / t_f / 1e9, 1), "hbm_roofline_frac": round(64 * n / t_f / 8e12, 4)}
        if log_n <= 22:
            c1 = a.copy()
            t1 = med(lambda: cref.ntt(c1, w, log_n, 1), reps=1, warm=0)
            tT = med(lambda: cref.ntt(c1, w, log_n, T), reps=3, warm=1)
            row.update({"cpu_1thread_us": round(t1 * 1e6, 1), f"cpu_{T}threads_us": round(tT * 1e6, 1)})
        out["ntt"].append(row)
        d.free()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
