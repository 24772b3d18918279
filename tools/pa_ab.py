"""Pair-affine accumulation (H2MI_MSM_PA, -DH2MI_AB library) against the product's XYZZ accumulation: lone-MSM latency and back-to-back
throughput at 2^k points with uniform scalars, and the device time of every MSM kernel of one call.  Run once per setting:
    H2MI_LIBRARY=halo2-scaffold_amd/libh2mi_ab.so [H2MI_MSM_PA=1] python tools/pa_ab.py 20"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
ks = [int(x) for x in sys.argv[1:]] or [20]
tag = "pair-affine" if os.environ.get("H2MI_MSM_PA") else "xyzz       "
for k in ks:
    n = 1 << k
    params = h2.ParamsKZG.setup(k, 0x1234567)
    hreg = params.g_handle
    sc = [h2.DevBuf.from_numpy(synth.uniform_fr(n, 5 + i)) for i in range(4)]
    out = h2.DevBuf(96 * 64)
    run = lambda i: lib.h2mi_msm_bn254_g1_dev(hreg, sc[i % 4].ptr, n, out.ptr + 96 * (i % 64), None)
    for i in range(6): assert run(i) == 0
    lib.h2mi_sync()
    ref = out.to_numpy(shape=(64, 12)).copy()
    R = 24
    t0 = time.perf_counter()
    for i in range(R):
        run(i); lib.h2mi_sync()
    lat = (time.perf_counter() - t0) / R
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(R): run(i)
        lib.h2mi_sync()
        best = min(best, (time.perf_counter() - t0) / R)
    lib.h2mi_profile_reset(); lib.h2mi_profile_filter(b""); lib.h2mi_profile_enable(1)
    run(0); lib.h2mi_sync(); lib.h2mi_profile_enable(0)
    buf = C.create_string_buffer(1 << 16); need = C.c_size_t()
    lib.h2mi_profile_dump(buf, len(buf), C.byref(need))
    parts = {}
    for line in buf.value.decode().splitlines():
        name, _, ms = line.split()
        parts[name.replace("k_msm_", "")] = parts.get(name.replace("k_msm_", ""), 0) + round(float(ms) * 1e3, 1)
    lib.h2mi_profile_reset()
    import hashlib
    print(f"k={k} {tag} lone {lat*1e3:7.3f} ms  back-to-back {best*1e3:7.3f} ms  results {hashlib.sha256(ref[:4].tobytes()).hexdigest()[:12]}  kernels(us) {parts}", flush=True)
    for b in sc: b.free()
    out.free(); params.release()
