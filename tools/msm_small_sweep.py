"""Small-set MSM: latency (one call, synchronised), back-to-back throughput and the latency of a prover-phase-like batch (four
MSMs, one join) for n = 2^5 .. 2^17, as the library chooses the path by itself (`default`: the latency path, switching to the general
pipeline for base sets above 2^12 points once four MSMs are queued without a join) and with the general pipeline forced
(H2MI_MSM_GENERAL), with the per-kernel device times of one call.  With the -DH2MI_AB library
(H2MI_LIBRARY=halo2-scaffold_amd/libh2mi_ab.so) H2MI_MSM_NO_AUTO_STREAM=1 keeps the latency path whatever the queue depth (the third
column of the comparison) and H2MI_MSM_SMALL_C / H2MI_MSM_SMALL_R sweep the window width and the cells per lane.
Usage: msm_small_sweep.py [k ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
ks = [int(x) for x in sys.argv[1:]] or [5, 8, 10, 12, 13]
full = h2.ParamsKZG.setup(max(ks), 0x1234567)
g = full.get_g()
full.release()
tag = f"small_c={os.environ.get('H2MI_MSM_SMALL_C', 'default')} small_r={os.environ.get('H2MI_MSM_SMALL_R', 'default')}"
for k in ks:
    n = 1 << k
    hreg = C.c_uint64()
    bases = np.ascontiguousarray(g[:n])
    assert lib.h2mi_bases_register(bases.ctypes.data, n, C.byref(hreg)) == 0
    sc = [h2.DevBuf.from_numpy(synth.uniform_fr(n, 5 + i)) for i in range(4)]
    out = h2.DevBuf(96 * 4)
    ptrs = [(C.c_void_p * 1)(b.ptr) for b in sc]
    pinned = os.environ.get("H2MI_MSM_NO_AUTO_STREAM") and os.environ.get("H2MI_LIBRARY")
    for small in (1, 0):
        flags = 0 if small else 4  # H2MI_MSM_GENERAL
        run = lambda i=0: lib.h2mi_msm_bn254_g1_phase_dev(hreg.value, ptrs[i], 1, n, out.ptr + 96 * i, flags, None)
        for _ in range(5): run()
        lib.h2mi_sync()
        R = 40
        t0 = time.perf_counter()
        for _ in range(R):
            run(); lib.h2mi_sync()
        lat = (time.perf_counter() - t0) / R
        t0 = time.perf_counter()
        for _ in range(R): run()
        lib.h2mi_sync()
        thr = (time.perf_counter() - t0) / R
        t0 = time.perf_counter()
        for _ in range(R):
            for i in range(4): run(i)
            lib.h2mi_msm_flush(); lib.h2mi_sync()
        phase = (time.perf_counter() - t0) / R
        lib.h2mi_profile_reset(); lib.h2mi_profile_filter(b""); lib.h2mi_profile_enable(1)
        run(); lib.h2mi_sync(); lib.h2mi_profile_enable(0)
        buf = C.create_string_buffer(1 << 16); need = C.c_size_t()
        lib.h2mi_profile_dump(buf, len(buf), C.byref(need))
        parts, t_first, t_last = {}, None, 0.0
        for line in buf.value.decode().splitlines():
            name, t0_, ms = line.split()
            parts[name.replace("k_msm_", "")] = parts.get(name.replace("k_msm_", ""), 0) + round(float(ms) * 1e3, 1)
            t_first = float(t0_) if t_first is None else t_first
            t_last = max(t_last, float(t0_) + float(ms))
        lib.h2mi_profile_reset()
        span = (t_last - (t_first or 0)) * 1e3
        print(f"k={k:2d} {('small path' if pinned else 'default   ') if small else 'general   '} [{tag}] latency {lat*1e6:7.1f} us  back-to-back {thr*1e6:7.1f} us  "
              f"4 MSMs + join {phase*1e6:7.1f} us  device span {span:6.1f} us  kernels(us) {parts}", flush=True)
    lib.h2mi_bases_release(hreg.value)
    for b in sc: b.free()
    out.free()
