"""Timing sweep for the device-resident MSM (one call at a time, synchronised): latency vs size.  H2MI_MSM_C=<c> in the
environment fixes the window width (tools/msm_c_sweep.sh runs this once per width)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
ks = [int(x) for x in sys.argv[1:]] or [14, 16, 17, 18, 20]
kmax = max(ks)
full = h2.ParamsKZG.setup(kmax, 0x1234567)
g = full.get_g()
full.release()
for k in ks:
    n = 1 << k
    p = h2.ParamsKZG.from_bases(k, g[:n])
    sc = h2.DevBuf.from_numpy(synth.uniform_fr(n, 5))
    out = h2.DevBuf(96 * 4)
    c, W, nb, nn = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    lib.h2mi_bases_info(p.g_handle, C.byref(c), C.byref(W), C.byref(nb), C.byref(nn))
    for _ in range(3): p.commit_dev(sc, out)
    lib.h2mi_sync()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R):
        p.commit_dev(sc, out); lib.h2mi_sync()
    lat = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for _ in range(R): p.commit_dev(sc, out)
    lib.h2mi_sync()
    thr = (time.perf_counter() - t0) / R
    lib.h2mi_profile_reset(); lib.h2mi_profile_filter(b""); lib.h2mi_profile_enable(1)
    p.commit_dev(sc, out); lib.h2mi_sync(); lib.h2mi_profile_enable(0)
    parts = {}
    tot, cnt = C.c_double(), C.c_uint64()
    for name in ["k_msm_digits", "hipcub_radix", "k_msm_bounds", "k_msm_bin_count", "binscan_hipcub", "k_msm_bin_scatter", "k_msm_bin_sort", "hipcub_scan", "k_scan_segsum", "k_scan_seg_bins", "k_scan_seg_tasks", "k_msm_accum", "k_msm_fold", "k_msm_finish", "k_msm_seg", "k_msm_rowcol", "k_msm_weighted", "k_msm_final"]:
        lib.h2mi_profile_query(name.encode(), C.byref(tot), C.byref(cnt)); 
        if cnt.value: parts[name.replace("k_msm_", "")] = round(tot.value * 1e3)
    lib.h2mi_profile_reset()
    print(f"k={k} c={c.value} W={W.value} nb={nb.value}: latency {lat*1e6:8.1f} us  back-to-back {thr*1e6:8.1f} us  kernels(us) {parts}", flush=True)
    p.release(); sc.free(); out.free()
