"""Timing sweep for the NTT (device-resident), used for tuning tile/thread parameters via env vars."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import field as F, synth
lib = h2.lib
for log_n in [int(x) for x in sys.argv[1:]] or [16, 20, 21, 22]:
    n = 1 << log_n
    d = h2.DevBuf.from_numpy(synth.uniform_fr(n, 5))
    w = F.fr_to_mont_limbs(F.omega_for(log_n))
    for _ in range(3): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, None, None, None)
    lib.h2mi_sync(); t0 = time.perf_counter(); R = 20
    for _ in range(R): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, None, None, None)
    lib.h2mi_sync(); dt = (time.perf_counter() - t0) / R
    # coset form (pre-scale by ZETA^i fused into the first pass) and inverse form (n^-1 post-scale fused into the last)
    z = F.fr_to_mont_limbs(F.FR_ZETA); ninv = F.fr_to_mont_limbs(F.fr_inv(n))
    for _ in range(3): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, z.ctypes.data, None, None)
    lib.h2mi_sync(); t0 = time.perf_counter()
    for _ in range(R): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, z.ctypes.data, None, None)
    lib.h2mi_sync(); dtc = (time.perf_counter() - t0) / R
    for _ in range(3): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, None, ninv.ctypes.data, None)
    lib.h2mi_sync(); t0 = time.perf_counter()
    for _ in range(R): lib.h2mi_ntt_bn254_fr_dev(d.ptr, log_n, w.ctypes.data, None, ninv.ctypes.data, None)
    lib.h2mi_sync(); dti = (time.perf_counter() - t0) / R
    print(f"log_n={log_n} plain {dt*1e6:8.1f} us ({64*n/dt/1e9:6.1f} GB/s algorithmic)  coset {dtc*1e6:8.1f} us  post-scaled {dti*1e6:8.1f} us  env={ {k:v for k,v in os.environ.items() if k.startswith('H2MI')} }", flush=True)
    d.free()
