"""N back-to-back device-resident MSMs of 2^k points then one sync, issued singly and in groups of G through h2mi_msm_bn254_g1_phase_dev
(an 8-GPU rank's commitment stream: 2^17 slices of a 2^20 proof).  Usage: msm_b2b_batch.py K [N] [G]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = 1 << k
p = h2.ParamsKZG.setup(k, 0x1234567)
sc = [h2.DevBuf.from_numpy(synth.uniform_fr(n, 5 + i)) for i in range(4)]
out = h2.DevBuf(96 * N)
single = lambda: [lib.h2mi_msm_bn254_g1_dev(p.g_handle, sc[i % 4].ptr, n, out.ptr + 96 * i, None) for i in range(N)]
def grouped():
    for i0 in range(0, N, G):
        m = min(G, N - i0)
        ptrs = (C.c_void_p * m)(*[sc[(i0 + j) % 4].ptr for j in range(m)])
        assert lib.h2mi_msm_bn254_g1_phase_dev(p.g_handle, ptrs, m, n, out.ptr + 96 * i0, 0, None) == 0
for name, run in (("single", single), (f"groups of {G}", grouped), ("single", single), (f"groups of {G}", grouped)):
    for _ in range(3):
        run(); lib.h2mi_sync()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    lib.h2mi_sync()
    print(f"k={k} {N} MSMs back to back, {name}: {(time.perf_counter() - t0) / N / 5 * 1e6:.1f} us per MSM", flush=True)
