"""N back-to-back device-resident MSMs of 2^k points then one sync (an 8-GPU rank's commitment stream: 2^17 slices of a 2^20
proof) — the program to put behind `rocprofv3 --kernel-trace --` (tools/msm_b2b_trace.sh).  Prints µs per MSM.
Usage: msm_b2b.py K [N]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
N = int(sys.argv[2]) if len(sys.argv) > 2 else 22
n = 1 << k
p = h2.ParamsKZG.setup(k, 0x1234567)
sc = [h2.DevBuf.from_numpy(synth.uniform_fr(n, 5 + i)) for i in range(4)]
out = h2.DevBuf(96 * N)
run = lambda: [lib.h2mi_msm_bn254_g1_dev(p.g_handle, sc[i % 4].ptr, n, out.ptr + 96 * i, None) for i in range(N)]
for _ in range(3):
    run(); lib.h2mi_sync()
t0 = time.perf_counter()
run(); lib.h2mi_sync()
print(f"k={k} {N} MSMs back to back: {(time.perf_counter() - t0) / N * 1e6:.1f} us per MSM", flush=True)
