"""host-pointer MSM (h2mi_msm_bn254_g1: what the Rust shim's best_multiexp binds) wall clock per call, registered handle, sizes 2^k.
Run under the -DH2MI_AB library with and without H2MI_MSM_IGNORE_INORDER=1 to A/B the in-order form.  Usage: hostptr_ab.py K..."""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
for k in [int(x) for x in sys.argv[1:]] or [8, 16, 20]:
    n = 1 << k
    p = h2.ParamsKZG.setup(k, 0x1234567)
    sc = np.ascontiguousarray(synth.uniform_fr(n, 5))
    out = np.zeros(12, dtype=np.uint64)
    f = lambda: lib.h2mi_msm_bn254_g1(p.g_handle, None, sc.ctypes.data, n, out.ctypes.data)
    for _ in range(3): assert f() == 0
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    print(f"k={k} host-pointer MSM {statistics.median(ts) * 1e6:.1f} us  env={ {a: b for a, b in os.environ.items() if a.startswith('H2MI_MSM')} }", flush=True)
