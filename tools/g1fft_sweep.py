"""Group-valued FFT over G1 (h2mi_fft_bn254_g1_dev): wall clock of ParamsKZG::setup's Lagrange-basis step (inverse transform of the
monomial SRS + n^-1) at 2^k points, checked against the secret route's g_lagrange.  Usage: g1fft_sweep.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import field as F
from halo2_scaffold_amd.device import DevBuf
lib = h2.lib
for k in [int(x) for x in sys.argv[1:]] or [12, 16, 18, 20]:
    n = 1 << k
    ref = h2.ParamsKZG.setup(k, 0x5EC2E7 + 0x48324D49)
    out = DevBuf(n * 64)
    w_inv, n_inv = F.fr_to_mont_limbs(F.fr_inv(F.omega_for(k))), F.fr_to_mont_limbs(F.fr_inv(n))
    ts = []
    for _ in range(3):
        lib.h2mi_sync()
        t0 = time.perf_counter()
        assert lib.h2mi_fft_bn254_g1_dev(ref._g_dev.ptr, out.ptr, k, w_inv.ctypes.data, n_inv.ctypes.data, None) == 0
        ts.append(time.perf_counter() - t0)
    ok = np.array_equal(out.to_numpy(shape=(n, 8)), ref.get_g_lagrange())
    muls = (n // 2) * k + n
    print(f"k={k}: group iFFT + n^-1 of 2^{k} points {min(ts)*1e3:9.2f} ms  ({muls} scalar multiplications, {muls / min(ts) / 1e6:.2f} M/s)  equals g_lagrange: {ok}", flush=True)
    out.free(); ref.release()
