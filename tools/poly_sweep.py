"""Timing of the opening-argument vector kernels (device-resident), with HBM roofline fractions."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import field as F, synth
lib = h2.lib
def t(fn, R=20):
    for _ in range(3): fn()
    lib.h2mi_sync(); t0 = time.perf_counter()
    for _ in range(R): fn()
    lib.h2mi_sync(); return (time.perf_counter() - t0) / R
for log_n in [int(x) for x in sys.argv[1:]] or [16, 20, 22]:
    n = 1 << log_n
    polys = [h2.DevBuf.from_numpy(synth.uniform_fr(n, 5 + i)) for i in range(4)]
    out = h2.DevBuf(n * 32); o32 = h2.DevBuf(32)
    x = F.fr_to_mont_limbs(0x1234567890ABCDEF); xi = F.fr_to_mont_limbs(F.fr_inv(0x1234567890ABCDEF))
    te = t(lambda: lib.h2mi_fr_eval_poly_dev(polys[0].ptr, n, x.ctypes.data, o32.ptr, None))
    tk = t(lambda: lib.h2mi_fr_kate_division_dev(polys[0].ptr, n, x.ctypes.data, xi.ctypes.data, out.ptr, None))
    ptrs = (C.c_void_p * 4)(*[p.ptr for p in polys]); sc = synth.uniform_fr(4, 9)
    tl = t(lambda: lib.h2mi_fr_lincomb_dev(ptrs, sc.ctypes.data, 4, n, out.ptr, None))
    print(f"log_n={log_n}: eval {te*1e6:7.1f} us ({32*n/te/8e12*100:4.1f}% of HBM roofline, 32n B) | kate_division {tk*1e6:7.1f} us ({64*n/tk/8e12*100:4.1f}%, 64n B) | "
          f"lincomb(4) {tl*1e6:7.1f} us ({160*n/tl/8e12*100:4.1f}%, 160n B)", flush=True)

# quotient numerator of StandardPlonk (17 extended vectors in, 1 out)
from halo2_scaffold_amd import plonk as gp
for k in [16, 20]:
    dom = h2.EvaluationDomain(3, k); ext = dom.extended_len()
    bufs = [h2.DevBuf.from_numpy(synth.uniform_fr(ext, 300 + i)) for i in range(17)]
    out = h2.DevBuf(ext * 32)
    fn = lambda: gp.evaluate_h(dom, bufs[0:3], bufs[3:8], bufs[8:11], bufs[11:14], bufs[14], bufs[15], bufs[16], 3, 5, 7, out)
    te = t(fn, R=10)
    algo = ext * 32 * 18
    print(f"k={k}: evaluate_h(standard_plonk) on 2^{dom.extended_k}: {te*1e6:8.1f} us ({algo/te/8e12*100:4.1f}% of HBM roofline, 18 x 32 B per point)", flush=True)
    for b in bufs + [out]: b.free()

# permutation grand product of one column (chunk of 1): 2 columns in, z out (+ 3n scratch)
for k in [16, 20]:
    n = 1 << k
    v, sg, z = (h2.DevBuf.from_numpy(synth.uniform_fr(n, 400 + i)) for i in range(3))
    fn = lambda: gp.permutation_product(k, [v], [sg], [0], 0x1234567, 0x7654321, n - 6, z)
    tp = t(fn, R=10)
    print(f"k={k}: permutation_product (1 column): {tp*1e6:8.1f} us ({96*n/tp/8e12*100:4.1f}% of HBM roofline, 96 B per row algorithmic)", flush=True)
    for b in (v, sg, z): b.free()
