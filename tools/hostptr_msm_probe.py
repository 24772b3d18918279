#!/usr/bin/env python3
"""Host-pointer MSM (the entry point an unmodified halo2 would bind: scalars in host memory) at 2^k, plain vs sliced over
n virtual devices on ONE GPU (H2MI_VIRTUAL_DEVICES=1): slices let the upload of slice i+1 overlap the MSM of slice i.
Usage: python tools/hostptr_msm_probe.py <ndev (0 = h2mi_init)> [k]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import _load_pkg

ndev = int(sys.argv[1]) if len(sys.argv) > 1 else 0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if ndev:
    os.environ["H2MI_VIRTUAL_DEVICES"] = "1"
h2 = _load_pkg.load()
lib = h2.lib
if ndev:
    assert lib.h2mi_init_devices(ndev) == 0
else:
    h2.init(0)
from halo2_scaffold_amd import synth  # noqa: E402

n = 1 << k
params = h2.ParamsKZG.setup(k, 0x1234567)  # registers g (sharded when ndev > 0)
sc = synth.uniform_fr(n, 5)
out = np.zeros(12, dtype=np.uint64)
for _ in range(3):
    assert lib.h2mi_msm_bn254_g1(params.g_handle, None, sc.ctypes.data, n, out.ctypes.data) == 0
ts = []
for _ in range(15):
    t0 = time.perf_counter()
    assert lib.h2mi_msm_bn254_g1(params.g_handle, None, sc.ctypes.data, n, out.ctypes.data) == 0
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"ndev={ndev} k={k}: host-pointer MSM median {1e3 * ts[len(ts) // 2]:.3f} ms  min {1e3 * ts[0]:.3f} ms")
