#!/usr/bin/env python3
"""per-kernel time of one MSM 2^k whose scalars are all zero except a few rows (a real advice column)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _load_pkg
h2 = _load_pkg.load(); h2.init(0)
from halo2_scaffold_amd import synth
lib = h2.lib
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = 1 << k
params = h2.ParamsKZG.setup(k, 0x5EC2E7)
sc = np.zeros((n, 4), dtype=np.uint64)
sc[:nz] = synth.uniform_fr(nz, 3)
d = h2.DevBuf.from_numpy(sc); out = h2.DevBuf(96)
for _ in range(3):
    lib.h2mi_msm_bn254_g1_dev(params.g_lagrange_handle, d.ptr, n, out.ptr, None); lib.h2mi_sync()
lib.h2mi_profile_reset(); lib.h2mi_profile_filter(b""); lib.h2mi_profile_enable(1)
lib.h2mi_msm_bn254_g1_dev(params.g_lagrange_handle, d.ptr, n, out.ptr, None); lib.h2mi_sync()
lib.h2mi_profile_enable(0)
need = C.c_size_t(); lib.h2mi_profile_dump(None, 0, C.byref(need))
buf = C.create_string_buffer(need.value + 16); lib.h2mi_profile_dump(buf, need.value + 16, None)
print(buf.value.decode())
ba, ra = C.c_uint64(), C.c_uint64(); lib.h2mi_msm_last_stats(params.g_lagrange_handle, C.byref(ba), C.byref(ra)); print("entries", ba.value)
