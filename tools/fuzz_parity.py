#!/usr/bin/env python3
"""Randomised parity sweep against the C oracle and the oracle verifiers — more cases than the test-suite can afford on every run (the
bounded run under `pytest -m gpu` is tests/test_gpu_fuzz.py; the cases themselves are tests/fuzz_cases.py).  Time-boxed; every case is
seeded and a failure prints its seed.  Usage: fuzz_parity.py [SECONDS] [SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

import _load_pkg  # noqa: E402

h2 = _load_pkg.load()
h2.init(0)
from fuzz_cases import Fuzzer  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
t0 = time.time()
f = Fuzzer(h2, seed0)
counts = f.run(budget, progress=lambda t, c: print(f"[{t:5.0f} s] {c}", flush=True))
print(f"fuzz_parity seed {seed0}: {counts} in {time.time() - t0:.0f} s — every case equal to the oracle", flush=True)
