#!/usr/bin/env python3
"""The halo2-lib-shaped cases of tests/fuzz_cases.py alone (random DEGREE, LOOKUP_BITS, 1 .. 12 range checks in one context: one to
dozens of columns; refusals compared with the oracle's).  Usage: fuzz_flex.py [CASES] [SEED]"""
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

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
f = Fuzzer(h2, seed0)
t0 = last = time.time()
for i in range(cases):
    f.fuzz_flex()
    if time.time() - last > 45:
        last = time.time()
        print(f"[{last - t0:5.0f} s] {i + 1} drawn {f.counts}", flush=True)
print(f"fuzz_flex seed {seed0}: {cases} drawn, {f.counts['flex']} proofs verified by the oracle ({f.counts['flex_wide']} over several columns), "
      f"{f.counts['flex_refused']} refused by both keygens, in {time.time() - t0:.0f} s", flush=True)
