#!/usr/bin/env python3
"""How often the device library's sin / cos of phi = 2 pi r1 (r1 = k / 2^32, the argument random_cosine() makes) carry the host libm's bits.
    python tools/libm_agreement.py [n]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(11)
x = (2.0 * math.pi) * (rng.integers(0, 2 ** 32, n).astype(np.float64) * (1.0 / 4294967296.0))
for op, name, f in ((1, "sin", math.sin), (2, "cos", math.cos)):
    got = R.debug_libm(np.full(n, op), x, np.zeros(n))
    want = np.array([f(v) for v in x])
    d = np.abs(got.view(np.int64) - want.view(np.int64))
    print(f"{name}: differs from the host libm on {100.0 * float((d != 0).mean()):.3f} % of {n} arguments, at most {int(d.max())} ulp")
