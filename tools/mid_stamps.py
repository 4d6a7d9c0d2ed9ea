#!/usr/bin/env python3
"""Per-wave phase clocks of the mid-size node-score kernel (measurement build only).
usage: GPSLC_SMALL_STAMPS=1 python tools/mid_stamps.py [n] [F]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPSLC_SMALL_STAMPS", "1")
import causalgpslc_jl_amd as gp   # noqa: E402
gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 272
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(n)
Fm = rng.standard_normal((n, F))
ls = 1.0 + rng.random(F)
y = rng.standard_normal(n)
ctx = gp.Context(n, 0, 0)
print(gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx))
