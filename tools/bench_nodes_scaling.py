#!/usr/bin/env python3
"""Time of one fused nodes call against the number of nodes (host staging + PCIe reads grow, the kernel does not)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402

for n, F in ((150, 3), (272, 8)):
    rng = np.random.default_rng(n)
    Fm = np.asfortranarray(rng.standard_normal((n, F)))
    y = rng.standard_normal(n)
    ctx = gp.Context(n, 0, 0)
    for cnt in (1, 2, 4, 8, 16, 24, 32, 64):
        nodes = [(Fm, 1.0 + rng.random(F), 1.3, 0.4, y) for _ in range(cnt)]
        gp.nodesLogpdf(nodes, ctx)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            gp.nodesLogpdf(nodes, ctx)
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n} F={F}: {cnt:3d} nodes per call: {dt * 1e6:.0f} us", flush=True)
