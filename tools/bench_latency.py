#!/usr/bin/env python3
"""Latency of one model-node score at the reference's own problem sizes (what an MH / slice step waits for)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causalgpslc_jl_amd as gp   # noqa: E402

for n, F in ((150, 2), (272, 8), (1000, 8)):
    rng = np.random.default_rng(n)
    Fm = rng.standard_normal((n, F))
    ls = 1.0 + rng.random(F)
    y = rng.standard_normal(n)
    ctx = gp.Context(n, 0, 0)
    ctx.set_data(None, np.zeros(n), np.zeros(n))
    gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n} F={F}: gpLogpdf {dt * 1e6:.0f} us per call")
