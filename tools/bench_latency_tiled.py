#!/usr/bin/env python3
"""Latency of ONE node score / one node draw on the batched tiled path (641 <= n <= 1024: few matrices per call) with the
persistent factorisation launch against one launch per tile column (gpslc_set_task_schedule) — what an MH / slice step of a
chain at these sizes waits for."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402

for n, F in ((641, 8), (800, 8), (1000, 8)):
    rng = np.random.default_rng(n)
    Fm = rng.standard_normal((n, F))
    ls = 1.0 + rng.random(F)
    y = rng.standard_normal(n)
    row = [f"n={n} F={F}:"]
    for name, maxt in (("task launch (forced: min_matrices = 1)", -1), ("per column (the default for < 256 matrices)", 0)):
        ctx = gp.Context(n, 0, 0)
        ctx.set_task_schedule(0, maxt, 1, 0)
        nodes1, nodes3 = [(Fm, ls, 1.3, 0.4, y)], [(Fm, ls, 1.3, 0.4, y)] * 3
        res = []
        for f in (lambda: gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx), lambda: gp.nodesLogpdf(nodes3, ctx),
                  lambda: gp.nodesDraw(nodes1, ctx)):
            f()
            reps = 100
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            res.append((time.perf_counter() - t0) / reps * 1e6)
        row.append(f"{name}: score {res[0]:.0f} us, 3-node score {res[1]:.0f} us, draw {res[2]:.0f} us;")
        ctx.close()
    print(" ".join(row), flush=True)
