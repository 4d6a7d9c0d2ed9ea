#!/usr/bin/env python3
"""Wall time of the gpslc() chains against the number of slice candidates scored per fused call (--slice) and the
number of consecutive MH moves of a chain scored speculatively per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402
from causalgpslc_jl_amd import inference as inf   # noqa: E402

for name in ("NEEC", "IHDP"):
    path = os.path.join(ROOT, "tests", "golden", "neec", f"{name}_sampled.csv")
    gp.gpslc(path, seed=1)
    if "--slice" in sys.argv:
        for depth in (1, 2, 4, 8, 12, 16, 24):
            inf._RealTChain.slice_depth = depth
            t0 = time.perf_counter()
            gp.gpslc(path, seed=1234)
            print(f"{name} slice depth {depth}: {time.perf_counter() - t0:.3f} s", flush=True)
        inf._RealTChain.slice_depth = 8
    for depth in (1, 2, 3, 4, 5):
        inf._RealTChain.mh_depth = depth
        t0 = time.perf_counter()
        gp.gpslc(path, seed=1234)
        print(f"{name} speculative MH depth {depth}: {time.perf_counter() - t0:.3f} s", flush=True)
    inf._RealTChain.mh_depth = 2
