#!/usr/bin/env python3
"""Latency of model-node scores at the reference's own problem sizes (what an MH / slice step waits for), and the
wall time of the NEEC gpslc() chain (BASELINE config 0)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402
if "--diag-lib" in sys.argv:      # measurement build (phase stamps of the small kernel with GPSLC_SMALL_STAMPS=<count>)
    gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")

for n, F in ((150, 2), (150, 8), (160, 3), (272, 8), (400, 8), (640, 8), (641, 8), (1000, 8)):
    rng = np.random.default_rng(n)
    Fm = rng.standard_normal((n, F))
    ls = 1.0 + rng.random(F)
    y = rng.standard_normal(n)
    ctx = gp.Context(n, 0, 0)
    gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.gpLogpdf(Fm, ls, 1.3, 0.4, y, ctx=ctx)
    dt = (time.perf_counter() - t0) / reps
    nodes = [(Fm, ls, 1.3, 0.4, y)] * 3
    gp.nodesLogpdf(nodes, ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.nodesLogpdf(nodes, ctx)
    dt3 = (time.perf_counter() - t0) / reps
    # the C entry point alone (arrays marshalled once): what a Julia ccall would see
    import ctypes as C
    sc, no, out = np.array([1.3]), np.array([0.4]), np.empty(1)
    Ff, lsf = np.asfortranarray(Fm), np.ascontiguousarray(ls)
    args = (ctx.h, 1, F, Ff.ctypes.data_as(C.c_void_p), 1, lsf.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
            no.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p))
    fn = ctx.lib.gpslc_gp_logpdf
    fn(*args)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(*args)
    dtc = (time.perf_counter() - t0) / reps
    print(f"n={n} F={F}: gpLogpdf {dt * 1e6:.0f} us per score ({dtc * 1e6:.0f} us for the C call alone); "
          f"fused 3-node call {dt3 * 1e6:.0f} us", flush=True)

neec = os.path.join(ROOT, "tests", "golden", "neec", "NEEC_sampled.csv")
gp.gpslc(neec, seed=1)
t0 = time.perf_counter()
g = gp.gpslc(neec, seed=1234)
print(f"gpslc(NEEC_sampled.csv), default hyper-parameters: {time.perf_counter() - t0:.2f} s", flush=True)
ihdp = os.path.join(ROOT, "tests", "golden", "neec", "IHDP_sampled.csv")
t0 = time.perf_counter()
g = gp.gpslc(ihdp, seed=1234)
print(f"gpslc(IHDP_sampled.csv: n = 272, 6 covariates, binary treatment), default hyper-parameters: {time.perf_counter() - t0:.2f} s",
      flush=True)
