#!/usr/bin/env python3
"""Unit B + C of SURVEY.md §8d on one GPU: full ITE covariance + its factor + predictive draws per (sample, level)
(what sampleITE / predictCounterfactualEffects cost), host-pointer API.  Usage: bench_unit_b.py [N S L spp]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causalgpslc_jl_amd as gp          # noqa: E402
from causalgpslc_jl_amd import synth    # noqa: E402

if "--diag-lib" in sys.argv:      # measurement build (GPSLC_* switches live there only)
    sys.argv.remove("--diag-lib")
    gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")
panel = 0
if "--panel" in sys.argv:         # tile-panel width of the blocked factorisations (gpslc_set_tuning), 0 = default
    i = sys.argv.index("--panel")
    panel = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
n, S, L, spp = (int(a) for a in (sys.argv[1:5] + ["4096", "128", "1", "8"][len(sys.argv) - 1:]))
D, K = 8, 2
X, T, Y, obj = synth.make_dataset(n, D)
post = synth.make_posterior(n, D, K, S, obj, seed=1234)
g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
doTs = synth.levels(T, L)
if panel:
    g.ctx().set_tuning(0, panel, 0)
gp.predict(g, doTs[:1], spp=spp, seed=1, want_draws=True)     # warm-up (arenas sized for this spp, first touch)
units = S * L
for rep in range(2):      # the first full-size call also grows the context's workspace (tens of GB of hipMalloc): report both
    t0 = time.perf_counter()
    ms, vs, mi, dr = gp.predict(g, doTs, spp=spp, seed=7, want_draws=True)
    dt = time.perf_counter() - t0
    if rep == 0:
        print(f"(first full-size call, workspace growth included: {units / dt:.1f} units/s, {dt * 1e3:.0f} ms)")
assert np.all(np.isfinite(dr))
print(f"N={n} S={S} L={L} spp={spp}: {units / dt:.1f} (sample, level) units/s, {units * spp / dt:.0f} draws/s, "
      f"{dt * 1e3:.0f} ms; unit-B ceiling at 78.6 TFLOP/s = {78.6e12 / (7.0 / 3.0 * n ** 3):.0f}/s")
