#!/usr/bin/env python3
"""Unit C of SURVEY.md §8d on its own clock: HIP-event time of the predictive-draw launches (gpslc profile class 2) for
`samples x levels` units of `spp` draws at size N, as TB/s of factor stream (4 N^2 B per unit, SURVEY's algorithmic figure),
plus a SHA-256 of the draw tensor — two builds that claim bit-identical draws must print the same digest.
Usage: bench_draws.py [--diag-lib] [N samples levels spp reps]"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causalgpslc_jl_amd as gp          # noqa: E402
from causalgpslc_jl_amd import synth    # noqa: E402

if "--diag-lib" in sys.argv:      # measurement build (GPSLC_* switches live there only)
    sys.argv.remove("--diag-lib")
    gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")
n, S, L, spp, reps = (int(a) for a in (sys.argv[1:6] + ["4096", "8", "8", "10", "3"][len(sys.argv) - 1:]))
D, K = 8, 2
X, T, Y, obj = synth.make_dataset(n, D)
post = synth.make_posterior(n, D, K, S, obj, seed=4321)
doT = synth.levels(T, L)
ctx = gp.Context(n, D, K, profile=True)
ctx.set_data(X, T, Y)


def p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


U = np.asfortranarray(post["U"])
arrs = [U, np.asfortranarray(post["uyLS"]), np.asfortranarray(post["xyLS"]), post["tyLS"], post["yScale"], post["yNoise"]]
mi = np.empty(n * S * L)
dr = np.empty(L * n * S * spp)
for r in range(reps + 1):
    if r == 1:
        ctx.profile_reset()
    st = ctx.lib.gpslc_predict(ctx.h, S, *[p(a) for a in arrs], L, p(doT), 1e-10, spp, 7, None, None, None, p(mi), p(dr))
    ctx.check(st)
launches, ms, draws = ctx.profile_get(2)
units = draws / spp
tbs = units * 4.0 * n * n / (ms * 1e-3) / 1e12
print(f"N={n} {S}x{L} units spp={spp}: {launches} draw launches, {ms / launches:.4f} ms per launch, "
      f"{tbs:.3f} TB/s of factor stream ({tbs / 8.0:.3f} of 8 TB/s), sha256(draws)={hashlib.sha256(dr.tobytes()).hexdigest()[:16]}")
