#!/usr/bin/env python3
"""Throughput of the Gen-node scores on one GPU: gpslc_gp_logpdf / gpslc_mvn_logpdf
through the host-pointer C ABI, S parameter sets per call.  Prints one JSON line per configuration."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import causalgpslc_jl_amd as gp  # noqa: E402
from causalgpslc_jl_amd import synth  # noqa: E402

for (n, D, K, S) in [(1024, 4, 1, 256), (4096, 8, 2, 128)]:
    X, T, Y, obj = synth.make_dataset(n, D)
    post = synth.make_posterior(n, D, K, S, obj)
    ctx = gp.Context(n, 0, 0)
    ctx.set_data(None, np.zeros(n), np.zeros(n))
    F = np.concatenate([post["U"], np.repeat(X[:, :, None], S, axis=2)], axis=1)     # T | U, X node
    ls = np.vstack([post["uyLS"], post["xyLS"]])
    for rep in range(2):
        t0 = time.perf_counter()
        out = gp.gpLogpdf(F, ls, post["yScale"], post["yNoise"], T, ctx=ctx)
        dt = time.perf_counter() - t0
    assert np.all(np.isfinite(out))
    sizes = [16] * (n // 16)
    SigmaU = np.eye(n)
    i = 0
    for m in sizes:
        SigmaU[i:i + m, i:i + m] = 1.0
        i += m
    SigmaU[np.diag_indices(n)] = 1 + 1e-6
    Uk = np.linalg.cholesky(SigmaU) @ np.random.default_rng(0).standard_normal((n, K * 8))
    for rep in range(2):
        t0 = time.perf_counter()
        out2 = gp.mvnLogpdf(SigmaU, Uk, covscale=np.full(K * 8, 1.3), ctx=ctx)
        dt2 = time.perf_counter() - t0
    print(json.dumps({"n": n, "nF": D + K, "gp_logpdf_sets_per_s": S / dt, "gp_logpdf_call_ms": dt * 1e3, "S": S,
                      "mvn_logpdf_call_ms": dt2 * 1e3, "mvn_vectors": K * 8}))
