#!/usr/bin/env python3
"""Latency of the fused whole-model score beyond the single-workgroup kernels (n > 640): a 3-node call (two covariate
nodes + the :Y node, heterogeneous feature counts) against a 1-node call at N = 4096 — one batched pass of the tiled
path since round 3 (round 2: one pass per node).  Usage: python tools/bench_nodes_batched.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causalgpslc_jl_amd as gp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
U, X, T = rng.standard_normal((n, 2)), rng.standard_normal((n, 8)), rng.standard_normal(n)
Y = np.sin(T) + 0.5 * X[:, 0] + 0.3 * rng.standard_normal(n)
node_x = lambda k: (U, np.array([1.1, 1.4]), 1.0, 0.5, X[:, k])
node_y = (np.hstack([U, X, T[:, None]]), rng.uniform(0.8, 2.0, 11), 0.9, 0.5, Y)
ctx = gp.Context(n, 0, 0)


def timed(nodes, reps=10):
    gp.nodesLogpdf(nodes, ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.nodesLogpdf(nodes, ctx)
    return (time.perf_counter() - t0) / reps * 1e3


one = timed([node_y])
three = timed([node_x(0), node_x(1), node_y])
ten = timed([node_x(k % 8) for k in range(9)] + [node_y])
print(f"n = {n}: 1 node {one:.2f} ms, 3 nodes {three:.2f} ms ({three / one:.2f}x), 10 nodes {ten:.2f} ms ({ten / one:.2f}x) per call")
