#!/usr/bin/env python3
"""The reference's documented example workflow (docs/example_data/NEEC_Example.jl:7-30) end to end on one GPU:
gpslc(NEEC_sampled.csv; nOuter = 100, nU = 2, nMHInner = 3, nESInner = 5) -> predictCounterfactualEffects(g, 100;
fidelity = 100) (91 posterior samples x 101 intervention levels x 100 draws: a 101 x 150 x 9100 tensor) ->
summarizeEstimates of the per-level SATE of one object."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402

neec = os.path.join(ROOT, "tests", "golden", "neec", "NEEC_sampled.csv")
hp = gp.getHyperParameters()
hp.nOuter, hp.nU, hp.nMHInner, hp.nESInner = 100, 2, 3, 5
gp.gpslc(neec, seed=1)                                   # warm-up (library load, workspaces)
t0 = time.perf_counter()
g = gp.gpslc(neec, hyperparams=hp, seed=1234)
t1 = time.perf_counter()
print(f"gpslc(nOuter=100, nU=2, nMHInner=3, nESInner=5): {t1 - t0:.2f} s, {gp.getNumPosteriorSamples(g)} posterior samples",
      flush=True)
for rep in range(2):
    t1 = time.perf_counter()
    ite, doT = gp.predictCounterfactualEffects(g, 100, fidelity=100, seed=7)
    t2 = time.perf_counter()
    print(f"predictCounterfactualEffects(g, 100; fidelity=100): {t2 - t1:.2f} s, ite {ite.shape} "
          f"({ite.nbytes / 1e9:.2f} GB), {ite.shape[0] * gp.getNumPosteriorSamples(g) / (t2 - t1):.0f} (sample, level) units/s",
          flush=True)
idx = np.array([o == "MA" for o in g.obj])
sate = ite[:, idx, :].mean(axis=1)
t2 = time.perf_counter()
s = gp.summarizeEstimates(sate)
print(f"summarizeEstimates(101 x 9100): {time.perf_counter() - t2:.3f} s; mean SATE(doT) from {s['Mean'][0]:.3f} to {s['Mean'][-1]:.3f}")
assert np.all(np.isfinite(ite))
