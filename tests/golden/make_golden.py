"""Generates tests/golden/gpslc_golden.npz from the CPU restatement (oracle/gpslc_oracle.py).

The reference is Julia and cannot be executed in this pipeline (no Julia runtime), so these vectors
come from the literal restatement, cross-checked against the structured restatement and an
80-bit evaluation (tests/test_oracle_crosscheck.py).  They freeze the oracle's outputs so that a later
change to the oracle (or to NumPy/LAPACK) that moves the numbers is noticed.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))

import cases  # noqa: E402

out = {}
for i, (n, shape, bt) in enumerate(cases.GOLDEN_GRID):
    c = cases.make_case(n, shape, bt, seed=i)
    e = cases.oracle_expected(c)
    key = cases.golden_name(n, shape, bt)
    for k in ("X", "T", "Y", "U", "uyLS", "xyLS", "tyLS", "yNoise", "yScale", "doTs"):
        if c[k] is not None:
            out[f"{key}/in/{k}"] = c[k]
    for k, v in e.items():
        if k == "covITE" and n > 24:
            # keep the file small: store the diagonal and the full sum only for n = 150
            out[f"{key}/out/covITE_diag"] = np.einsum("slii->sli", v)
            continue
        out[f"{key}/out/{k}"] = v
np.savez_compressed(os.path.join(HERE, "gpslc_golden.npz"), **out)
print("wrote", len(out), "arrays")
