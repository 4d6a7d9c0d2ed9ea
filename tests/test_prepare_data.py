"""prepareData (src/data.jl:20-70) on the reference's own CSV inputs: host logic, no GPU.

DataFrames.sort!(df, :obj) sees the column as CSV.jl typed it, so numeric object labels sort numerically
and the row order of X / T / Y / obj (hence of every per-individual output) follows that order."""
import csv
import os

import numpy as np

import causalgpslc_jl_amd as gp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neec")


def _rows(name):
    with open(os.path.join(GOLD, name), newline="") as f:
        return list(csv.DictReader(f))


def test_integer_labels_sort_numerically():
    SigmaU, obj, X, T, Y = gp.prepareData(os.path.join(GOLD, "additive_linear.csv"))
    assert gp.removeAdjacent(obj) == list(range(1, 21))          # 1, 2, ..., 20 — not 1, 10, 11, ..., 2
    assert all(isinstance(o, int) for o in obj)
    rows = _rows("additive_linear.csv")
    # stable: inside an object the file order is kept
    exp = [r for k in range(1, 21) for r in rows if int(r["obj"]) == k]
    assert np.array_equal(T, [float(r["T"]) for r in exp]) and np.array_equal(Y, [float(r["Y"]) for r in exp])
    assert np.array_equal(X[:, 0], [float(r["X1"]) for r in exp])
    sizes = [sum(1 for r in rows if int(r["obj"]) == k) for k in range(1, 21)]
    assert SigmaU.shape == (len(rows), len(rows))
    assert SigmaU[0, sizes[0] - 1] == 1.0 and SigmaU[0, sizes[0]] == 0.0


def test_float_labels_sort_numerically_and_bool_treatment():
    SigmaU, obj, X, T, Y = gp.prepareData(os.path.join(GOLD, "IHDP_sampled.csv"))
    assert obj == sorted(obj) and all(isinstance(o, float) for o in obj)
    lab = gp.removeAdjacent(obj)
    assert lab[:3] == [1.0, 2.0, 3.0] and lab == sorted(set(float(r["obj"]) for r in _rows("IHDP_sampled.csv")))
    assert T.dtype == np.bool_ and X.shape[1] == 6


def test_string_labels_sort_lexicographically():
    SigmaU, obj, X, T, Y = gp.prepareData(os.path.join(GOLD, "NEEC_sampled.csv"))
    assert obj == sorted(obj) and all(isinstance(o, str) for o in obj) and len(set(obj)) == 6


def test_dict_input_with_numeric_labels():
    SigmaU, obj, X, T, Y = gp.prepareData({"T": [0.1, 0.2, 0.3, 0.4], "Y": [1.0, 2.0, 3.0, 4.0], "obj": [10, 2, 10, 1]})
    assert obj == [1, 2, 10, 10] and np.array_equal(Y, [4.0, 2.0, 1.0, 3.0])
    assert np.array_equal(SigmaU[2:, 2:], np.ones((2, 2)) + 1e-13 * np.eye(2)) and SigmaU[0, 1] == 0.0
