"""BASELINE config 0 / the reference's only end-to-end check (test/driver.jl:45-52): gpslc() on the NEEC
sample with default hyper-parameters, sampleITE(g, 0.6), summarizeEstimates, and at least 50 % of the
per-individual means inside the stored 90 % intervals of test/test_results/NEEC_sampled_0.6.csv."""
import csv
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NEEC = os.path.join(HERE, "golden", "neec", "NEEC_sampled.csv")
EXPECTED = os.path.join(HERE, "golden", "neec", "NEEC_sampled_0.6_expected.csv")


def count_close_enough(expected_rows, mean):   # test/test_utils.jl:3-12
    ok = [float(r["LowerBound"]) <= m <= float(r["UpperBound"]) for r, m in zip(expected_rows, mean)]
    return sum(ok) / len(ok)


def test_prepare_data_neec(gp):
    SigmaU, obj, X, T, Y = gp.prepareData(NEEC)
    assert X is None and len(T) == 150 and SigmaU.shape == (150, 150)
    assert obj == sorted(obj) and len(set(obj)) == 6                     # 6 objects x 25, sorted by obj
    assert np.array_equal(SigmaU[:25, :25], np.ones((25, 25)) + 1e-13 * np.eye(25))
    assert SigmaU[0, 25] == 0.0


def test_summarize_estimates_neec_using_gpslc(gp):
    with open(EXPECTED, newline="") as f:
        expected = list(csv.DictReader(f))
    g = gp.gpslc(NEEC, seed=1234)
    assert gp.getNumPosteriorSamples(g) == 15 and gp.getN(g) == 150      # 24 - 10 + 1
    ITEsamples = gp.sampleITE(g, 0.6, seed=7)
    assert ITEsamples.shape == (150, 150)                                # n x (15 posterior samples x 10 draws)
    actual = gp.summarizeEstimates(ITEsamples)
    frac = count_close_enough(expected, actual["Mean"])
    assert frac >= 0.50, frac
    assert np.all(actual["LowerBound"] <= actual["Mean"]) and np.all(actual["Mean"] <= actual["UpperBound"])


def test_chain_moves_and_scores_are_consistent(gp):
    """Smoke test in the spirit of test/inference.jl:31-87: latent addresses change between iterations, and
    the cached node scores equal a from-scratch evaluation."""
    from causalgpslc_jl_amd import inference as inf
    SigmaU, obj, X, T, Y = gp.prepareData(NEEC, 1e-6)
    pp = gp.getPriorParameters()
    pp["SigmaU"] = SigmaU
    rng = np.random.Generator(np.random.Philox(5))
    ch = inf._NoCovRealTChain(pp, SigmaU, T, Y, 2, rng)
    before = ch.snapshot()
    acc = 0
    for _ in range(3):
        for name in ("uNoise", "tNoise", "yNoise", "tyLS", "tScale", "yScale"):
            acc += ch.mh(name)
        for k in range(2):
            acc += ch.mh("utLS", k) + ch.mh("uyLS", k)
            ch.elliptical_slice(k)
    after = ch.snapshot()
    assert acc > 0 and any(not np.array_equal(a, b) for a, b in zip(before["U"], after["U"]))
    assert np.isclose(ch.s_t, ch.score_t(), rtol=1e-12) and np.isclose(ch.s_y, ch.score_y(), rtol=1e-12)
    assert np.isclose(ch.s_u, ch.score_u(), rtol=1e-9)
    # the model-side interleave of U for nU = 2: column-major reshape of the transposed stack
    Um = gp.toMatrixModel([np.arange(4.0), 10 + np.arange(4.0)], 4, 2)
    assert np.array_equal(Um, np.array([[0, 2], [10, 12], [1, 3], [11, 13]], dtype=float))
