"""Pins the CPU restatement against every known-answer test the reference holds for the hot path
(SURVEY.md §8c).  Paths are relative to the reference repository."""
import numpy as np
import pytest

import gpslc_oracle as orc

MAGIC = -np.array([[0, 8, 32], [8, 0, 8], [32, 8, 0]], dtype=float)


def test_rbf_scalar_identical_inputs():  # test/kernel.jl:2-48
    for val in (11, 11.1, True, False):
        assert orc.rbf_kernel_log_scalar([val], [val], 0.1) == 0.0
    x = np.random.default_rng(0).random(10)
    assert orc.rbf_kernel_log_scalar(x, x, 0.3) == 0.0


def test_rbf_ones_matrix():  # test/kernel.jl:50-55
    X = np.ones((10, 5))
    assert np.array_equal(orc.rbf_kernel_log(X, X, 0.1), np.zeros((10, 10)))


def test_rbf_magic_matrix():  # test/kernel.jl:56-61
    X = np.array([[1, 2], [3, 4], [5, 6]])
    assert np.array_equal(orc.rbf_kernel_log(X, X, 1), MAGIC)


def test_rbf_magic_matrix_lists():  # test/kernel.jl:62-67
    X = [[1, 2], [3, 4], [5, 6]]
    assert np.array_equal(orc.rbf_kernel_log(X, X, 1), MAGIC)


def test_rbf_lengthscale_mismatch_asserts():  # src/kernel.jl:14-16
    with pytest.raises(AssertionError):
        orc.rbf_kernel_log(np.ones((3, 2)), np.ones((3, 2)), np.ones(3))
    with pytest.raises(AssertionError):
        orc.rbf_kernel_log(np.ones((3, 2)), np.ones((4, 2)), 1.0)


def test_process_cov():  # test/kernel.jl:69-90
    assert np.array_equal(orc.process_cov(np.zeros((1, 1)), 2.0), np.ones((1, 1)) * 2.0)
    assert np.array_equal(orc.process_cov(np.zeros((1, 1)), 0.0, 1e-5), np.zeros((1, 1)) + 1e-5)
    assert np.array_equal(orc.process_cov(np.zeros((1, 1)), 2.0, 1e-5), np.ones((1, 1)) * 2.0 + 1e-5)


def test_logit_expit():  # test/kernel.jl:91-96
    assert orc.logit(0.5) == 0
    assert orc.expit(0) == 0.5


def _toy(binary):  # test/test_data.jl:35-52 getEstimationTestParams
    return dict(uyLS=np.array([1.0]), xyLS=np.array([1.0]), tyLS=1.0, yScale=1.0, yNoise=1.0,
                U=np.array([[1.0]]), X=np.ones((1, 1)), T=np.array([1.0]), Y=np.array([0.37]),
                doT=1.0)


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("shape", ["T", "X", "U", "UX"])
def test_conditional_ite_zero_when_doT_equals_T(shape, binary):  # test/estimation.jl:6-66, 69-136
    t = _toy(binary)
    has_u, has_x = shape in ("U", "UX"), shape in ("X", "UX")
    m, c = orc.conditional_ite(t["uyLS"] if has_u else None, t["xyLS"] if has_x else None, t["tyLS"],
                               t["yNoise"], t["yScale"], t["U"] if has_u else None,
                               t["X"] if has_x else None, t["T"], t["Y"], t["doT"])
    assert np.all(m == 0.0) and np.all(c == 0.0)
    ms, vs = orc.conditional_sate(m, c)
    assert ms == 0.0 and vs == 0.0


@pytest.mark.parametrize("shape", ["T", "X", "U", "UX"])
def test_ite_distributions_jitter_placement(shape):  # test/estimation.jl:139-246
    t = _toy(False)
    has_u, has_x = shape in ("U", "UX"), shape in ("X", "UX")
    p = orc.PosteriorSample(t["uyLS"] if has_u else None, t["xyLS"] if has_x else None, 0.8, 1.3, 0.9,
                            t["U"] if has_u else None)
    M, C = orc.ite_distributions([p] * 15, t["X"] if has_x else None, t["T"], t["Y"], 1.0)
    assert M.shape == (15, 1) and C.shape == (15, 1, 1)
    assert np.mean(M) == 0.0
    assert np.isclose(np.mean(C), 1e-10, rtol=1e-12, atol=0)
    ms, vs = orc.sate_distributions([p] * 15, t["X"] if has_x else None, t["T"], t["Y"], 1.0)
    assert np.mean(ms) == 0.0 and np.isclose(np.mean(vs), 1e-10, rtol=1e-12, atol=0)


def test_generate_sigma_u():  # test/utils.jl:2-16
    d = 1.1
    expected = np.array([[d, 2, 0, 0, 0], [2, d, 0, 0, 0], [0, 0, d, 2, 2], [0, 0, 2, d, 2], [0, 0, 2, 2, d]])
    assert np.array_equal(orc.generate_sigma_u([2, 3], 0.1, 2.0), expected)


def test_num_posterior_samples_default():  # test/utils.jl:50-55, src/estimation.jl:72
    assert orc.num_posterior_samples() == 24 - 10 + 1


def test_summarize_estimates_quantiles():  # test/driver.jl:54-70
    s = np.arange(101, dtype=float)[None, :]
    _, lo, hi = orc.summarize_estimates(s, 0.9)
    assert np.isclose(lo[0], 5.0) and np.isclose(hi[0], 95.0)
    _, lo, hi = orc.summarize_estimates(s, 0.8)
    assert np.isclose(lo[0], 10.0) and np.isclose(hi[0], 90.0)


def test_julia_quantile_matches_numpy_linear():
    rng = np.random.default_rng(1)
    for m in (2, 7, 150, 1001):
        v = rng.standard_normal(m)
        for p in (0.05, 0.1, 0.5, 0.95):
            assert np.isclose(orc.julia_quantile(v, p), np.quantile(v, p, method="linear"), rtol=1e-14, atol=1e-15)


def test_sate_samples_uses_variance_as_sigma():  # src/estimation.jl:159
    out = orc.sate_samples(np.array([1.0, 2.0]), np.array([0.25, 4.0]), 2, np.array([1.0, -1.0, 0.5, 2.0]))
    assert np.allclose(out, [1.25, 0.75, 4.0, 10.0])


def test_ite_samples_column_order():  # src/estimation.jl:100-107
    M = np.array([[0.0, 0.0], [10.0, 10.0]])
    C = np.stack([np.eye(2), 4 * np.eye(2)])
    z = np.ones((2, 4))
    out = orc.ite_samples(M, C, 2, z)
    assert np.allclose(out, [[1, 1, 12, 12], [1, 1, 12, 12]])


def test_do_t_range():  # src/prediction.jl:24-28 ; test/prediction.jl:1-13
    r = orc.do_t_range(0.0, 1.0, 100)
    assert len(r) == 101 and r[0] == 0.0 and np.isclose(r[-1], 1.0)
    with pytest.raises(ValueError):
        orc.do_t_range(1.0, 1.0, 10)
