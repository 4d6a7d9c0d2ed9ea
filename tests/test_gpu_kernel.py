"""HIP rbfKernelLog / processCov against the reference's known answers and the oracle."""
import numpy as np
import pytest

import gpslc_oracle as orc

pytestmark = pytest.mark.gpu

MAGIC = -np.array([[0, 8, 32], [8, 0, 8], [32, 8, 0]], dtype=float)


def test_rbf_magic_matrix(gp):  # test/kernel.jl:56-67
    X = np.array([[1, 2], [3, 4], [5, 6]])
    assert np.array_equal(gp.rbfKernelLog(X, X, 1), MAGIC)
    assert np.array_equal(gp.rbfKernelLog([[1, 2], [3, 4], [5, 6]], [[1, 2], [3, 4], [5, 6]], 1), MAGIC)


def test_rbf_ones(gp):  # test/kernel.jl:50-55
    X = np.ones((10, 5))
    assert np.array_equal(gp.rbfKernelLog(X, X, 0.1), np.zeros((10, 10)))


def test_rbf_bool_and_vector_inputs(gp):  # test/kernel.jl:2-48 (Bool / vector inputs promote)
    T = np.array([True, False, True])
    out = gp.rbfKernelLog(T, T, 0.5)
    assert np.array_equal(out, orc.rbf_kernel_log(T, T, 0.5))
    assert out[0, 1] == -4.0 and out[0, 2] == 0.0


@pytest.mark.parametrize("n,d,vec", [(7, 1, False), (33, 5, True), (300, 8, True), (130, 3, False)])
def test_rbf_vs_oracle(gp, n, d, vec):
    rng = np.random.default_rng(n)
    A, B = rng.standard_normal((n, d)), rng.standard_normal((n, d))
    ls = rng.uniform(0.3, 2.0, d) if vec else 0.7
    ref = orc.rbf_kernel_log(A, B, ls)
    out = gp.rbfKernelLog(A, B, ls)
    # same operations in the same order -> <= 2 ulp (fma contraction of the accumulate)
    assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-15


def test_process_cov_exact(gp):  # test/kernel.jl:69-90
    assert np.array_equal(gp.processCov(np.zeros((1, 1)), 2.0), np.ones((1, 1)) * 2.0)
    assert np.array_equal(gp.processCov(np.zeros((1, 1)), 0.0, 1e-5), np.zeros((1, 1)) + 1e-5)
    assert np.array_equal(gp.processCov(np.zeros((1, 1)), 2.0, 1e-5), np.ones((1, 1)) * 2.0 + 1e-5)


def test_process_cov_vs_oracle(gp):
    rng = np.random.default_rng(3)
    lc = -rng.uniform(0, 30, (50, 50))
    out = gp.processCov(lc, 1.7, 0.3)
    ref = orc.process_cov(lc, 1.7, 0.3)
    assert np.allclose(out, ref, rtol=4e-16, atol=0)   # exp differs by <= 1 ulp between libm and ocml
