"""the other Gaussian-process nodes of the reference's Gen models on the GPU —
:X => k => :X, :T / :logitT (src/model_likelihood.jl:13-80) and :U => u => :U (:4-10) — against the
oracle's restatement of Gen's mvnormal score."""
import numpy as np
import pytest

import gpslc_oracle as orc

pytestmark = pytest.mark.gpu


def _data(n, nU, nX, seed):
    rng = np.random.default_rng(seed)
    U = rng.standard_normal((n, nU))
    X = rng.standard_normal((n, nX))
    T = rng.standard_normal(n)
    return rng, U, X, T


@pytest.mark.parametrize("n", [10, 150, 300])
def test_x_nodes_given_u(gp, n):
    nU, nX = 2, 3
    rng, U, X, _ = _data(n, nU, nX, n)
    uxLS = rng.uniform(0.5, 2.0, (nX, nU))
    xScale = rng.uniform(0.5, 2.0, nX)
    xNoise = rng.uniform(0.3, 1.5, nX)
    # the nX nodes as ONE call: shared features U, one parameter set per covariate, targets X[:, k]
    out = gp.gpLogpdf(U, uxLS.T, xScale, xNoise, X)
    ref = [orc.x_node_logpdf(U, uxLS[k], xScale[k], xNoise[k], X[:, k]) for k in range(nX)]
    assert np.allclose(out, ref, rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("n", [24, 260])
def test_t_node_given_u_and_x(gp, n, binary):
    nU, nX = 2, 4
    rng, U, X, T = _data(n, nU, nX, 100 + n)
    target = rng.standard_normal(n) if binary else T      # :logitT for binary treatments, :T otherwise
    utLS, xtLS = rng.uniform(0.5, 2.0, nU), rng.uniform(0.5, 2.0, nX)
    tScale, tNoise = 1.3, 0.7
    out = gp.gpLogpdf(np.hstack([U, X]), np.concatenate([utLS, xtLS]), tScale, tNoise, target)
    ref = orc.t_node_logpdf(U, X, utLS, xtLS, tScale, tNoise, target)
    assert np.allclose(out, [ref], rtol=1e-11, atol=1e-9)
    # no-covariates and no-confounders variants (generateRealTfromU / generateRealTfromX)
    out = gp.gpLogpdf(U, utLS, tScale, tNoise, target)
    assert np.allclose(out, [orc.t_node_logpdf(U, None, utLS, None, tScale, tNoise, target)], rtol=1e-11, atol=1e-9)
    out = gp.gpLogpdf(X, xtLS, tScale, tNoise, target)
    assert np.allclose(out, [orc.t_node_logpdf(None, X, None, xtLS, tScale, tNoise, target)], rtol=1e-11, atol=1e-9)


def test_y_node_as_generic_node_matches_specialised_entry(gp):
    """The :Y node is the generic node with T as one more feature (src/model_likelihood.jl:83-91)."""
    import cases
    c = cases.make_case(200, "UX", False, S=3, seed=12)
    obj = cases.gpslc_object(gp, c)
    spec = gp.yLogpdf(obj)
    n, S = 200, 3
    F = np.concatenate([c["U"], np.repeat(c["X"][:, :, None], S, axis=2), np.repeat(c["T"][:, None, None], S, axis=2)],
                       axis=1)
    ls = np.vstack([c["uyLS"], c["xyLS"], c["tyLS"][None, :]])
    gen = gp.gpLogpdf(F, ls, c["yScale"], c["yNoise"], c["Y"])
    assert np.allclose(gen, spec, rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("eps", [1e-3, 1e-6])
def test_u_prior_nodes(gp, eps):
    """uCov = SigmaU * uNoise with SigmaU = generateSigmaU(object sizes) (src/utils.jl:17-33).  With the
    reference's default sigmaUNoise = 1e-13 the matrix is numerically singular (cond ~ 1e14) and ANY
    Cholesky returns rounding noise, so parity is checked at better-conditioned jitters."""
    sizes = [25, 40, 35, 30, 20]
    n = sum(sizes)
    SigmaU = orc.generate_sigma_u(sizes, eps, 1.0)
    rng = np.random.default_rng(4)
    nU, uNoise = 3, 1.7
    Lc = np.linalg.cholesky(SigmaU * uNoise)
    Uk = Lc @ rng.standard_normal((n, nU))          # plausible draws: object-constant + tiny within-object noise
    out = gp.mvnLogpdf(SigmaU, Uk, covscale=np.full(nU, uNoise))
    ref = [orc.u_node_logpdf(SigmaU, uNoise, Uk[:, k]) for k in range(nU)]
    tol = 1e-6 if eps < 1e-4 else 1e-9              # the quadratic form is amplified by 1/eps
    assert np.allclose(out, ref, rtol=tol, atol=1e-6)
    # cached factor: a second evaluation with cov=None (what an MCMC step over uNoise / U does)
    ctx = gp.Context(n, 0, 0)
    gp.mvnLogpdf(SigmaU, Uk[:, :0].reshape(n, 0), ctx=ctx)          # factor only
    out2 = gp.mvnLogpdf(None, Uk, covscale=np.full(nU, uNoise), ctx=ctx)
    assert np.allclose(out2, ref, rtol=tol, atol=1e-6)
    out3 = gp.mvnLogpdf(None, Uk[:, 1], covscale=[2.5], ctx=ctx)
    assert np.allclose(out3, [orc.u_node_logpdf(SigmaU, 2.5, Uk[:, 1])], rtol=tol, atol=1e-6)


def test_mvn_not_positive_definite(gp):
    cov = np.ones((6, 6))
    with pytest.raises(gp.PosDefException):
        gp.mvnLogpdf(cov, np.ones(6))


@pytest.mark.parametrize("j", [0, 1, 15, 16, 17, 100, 127, 128, 129, 200, 271, 299])
def test_failing_pivot_is_reported_like_lapack(gp, j):
    """The first non-positive pivot comes back as LAPACK's info (1-based), across 16 x 16 sub-block and
    128 x 128 tile boundaries of the diagonal-block kernel — what PDMats turns into PosDefException(info)."""
    n = 300
    rng = np.random.default_rng(j)
    d = 1.0 + rng.random(n)
    d[j] = -0.5
    cov = np.diag(d)
    # couple the leading block so that the factorisation does real work before the bad pivot
    if j > 2:
        v = 0.1 * rng.standard_normal(j)
        cov[:j, :j] += np.outer(v, v)
    with pytest.raises(gp.PosDefException) as ei:
        gp.mvnLogpdf(cov, rng.standard_normal(n))
    assert ei.value.info == j + 1


def test_near_singular_matrix_tiled_path_matches_lapack(gp):
    """A covariance that is positive definite only by a 1e-10 jitter (rank 40 + 1e-10 I, cond ~ 1e12), large enough
    (n = 700) to take the TILED factorisation: LAPACK's potrf succeeds on it, so must the library (the substitution-
    based diagonal-tile / panel kernels of k_robust.hip; products with inverted blocks lose eps * cond(L) and broke
    down here).  The log-determinant is compared, and the quadratic form for a vector in the well-conditioned range
    of the matrix."""
    n, r = 700, 40
    rng = np.random.default_rng(3)
    G = rng.standard_normal((n, r))
    cov = G @ G.T / r + 1e-10 * np.eye(n)
    Lc = np.linalg.cholesky(cov)                       # LAPACK: fine
    x = G @ rng.standard_normal(r)                     # in the range of the low-rank part: x' cov^-1 x = O(r)
    out = gp.mvnLogpdf(cov, x)
    z = np.linalg.solve(Lc, x)
    ref = -0.5 * (n * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(Lc))) + z @ z)
    assert abs(out[0] - ref) <= 1e-6 * abs(ref), (out[0], ref)


@pytest.mark.parametrize("n", [700, 2048])
def test_fused_model_score_beyond_the_single_workgroup_kernels_is_one_batched_pass(gp, n):
    """gpslc_nodes_logpdf for n > 640: the nodes of one Gen `update` (src/model.jl:11-27: two :X => k => :X nodes with
    F = U, the :T node with F = [U | X], the :Y node with F = [U | X | T]) have DIFFERENT feature counts and are scored
    by ONE batched pass of the tiled path (zero-padded feature columns of lengthscale 1).  Against the oracle's
    restatement of the Gen node scores, and bit-identical to scoring each node on its own."""
    rng = np.random.default_rng(n)
    nU, nX = 2, 3
    U = rng.standard_normal((n, nU))
    X = rng.standard_normal((n, nX))
    T = rng.standard_normal(n)
    Y = np.sin(T) + 0.5 * X[:, 0] + 0.3 * rng.standard_normal(n)
    uxLS = rng.uniform(0.8, 2.0, (nX, nU))
    xScale, xNoise = rng.uniform(0.5, 2.0, nX), rng.uniform(0.3, 1.0, nX)
    utLS, xtLS = rng.uniform(0.8, 2.0, nU), rng.uniform(0.8, 2.0, nX)
    uyLS, xyLS, tyLS = rng.uniform(0.8, 2.0, nU), rng.uniform(0.8, 2.0, nX), 1.3
    tScale, tNoise, yScale, yNoise = 1.2, 0.6, 0.9, 0.5
    nodes = [(U, uxLS[k], xScale[k], xNoise[k], X[:, k]) for k in range(2)]
    nodes.append((np.hstack([U, X]), np.concatenate([utLS, xtLS]), tScale, tNoise, T))
    nodes.append((np.hstack([U, X, T[:, None]]), np.concatenate([uyLS, xyLS, [tyLS]]), yScale, yNoise, Y))
    ctx = gp.Context(n, 0, 0)
    out = gp.nodesLogpdf(nodes, ctx)
    ref = [orc.x_node_logpdf(U, uxLS[k], xScale[k], xNoise[k], X[:, k]) for k in range(2)]
    ref.append(orc.t_node_logpdf(U, X, utLS, xtLS, tScale, tNoise, T))
    ref.append(orc.y_logpdf(uyLS, xyLS, tyLS, yScale, yNoise, U, X, T, Y))
    assert np.allclose(out, ref, rtol=1e-11, atol=1e-9), (out, ref)
    assert not ctx.last_info(len(nodes)).any()
    for i, q in enumerate(nodes):          # padding columns add exact zeros: same bits as the node's own pass
        assert gp.nodesLogpdf([q], ctx)[0] == out[i], i
    # nodes that all share ONE feature block (the :X => k => :X nodes, F = U) hand it over once (f_shared): same bits
    assert np.array_equal(gp.nodesLogpdf(nodes[:2], ctx), out[:2])
    # a failing node reports its pivot and leaves the others' scores alone
    bad = list(nodes)
    bad[1] = (U, uxLS[1], xScale[1], -5.0, X[:, 1])
    got = gp.nodesLogpdf(bad, ctx, fail_value=-np.inf)
    assert got[1] == -np.inf and got[0] == out[0] and got[2] == out[2] and got[3] == out[3]
