"""The single-launch node score (k_small.hip, gpslc_nodes_logpdf): Gram build, Cholesky, forward solve and the
reductions of one Gaussian-process node inside one workgroup, many heterogeneous nodes per launch — the
reference's own problem sizes (NEEC n = 150; src/model_likelihood.jl:4-120, src/inference.jl:21-56).
Oracle: Gen's mvnormal score restated (orc.mvnormal_logpdf of the restated covariance)."""
import numpy as np
import pytest

import gpslc_oracle as orc

pytestmark = pytest.mark.gpu


def _ref(F, ls, scale, noise, target):
    n = len(target)
    if F is None:
        cov = scale * np.ones((n, n)) + noise * np.eye(n)
    else:
        cov = orc.process_cov(orc.rbf_kernel_log(F, F, np.asarray(ls, dtype=float)), scale, noise)
    return orc.mvnormal_logpdf(np.asarray(target, dtype=float), cov)


@pytest.mark.parametrize("n", [1, 5, 15, 16, 17, 31, 33, 100, 150, 160, 176, 177, 200, 272, 400, 513, 640, 641])
def test_heterogeneous_nodes_one_call(gp, n):
    """Nodes with different feature counts in one call; n = 176 is the last size that fits the LDS-resident kernel with
    these feature counts, 177..640 run the left-looking single-workgroup kernel (finished block columns in an L2-resident
    scratch; IHDP's n = 272), 641 takes the general tiled path node by node: same answers."""
    rng = np.random.default_rng(1000 + n)
    ctx = gp.Context(n, 0, 0)
    nodes = []
    for nF in (0, 1, 3, 8, 16):
        F = None if nF == 0 else rng.standard_normal((n, nF))
        ls = None if nF == 0 else rng.uniform(0.6, 2.0, nF)
        nodes.append((F, ls, rng.uniform(0.5, 2.0), rng.uniform(0.3, 1.5), rng.standard_normal(n)))
    out = gp.nodesLogpdf(nodes, ctx)
    ref = np.array([_ref(*q) for q in nodes])
    assert np.allclose(out, ref, rtol=1e-11, atol=1e-9), (out, ref)
    assert not ctx.last_info(len(nodes)).any()
    # the per-node entry point gives the same numbers as the fused call
    one = gp.gpLogpdf(nodes[3][0], nodes[3][1], nodes[3][2], nodes[3][3], nodes[3][4], ctx=ctx)
    assert abs(one[0] - out[3]) <= 1e-12 * abs(out[3])


def test_wide_feature_block_limits(gp):
    """nF = 32 fits the LDS-resident kernel up to n = 160; n = 170 with nF = 32 silently takes the left-looking one."""
    rng = np.random.default_rng(5)
    for n in (160, 170):
        F = rng.standard_normal((n, 32))
        ls = rng.uniform(2.0, 4.0, 32)
        y = rng.standard_normal(n)
        out = gp.nodesLogpdf([(F, ls, 1.1, 0.6, y)], gp.Context(n, 0, 0))
        assert abs(out[0] - _ref(F, ls, 1.1, 0.6, y)) <= 1e-11 * abs(out[0])


def test_y_node_with_treatment_column_small_n(gp):
    """gpslc_y_logpdf at n = 150 (the NEEC size) runs the single-launch path with T as a feature column; S parameter
    sets = S workgroups of one launch; the X and Y overrides are honoured."""
    import cases
    c = cases.make_case(150, "UX", False, S=5, seed=33)
    g = cases.gpslc_object(gp, c)
    lp = gp.yLogpdf(g)
    for s, p in enumerate(cases.samples_of(c)):
        ref = orc.y_logpdf(p.uyLS, p.xyLS, p.tyLS, p.yScale, p.yNoise, p.U, c["X"], c["T"], c["Y"])
        assert abs(lp[s] - ref) <= 1e-11 * abs(ref)
    X2 = c["X"] + 0.1
    y2 = c["Y"][::-1].copy()
    lp2 = gp.yLogpdf(g, X_override=X2, Y_override=y2)
    p = cases.samples_of(c)[2]
    ref = orc.y_logpdf(p.uyLS, p.xyLS, p.tyLS, p.yScale, p.yNoise, p.U, X2, c["T"], y2)
    assert abs(lp2[2] - ref) <= 1e-11 * abs(ref)


@pytest.mark.parametrize("j", [0, 1, 15, 16, 17, 47, 48, 100, 149])
def test_failing_pivot_small_path(gp, j):
    """Not positive definite -> LAPACK-style info (1-based pivot), across 16 x 16 block boundaries of the in-LDS
    factorisation: duplicate instances + zero noise make the Gram matrix exactly singular at the second copy."""
    n = 150
    rng = np.random.default_rng(j)
    F = 3.0 * rng.standard_normal((n, 8))     # far-apart points: the Gram matrix is close to scale * I, well conditioned
    if j > 0:
        F[j] = F[j - 1]                   # row j duplicates row j-1: pivot j is exactly scale - scale = 0
    ctx = gp.Context(n, 0, 0)
    noise = 0.0 if j > 0 else -2.0        # j = 0: negative diagonal from the start
    with pytest.raises(gp.PosDefException) as ei:
        gp.nodesLogpdf([(F, np.ones(8), 1.0, noise, rng.standard_normal(n))], ctx)
    assert ei.value.info == j + 1
    assert ctx.last_info(1)[0] == j + 1


def test_many_nodes_one_launch(gp):
    """A few hundred nodes (e.g. many proposals scored at once) = a few hundred workgroups of one launch."""
    n, cnt = 96, 300
    rng = np.random.default_rng(77)
    F = rng.standard_normal((n, 3))
    y = rng.standard_normal(n)
    nodes = [(F, rng.uniform(0.5, 2.0, 3), rng.uniform(0.5, 2.0), rng.uniform(0.3, 1.0), y) for _ in range(cnt)]
    out = gp.nodesLogpdf(nodes, gp.Context(n, 0, 0))
    for i in (0, 7, 150, 299):
        assert abs(out[i] - _ref(*nodes[i])) <= 1e-11 * abs(out[i])


def test_dense_covariance_nodes_small_path(gp):
    """gpslc_mvn_logpdf at small n: the dense matrix is cached on the device and every evaluation is one workgroup that
    scales, factorises and solves in LDS (the :U => u => :U prior nodes, uCov = SigmaU * uNoise,
    src/model_likelihood.jl:4-10); a matrix that is not positive definite is reported when it is handed over."""
    sizes = [25, 40, 35, 30, 20]
    n = sum(sizes)
    SigmaU = orc.generate_sigma_u(sizes, 1e-4, 1.0)
    rng = np.random.default_rng(11)
    Uk = np.linalg.cholesky(SigmaU * 1.3) @ rng.standard_normal((n, 4))
    ctx = gp.Context(n, 0, 0)
    assert gp.mvnLogpdf(SigmaU, Uk[:, :0].reshape(n, 0), ctx=ctx).shape == (0,)      # hand over + validate only
    sc = np.array([1.3, 0.7, 2.0, 1.0])
    out = gp.mvnLogpdf(None, Uk, covscale=sc, ctx=ctx)
    ref = [orc.u_node_logpdf(SigmaU, sc[k], Uk[:, k]) for k in range(4)]
    assert np.allclose(out, ref, rtol=1e-8, atol=1e-6)
    bad = SigmaU.copy()
    bad[70, 70] = -1.0
    with pytest.raises(gp.PosDefException) as ei:
        gp.mvnLogpdf(bad, Uk[:, :0].reshape(n, 0), ctx=ctx)
    assert ei.value.info == 71


@pytest.mark.parametrize("seed", range(12))
def test_random_sizes_and_feature_counts(gp, seed):
    """Randomised shapes across the LDS-resident (n <= 176) and left-looking (n <= 640) kernels: every block-count /
    padding / wave-assignment combination a fixed grid of sizes would miss."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 330)) if seed % 3 else int(rng.integers(330, 641))
    ctx = gp.Context(n, 0, 0)
    nodes = []
    for _ in range(int(rng.integers(1, 6))):
        nF = int(rng.integers(0, 13))
        F = None if nF == 0 else rng.standard_normal((n, nF)) * rng.uniform(0.5, 2.0)
        ls = None if nF == 0 else rng.uniform(0.5, 3.0, nF)
        nodes.append((F, ls, rng.uniform(0.3, 3.0), rng.uniform(0.2, 2.0), rng.standard_normal(n) * rng.uniform(0.1, 10.0)))
    out = gp.nodesLogpdf(nodes, ctx)
    ref = np.array([_ref(*q) for q in nodes])
    assert np.allclose(out, ref, rtol=1e-10, atol=1e-9), (n, out, ref)
    again = gp.nodesLogpdf(nodes, ctx)
    assert np.array_equal(out, again)          # deterministic: no race in the LDS / scratch hand-overs


@pytest.mark.parametrize("n", [1, 5, 16, 17, 100, 150, 176, 177, 272, 400, 640])
def test_prior_draws_are_the_cholesky_factor_times_the_normals(gp, n):
    """gpslc_nodes_draw: chol(K) z for heterogeneous nodes in one launch — Gen's mvnormal(zeros(n), cov) with the
    host's normals (the auxiliary vector of elliptical_slice, src/inference.jl:225-232) — against numpy's Cholesky of the
    oracle's covariance, across the LDS-resident and the left-looking kernel."""
    rng = np.random.default_rng(4000 + n)
    ctx = gp.Context(n, 0, 0)
    nodes, refs = [], []
    for nF in (1, 3, 8, 16):
        F = rng.standard_normal((n, nF))
        ls = rng.uniform(0.6, 2.0, nF)
        scale, noise = rng.uniform(0.5, 2.0), rng.uniform(0.3, 1.5)
        z = rng.standard_normal(n)
        nodes.append((F, ls, scale, noise, z))
        K = orc.process_cov(orc.rbf_kernel_log(F, F, ls), scale, noise)
        refs.append(np.linalg.cholesky(K) @ z)
    out = gp.nodesDraw(nodes, ctx)
    assert out.shape == (n, len(nodes))
    for i, ref in enumerate(refs):
        assert np.allclose(out[:, i], ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max()), (n, i, np.abs(out[:, i] - ref).max())
    # the scores of the same call are unaffected by the draw mode
    lp = gp.nodesLogpdf(nodes, ctx)
    assert np.allclose(lp, [_ref(*q) for q in nodes], rtol=1e-11)


@pytest.mark.parametrize("n,count", [(641, 2), (700, 3), (1100, 1)])
def test_prior_draws_beyond_the_single_workgroup_kernels(gp, n, count):
    """Round 6 (VERDICT r05 missing #4): gpslc_nodes_draw on the batched tiled path — the factorisation the node scores use
    (one persistent launch of tile tasks at these sizes) followed by L z on the predictive-draw kernel — against numpy's
    Cholesky of the oracle's covariance; heterogeneous feature counts in one call, the scores as a by-product."""
    rng = np.random.default_rng(9000 + n)
    ctx = gp.Context(n, 0, 0)
    nodes, refs = [], []
    for i in range(count):
        nF = (2, 5, 9)[i]
        F = rng.standard_normal((n, nF))
        ls = rng.uniform(0.8, 2.0, nF)
        scale, noise = rng.uniform(0.5, 2.0), rng.uniform(0.3, 1.5)
        z = rng.standard_normal(n)
        nodes.append((F, ls, scale, noise, z))
        K = orc.process_cov(orc.rbf_kernel_log(F, F, ls), scale, noise)
        refs.append(np.linalg.cholesky(K) @ z)
    out = gp.nodesDraw(nodes, ctx)
    assert out.shape == (n, count)
    for i, ref in enumerate(refs):
        assert np.allclose(out[:, i], ref, rtol=1e-9, atol=1e-10 * np.abs(ref).max()), (n, i, np.abs(out[:, i] - ref).max())
    assert np.array_equal(out, gp.nodesDraw(nodes, ctx))


@pytest.mark.parametrize("n", [150, 272, 700])
def test_u_prior_draws_from_the_cached_covariance(gp, n):
    """gpslc_mvn_draw: chol(uNoise SigmaU) z from the covariance gpslc_mvn_logpdf caches — the node kernels' draw mode up to
    n = 640, the cached tiled factor on the predictive-draw kernel beyond — against numpy on a block covariance with a
    jitter that leaves it comfortably positive definite (1e-6; with the reference's 1e-13 any factor is rounding noise in the
    null directions: src/utils.jl:17-33)."""
    rng = np.random.default_rng(50 + n)
    obj = np.repeat(np.arange((n + 24) // 25), 25)[:n]
    Sig = (obj[:, None] == obj[None, :]).astype(float) + 1e-6 * np.eye(n)
    ctx = gp.Context(n, 0, 0)
    gp.mvnLogpdf(Sig, np.zeros((n, 0)), ctx=ctx)                 # hand-over, as the chain does
    z = rng.standard_normal((n, 3))
    cs = np.array([0.7, 1.9, 4.0])
    out = gp.mvnDraw(None, z, covscale=cs, ctx=ctx)
    L = np.linalg.cholesky(Sig)
    for s in range(3):
        ref = np.sqrt(cs[s]) * (L @ z[:, s])
        # the factor of a covariance with a 1e-6 jitter moves by cond * eps ~ 1e-9 relative between two valid Cholesky routines
        assert np.allclose(out[:, s], ref, rtol=0, atol=1e-7 * np.abs(ref).max()), (n, s, np.abs(out[:, s] - ref).max())
    # what matters for the chain: the draw has the right covariance action — L L^T = Sig to rounding
    e = np.eye(n)[:, :2]
    cols = gp.mvnDraw(None, e, ctx=ctx)                          # first two columns of chol(Sig)
    assert np.allclose(cols, L[:, :2], atol=1e-9)
