"""Randomised shapes through the C ABI against the oracle's structured restatement: n (incl. tile
boundaries), nU, nX, S, number of levels, treatment type, tuning knobs."""
import numpy as np
import pytest

import gpslc_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes(gp, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 5, 17, 64, 127, 128, 129, 200, 255, 256, 257, 300, 385]))
    nU = int(rng.integers(0, 5))
    nX = int(rng.integers(0, 7))
    S = int(rng.integers(1, 6))
    L = int(rng.choice([1, 2, 3, 5, 9, 17]))
    binary = bool(rng.integers(0, 2))
    X = rng.standard_normal((n, nX)) if nX else None
    T = (rng.random(n) < 0.5).astype(float) if binary else rng.standard_normal(n)
    Y = rng.standard_normal(n)
    ig = lambda size: np.maximum(4.0 / rng.gamma(4.0, 1.0, size=size), 0.3)   # noqa: E731
    U = rng.standard_normal((n, nU, S)) if nU else None
    uyLS = ig((nU, S)) if nU else None
    xyLS = ig((nX, S)) if nX else None
    tyLS, yNoise, yScale = ig(S), ig(S), ig(S)
    doTs = rng.uniform(-1.5, 1.5, L)
    if binary:
        doTs[0] = 1.0
    g = gp.GPSLCObject(X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale)
    g.ctx().set_tuning(int(rng.choice([0, 1, 2, 3])), int(rng.choice([0, 1, 2, 5])), int(rng.choice([0, 1, 2])))
    ms, vs, mi = gp.predict(g, doTs, want_mean_ite=True)
    lp = gp.yLogpdf(g)
    for s in range(S):
        p = orc.PosteriorSample(None if nU == 0 else uyLS[:, s], None if nX == 0 else xyLS[:, s], float(tyLS[s]),
                                float(yNoise[s]), float(yScale[s]), None if nU == 0 else U[:, :, s])
        rm, rv, logdet, quad = orc.structured_sate(p, X, T, Y, doTs)
        assert np.all(np.abs(ms[s] - rm) <= 1e-9 * np.abs(rm) + 1e-13), (n, nU, nX, S, L, binary)
        assert np.all(np.abs(vs[s] - rv) <= 1e-8 * np.abs(rv) + 1e-12 * p.yScale)
        assert abs(lp[s] - (-0.5 * (n * np.log(2 * np.pi) + logdet + quad))) <= 1e-10 * abs(lp[s]) + 1e-10
        for l in range(min(L, 3)):
            m, _ = orc.structured_ite(p, X, T, Y, doTs[l])
            assert np.max(np.abs(mi[:, s, l] - m)) <= 1e-9 * np.max(np.abs(m)) + 1e-13


@pytest.mark.parametrize("nU,nX,L,binary", [(2, 10, 1, False), (3, 10, 4, True), (4, 12, 1, False), (4, 16, 9, False),
                                            (5, 16, 2, True), (8, 24, 1, False), (8, 24, 6, True), (1, 4, 1, True),
                                            (0, 5, 3, False), (2, 4, 1, False), (0, 8, 1, True)])
def test_wide_feature_counts(gp, nU, nX, L, binary):
    """Every feature-count instantiation of the Gram / MeanITE kernels (exact 4, 5, 6, 8, 10, 12, 20; register
    classes 16, 20, 32; the runtime-count paths) up to the ABI's maximum nU + nX = 32."""
    rng = np.random.default_rng(7 + 31 * nU + nX + L)
    n, S = 150, 2
    X = 0.3 * rng.standard_normal((n, nX))
    T = (rng.random(n) < 0.5).astype(float) if binary else rng.standard_normal(n)
    Y = rng.standard_normal(n)
    ig = lambda size: np.maximum(4.0 / rng.gamma(4.0, 1.0, size=size), 0.5)   # noqa: E731
    U = 0.3 * rng.standard_normal((n, nU, S)) if nU else None
    uyLS = ig((nU, S)) if nU else None
    xyLS = ig((nX, S))
    tyLS, yNoise, yScale = ig(S), ig(S), ig(S)
    doTs = rng.uniform(-1.0, 1.0, L)
    if binary:
        doTs[0] = 1.0
    g = gp.GPSLCObject(X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale)
    ms, vs, mi = gp.predict(g, doTs, want_mean_ite=True)
    for s in range(S):
        p = orc.PosteriorSample(None if nU == 0 else uyLS[:, s], xyLS[:, s], float(tyLS[s]), float(yNoise[s]),
                                float(yScale[s]), None if nU == 0 else U[:, :, s])
        rm, rv, _, _ = orc.structured_sate(p, X, T, Y, doTs)
        assert np.all(np.abs(ms[s] - rm) <= 1e-9 * np.abs(rm) + 1e-13)
        assert np.all(np.abs(vs[s] - rv) <= 1e-8 * np.abs(rv) + 1e-12 * p.yScale)
        for l in range(L):
            m, _ = orc.structured_ite(p, X, T, Y, doTs[l])
            assert np.max(np.abs(mi[:, s, l] - m)) <= 1e-9 * np.max(np.abs(m)) + 1e-13


def test_results_are_bitwise_reproducible(gp):
    """The persistent workgroups take their tiles from ticket counters, so WHICH workgroup computes a tile varies
    from run to run — the arithmetic of a tile does not: two runs of the same call must agree to the last bit."""
    rng = np.random.default_rng(99)
    n, nU, nX, S, L = 700, 2, 5, 24, 3
    X = rng.standard_normal((n, nX))
    T = rng.standard_normal(n)
    Y = rng.standard_normal(n)
    ig = lambda size: np.maximum(4.0 / rng.gamma(4.0, 1.0, size=size), 0.4)   # noqa: E731
    g = gp.GPSLCObject(X, T, Y, rng.standard_normal((n, nU, S)), ig((nU, S)), ig((nX, S)), ig(S), ig(S), ig(S))
    doTs = np.array([-0.5, 0.1, 0.8])
    a = gp.predict(g, doTs, want_mean_ite=True)
    for _ in range(3):
        b = gp.predict(g, doTs, want_mean_ite=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    lp1, lp2 = gp.yLogpdf(g), gp.yLogpdf(g)
    assert np.array_equal(lp1, lp2)


@pytest.mark.parametrize("L", [15, 16, 31, 32, 33, 64, 126, 127, 128])
@pytest.mark.parametrize("n,binary", [(385, False), (300, True)])
def test_level_count_boundaries(gp, L, n, binary):
    """The number of right-hand sides (L + 1) switches code paths at 16 / 17 (one or two 16-row blocks of augmented rows
    riding with the diagonal items), 32 / 33 (augmented tiles as ordinary short items; only the live 32 rows written below
    that), and 127 / 128 (one augmented tile row: the epilogue sums z.w / w.w from the rows of R and the augmented diagonal
    tile is never updated — two tile rows: the Schur block as before).  Several tiles per side so that the in-panel chain
    (folded diagonal tile, strip kernel) runs.  Every level's SATE, MeanITE at the first and the last level."""
    rng = np.random.default_rng(5000 + L + n)
    nU, nX, S = 2, 3, 2
    X = rng.standard_normal((n, nX))
    T = (rng.random(n) < 0.5).astype(float) if binary else rng.standard_normal(n)
    Y = np.sin(T) + 0.4 * X[:, 0] + 0.3 * rng.standard_normal(n)
    ig = lambda size: np.maximum(4.0 / rng.gamma(4.0, 1.0, size=size), 0.3)   # noqa: E731
    U = rng.standard_normal((n, nU, S))
    uyLS, xyLS = ig((nU, S)), ig((nX, S))
    tyLS, yNoise, yScale = ig(S), ig(S), ig(S)
    doTs = np.linspace(-1.5, 1.5, L)
    if binary:
        doTs[0], doTs[-1] = 0.0, 1.0
    g = gp.GPSLCObject(X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale)
    ms, vs, mi = gp.predict(g, doTs, want_mean_ite=True)
    lp = gp.yLogpdf(g)
    for s in range(S):
        p = orc.PosteriorSample(uyLS[:, s], xyLS[:, s], float(tyLS[s]), float(yNoise[s]), float(yScale[s]), U[:, :, s])
        rm, rv, logdet, quad = orc.structured_sate(p, X, T, Y, doTs)
        assert np.all(np.abs(ms[s] - rm) <= 1e-9 * np.abs(rm) + 1e-13), (L, n, s)
        assert np.all(np.abs(vs[s] - rv) <= 1e-8 * np.abs(rv) + 1e-12 * p.yScale), (L, n, s)
        assert abs(lp[s] - (-0.5 * (n * np.log(2 * np.pi) + logdet + quad))) <= 1e-10 * abs(lp[s]) + 1e-10
        for l in (0, L - 1):
            m, _ = orc.structured_ite(p, X, T, Y, doTs[l])
            assert np.max(np.abs(mi[:, s, l] - m)) <= 1e-9 * np.max(np.abs(m)) + 1e-13, (L, n, s, l)
