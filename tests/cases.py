"""Seeded parity cases shared by the golden generator and the tests (test infrastructure)."""
import numpy as np

import gpslc_oracle as orc

SHAPES = {"UX": (True, True), "U": (True, False), "X": (False, True), "T": (False, False)}


def make_case(n, shape, binary_t, S=2, nU=2, nX=3, seed=0, obj_size=5):
    """A small data set + S posterior samples for one of the 4 model shapes x 2 treatment types."""
    has_u, has_x = SHAPES[shape]
    rng = np.random.Generator(np.random.Philox(1234 + seed))
    X = rng.standard_normal((n, nX)) if has_x else None
    T = (rng.random(n) < 0.5).astype(np.float64) if binary_t else rng.standard_normal(n)
    nobj = (n + obj_size - 1) // obj_size
    obj = np.repeat(np.arange(nobj), obj_size)[:n]
    Y = np.sin(T) + 0.3 * rng.standard_normal(n) + 0.5 * rng.standard_normal(nobj)[obj]
    if has_x:
        Y = Y + 0.5 * X[:, 0]

    def ig(size):
        return np.maximum(4.0 / rng.gamma(4.0, 1.0, size=size), 0.25)

    U = uyLS = xyLS = None
    if has_u:
        U = np.asfortranarray(rng.standard_normal((nobj, nU, S))[obj] + 1e-6 * rng.standard_normal((n, nU, S)))
        uyLS = np.asfortranarray(ig((nU, S)))
    if has_x:
        xyLS = np.asfortranarray(ig((nX, S)))
    tyLS, yNoise, yScale = ig(S), ig(S), ig(S)
    if binary_t:
        doTs = np.array([0.0, 1.0])
    else:
        doTs = np.array([float(np.quantile(T, 0.3)) if n > 1 else 0.4, 0.6])
    return dict(n=n, shape=shape, binary_t=binary_t, S=S, X=X, T=T, Y=Y, U=U, uyLS=uyLS, xyLS=xyLS,
                tyLS=tyLS, yNoise=yNoise, yScale=yScale, doTs=doTs)


def samples_of(case):
    out = []
    for s in range(case["S"]):
        out.append(orc.PosteriorSample(
            None if case["uyLS"] is None else case["uyLS"][:, s],
            None if case["xyLS"] is None else case["xyLS"][:, s],
            float(case["tyLS"][s]), float(case["yNoise"][s]), float(case["yScale"][s]),
            None if case["U"] is None else case["U"][:, :, s]))
    return out


def gpslc_object(gp, case, **kw):
    return gp.GPSLCObject(case["X"], case["T"], case["Y"], case["U"], case["uyLS"], case["xyLS"],
                          case["tyLS"], case["yNoise"], case["yScale"], **kw)


def oracle_expected(case, pred_noise=orc.PREDICTION_COVARIANCE_NOISE):
    """Literal-restatement outputs for every (sample, level)."""
    smp = samples_of(case)
    S, L, n = case["S"], len(case["doTs"]), case["n"]
    meanITE = np.zeros((n, S, L))
    covITE = np.zeros((S, L, n, n))
    mS = np.zeros((S, L))
    vS = np.zeros((S, L))
    for l, doT in enumerate(case["doTs"]):
        M, Cv = orc.ite_distributions(smp, case["X"], case["T"], case["Y"], doT, pred_noise)
        for s in range(S):
            meanITE[:, s, l] = M[s]
            covITE[s, l] = Cv[s]
            mS[s, l], vS[s, l] = orc.conditional_sate(M[s], Cv[s])
    logpdf = np.array([orc.y_logpdf(p.uyLS, p.xyLS, p.tyLS, p.yScale, p.yNoise, p.U, case["X"], case["T"],
                                    case["Y"]) for p in smp])
    return dict(meanITE=meanITE, covITE=covITE, meanSATE=mS, varSATE=vS, logpdf=logpdf)


GOLDEN_GRID = [(n, shape, bt) for n in (1, 3, 24, 150) for shape in SHAPES for bt in (False, True)]


def golden_name(n, shape, bt):
    return f"n{n}_{shape}_{'bin' if bt else 'real'}"


def draw_bounds(lam_min, lam_max, znorm, refnorm=0.0):
    """Bounds on ||draw_gpu - (M + chol(C) z)|| for C = CovITE + jitter with the given extreme eigenvalues.

    `bound`: SURVEY §8d's 1e-8 ||L_c|| ||z|| where the conditioning permits it, i.e. max(1e-8, 1e-15 cond(C)) — a Cholesky
    factor moves by cond(C) x the 1e-15 relative rounding of forming C.  `tight` (None when cond >= 1e8): what fp64 actually
    delivers on a reasonably conditioned matrix, 1e-9 ||L_c|| ||z|| — observed errors sit four orders of magnitude below
    even that (7e-12 at N = 4096, cond 1.3e7), so a regression of the CovITE factor by a factor 1e4 no longer passes."""
    cond = lam_max / max(lam_min, 1e-300)
    lc = float(np.sqrt(lam_max))
    bound = max(1e-8, 1e-15 * cond) * lc * znorm + 1e-9 * refnorm
    tight = (1e-9 * lc * znorm + 1e-12 * refnorm) if cond < 1e8 else None
    return bound, tight, cond
