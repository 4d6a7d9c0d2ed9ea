"""Host-side mirror of the reference's interface for the hot path.

Names, argument meaning and error behaviour follow CausalGPSLC.jl so that the parity tests read
like the reference's own tests:

    rbfKernelLog, processCov                      src/kernel.jl:24-32, 53-59
    GPSLCObject (data + flattened posterior)      src/types.jl:249-258, src/utils.jl:92-124
    conditionalITE, ITEDistributions, ITEsamples,
    conditionalSATE, SATEDistributions, SATEsamples   src/estimation.jl:36-163
    sampleITE, sampleSATE, summarizeEstimates     src/driver.jl:86-89, 108-111, 129-149
    predictCounterfactualEffects                  src/prediction.jl:23-36
    yLogpdf                                       src/model_likelihood.jl:83-120 (:Y node score)

Everything numeric is done by libgpslc_hip.so through the C ABI (``_lib``); this module only
marshals arrays (NumPy, column-major) and reproduces the reference's output layouts.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GPSLCError, PosDefException

PREDICTION_COVARIANCE_NOISE = 1e-10   # src/hyperparameters.jl:92


def _f(a, shape=None):
    """float64, column-major, contiguous."""
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.astype(np.float64)
    a = np.asfortranarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise AssertionError(f"expected shape {shape}, got {a.shape}")
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """RAII wrapper of gpslc_ctx (one per GPU and data set)."""

    def __init__(self, n, nX, nU, device=0, profile=False, fp32_kernel=False):
        self.lib = _lib.load()
        self.n, self.nX, self.nU = int(n), int(nX), int(nU)
        h = C.c_void_p()
        st = self.lib.gpslc_create(C.byref(h), int(device), self.n, self.nX, self.nU,
                                   (1 if profile else 0) | (2 if fp32_kernel else 0))
        if st != 0:
            raise GPSLCError(st, {-1003: "no usable gfx950 device"}.get(st, "gpslc_create failed"))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpslc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st < 0:
            raise GPSLCError(st, self.lib.gpslc_last_error(self.h).decode())
        if st > 0:
            raise PosDefException(st)

    def set_data(self, X, T, Y):
        X = None if self.nX == 0 else _f(X).reshape(self.n, self.nX, order="F")
        T = _f(T, (self.n,))
        Y = _f(Y, (self.n,))
        self._keep = (X, T, Y)
        self.check(self.lib.gpslc_set_data(self.h, _p(X), _p(T), _p(Y)))

    def set_tuning(self, max_batch=0, panel_tiles=0, n_streams=0):
        self.check(self.lib.gpslc_set_tuning(self.h, max_batch, panel_tiles, n_streams))

    def set_task_schedule(self, min_tiles=0, max_tiles=-1, min_matrices=0, group=0):
        """Persistent factorisation launch for chunks of at least min_matrices matrices of min_tiles .. max_tiles tiles per
        side (max_tiles = 0: off; 0 / -1 / 0 / 0 = keep)."""
        self.check(self.lib.gpslc_set_task_schedule(self.h, min_tiles, max_tiles, min_matrices, group))

    def set_ensemble(self, sample_offset=0, S_total=0):
        """Placement of the next calls' samples inside a larger ensemble (Philox stream ids): gpslc_set_ensemble."""
        self.check(self.lib.gpslc_set_ensemble(self.h, int(sample_offset), int(S_total)))

    def last_info(self, S):
        out = np.zeros(S, dtype=np.int32)
        self.check(self.lib.gpslc_last_info(self.h, out.ctypes.data_as(_lib.c_int32_p), S))
        return out

    def profile_reset(self):
        self.lib.gpslc_profile_reset(self.h)

    def profile_get(self, kernel_class=0):
        """(launches, ms, algorithmic flop) of the tile-update kernel; class 0 = trailing updates (the dominant
        kernel), 1 = the fused in-panel launches."""
        n = C.c_int64()
        ms = C.c_double()
        fl = C.c_double()
        self.lib.gpslc_profile_get_class(self.h, int(kernel_class), C.byref(n), C.byref(ms), C.byref(fl))
        return n.value, ms.value, fl.value


_scratch_ctx: Optional[Context] = None


def _kernel_ctx() -> Context:
    global _scratch_ctx
    if _scratch_ctx is None:
        _scratch_ctx = Context(1, 0, 0)
    return _scratch_ctx


# ------------------------------------------------------------------------------------------
# src/kernel.jl
# ------------------------------------------------------------------------------------------

def rbfKernelLog(X1, X2, LS):
    """rbfKernelLog(X1, X2, LS) -> n x n (src/kernel.jl:24-32 matrix/vector; :34-42 vector of vectors)."""
    A = np.asarray(X1)
    B = np.asarray(X2)
    if A.dtype == object or B.dtype == object:
        raise TypeError("ragged input")
    if A.shape != B.shape:
        raise AssertionError("X1 and X2 are different sizes!")
    A = _f(A if A.ndim == 2 else A.reshape(A.shape[0], -1))
    B = _f(B if B.ndim == 2 else B.reshape(B.shape[0], -1))
    n, d = A.shape
    ls = np.atleast_1d(np.asarray(LS, dtype=np.float64))
    if ls.shape[0] not in (1, d) or ls.ndim != 1:
        raise AssertionError("vector lengthscale doesn't match individual")
    ls = np.ascontiguousarray(ls)
    out = np.empty((n, n), order="F")
    ctx = _kernel_ctx()
    ctx.check(ctx.lib.gpslc_rbf_log(ctx.h, _p(A), _p(B), n, d, _p(ls), ls.shape[0], _p(out)))
    return out


def rbfKernelLogScalar(Xi, Xiprime, LS):
    """rbfKernelLogScalar(Xi, Xiprime, LS) = -sum((Xi - Xiprime)^2 / LS^2) for one pair of individuals
    (src/kernel.jl:13-19), evaluated by the same device kernel as the matrix form (n = 1)."""
    a = np.atleast_1d(np.asarray(Xi, dtype=np.float64))
    b = np.atleast_1d(np.asarray(Xiprime, dtype=np.float64))
    ls = np.asarray(LS, dtype=np.float64)
    if not (ls.shape == () or ls.shape[0] == a.shape[0]):
        raise AssertionError("vector lengthscale doesn't match individual")
    return float(rbfKernelLog(a[None, :], b[None, :], ls)[0, 0])


def logit(prob):
    """logit(prob) = log(prob / (1 - prob)) (src/kernel.jl:46)."""
    return float(np.log(prob / (1 - prob)))


def expit(x):
    """expit(x) = exp(x) / (1 + exp(x)) (src/kernel.jl:49)."""
    return float(np.exp(x) / (1.0 + np.exp(x)))


def processCov(logCov, scale, noise=None):
    """processCov(logCov, scale[, noise]) (src/kernel.jl:53-59)."""
    lc = _f(np.atleast_2d(logCov))
    n = lc.shape[0]
    if lc.shape != (n, n):
        raise AssertionError("logCov must be square")
    out = np.empty((n, n), order="F")
    ctx = _kernel_ctx()
    ctx.check(ctx.lib.gpslc_process_cov(ctx.h, _p(lc), n, float(scale), 0.0 if noise is None else float(noise),
                                        _p(out)))
    return out


# ------------------------------------------------------------------------------------------
# GPSLCObject: data + the flattened posterior pack extractParameters would produce
# ------------------------------------------------------------------------------------------

@dataclass
class HyperParameters:
    """src/types.jl:22-30 with the defaults of src/hyperparameters.jl:85-102."""
    nU: Optional[int] = 1
    nOuter: int = 24
    nMHInner: int = 10
    nESInner: int = 5
    nBurnIn: int = 10
    stepSize: int = 1
    predictionCovarianceNoise: float = PREDICTION_COVARIANCE_NOISE


@dataclass
class GPSLCObject:
    """Data (X, T, Y) plus the posterior samples in flat form.

    ``U`` (n, nU, S), ``uyLS`` (nU, S), ``xyLS`` (nX, S), ``tyLS``/``yNoise``/``yScale`` (S,) are the
    values ``extractParameters(g, i)`` (src/utils.jl:92-124) returns for i in nBurnIn:stepSize:nOuter
    (burn-in index inclusive, src/estimation.jl:72,78), stacked along the last axis.  ``U``/``uyLS``
    are None for the models without latent confounders, ``X``/``xyLS`` None without covariates.
    """
    X: Optional[np.ndarray]
    T: np.ndarray
    Y: np.ndarray
    U: Optional[np.ndarray]
    uyLS: Optional[np.ndarray]
    xyLS: Optional[np.ndarray]
    tyLS: np.ndarray
    yNoise: np.ndarray
    yScale: np.ndarray
    hyperparams: HyperParameters = field(default_factory=HyperParameters)
    device: int = 0
    fp32_kernel: bool = False   # GPSLC_FLAG_FP32_KERNEL: RBF evaluation in fp32, factorisation in fp64
    _ctx: Optional[Context] = field(default=None, repr=False)

    def __post_init__(self):
        self.T = _f(self.T).reshape(-1)
        self.Y = _f(self.Y).reshape(-1)
        n = self.Y.shape[0]
        if self.T.shape[0] != n:
            raise AssertionError("size(T, 1) != n")
        if self.X is not None:
            self.X = _f(self.X).reshape(n, -1, order="F")
        self.tyLS = np.ascontiguousarray(np.atleast_1d(self.tyLS), dtype=np.float64)
        S = self.tyLS.shape[0]
        self.yNoise = np.ascontiguousarray(np.atleast_1d(self.yNoise), dtype=np.float64)
        self.yScale = np.ascontiguousarray(np.atleast_1d(self.yScale), dtype=np.float64)
        if self.U is not None:
            U = np.asarray(self.U, dtype=np.float64)
            if U.ndim == 2:
                U = U[:, :, None] if S == 1 else U[:, None, :]
            self.U = np.asfortranarray(U)
            if self.U.shape[0] != n or self.U.shape[2] != S:
                raise AssertionError("size(U, 1) != n")
            self.uyLS = np.asfortranarray(np.asarray(self.uyLS, dtype=np.float64).reshape(self.U.shape[1], S, order="F"))
        if self.X is not None:
            self.xyLS = np.asfortranarray(np.asarray(self.xyLS, dtype=np.float64).reshape(self.X.shape[1], S, order="F"))

    # ---- size getters, src/utils.jl:130-161
    def getN(self):
        return self.Y.shape[0]

    def getNX(self):
        return 0 if self.X is None else self.X.shape[1]

    def getNU(self):
        return 0 if self.U is None else self.U.shape[1]

    def getNumPosteriorSamples(self):
        return self.tyLS.shape[0]

    def ctx(self) -> Context:
        if self._ctx is None:
            c = Context(self.getN(), self.getNX(), self.getNU(), device=self.device, fp32_kernel=self.fp32_kernel)
            c.set_data(self.X, self.T, self.Y)
            self._ctx = c
        return self._ctx

    def _params(self):
        return (_p(self.U), _p(self.uyLS), _p(self.xyLS), _p(self.tyLS), _p(self.yScale), _p(self.yNoise))

    def ctxs(self, devices: Sequence[int]):
        """One context per entry of ``devices`` (repeats allowed: distinct contexts on one GPU), each holding the data —
        what ``gpslc_predict_multi`` shards the posterior samples over.  Cached per device list."""
        key = tuple(int(d) for d in devices)
        if not key:
            raise AssertionError("devices must name at least one GPU")
        cache = self.__dict__.setdefault("_multi", {})
        if key not in cache:
            cs = []
            for d in key:
                c = Context(self.getN(), self.getNX(), self.getNU(), device=d, fp32_kernel=self.fp32_kernel)
                c.set_data(self.X, self.T, self.Y)
                cs.append(c)
            cache[key] = cs
        return cache[key]


def getN(g):
    return g.getN()


def getNX(g):
    return g.getNX()


def getNU(g):
    return g.getNU()


def getNumPosteriorSamples(g):
    return g.getNumPosteriorSamples()


def _single(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y) -> GPSLCObject:
    if U is not None:
        U = np.asarray(U, dtype=np.float64)
        U = U.reshape(U.shape[0], -1)[:, :, None]
        uyLS = np.atleast_1d(np.asarray(uyLS, dtype=np.float64))[:, None]
    if X is not None:
        X = np.asarray(X, dtype=np.float64)
        X = X.reshape(X.shape[0], -1)
        xyLS = np.atleast_1d(np.asarray(xyLS, dtype=np.float64))[:, None]
    return GPSLCObject(X, T, Y, U, uyLS, xyLS, [tyLS], [yNoise], [yScale])


# ------------------------------------------------------------------------------------------
# src/estimation.jl
# ------------------------------------------------------------------------------------------

def likelihoodDistribution(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT):
    """likelihoodDistribution(...) -> (Y, CovWW, CovWWs, CovWWp, CovC11, CovC12, CovC21, CovC22)
    (src/likelihood.jl:8-174; the method is chosen by which of U / X is None, like the reference's dispatch)."""
    g = _single(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y)
    n = g.getN()
    ctx = g.ctx()
    outs = [np.empty((n, n), order="F") for _ in range(7)]
    st = ctx.lib.gpslc_likelihood_distribution(ctx.h, _p(g.U), _p(g.uyLS), _p(g.xyLS), float(tyLS), float(yScale),
                                               float(yNoise), float(doT), *[_p(o) for o in outs])
    ctx.check(st)
    return (g.Y.copy(), *outs)


def extractParameters(g: "GPSLCObject", posteriorSampleIdx: int):
    """extractParameters(g, i) -> (uyLS, xyLS, tyLS, yNoise, yScale, U) (src/utils.jl:92-124); ``i`` is
    1-based like the reference, counting the retained posterior samples."""
    i = int(posteriorSampleIdx) - 1
    if not 0 <= i < g.getNumPosteriorSamples():
        raise IndexError("posterior sample index out of range")   # Julia: BoundsError
    return (None if g.uyLS is None else g.uyLS[:, i].copy(), None if g.xyLS is None else g.xyLS[:, i].copy(),
            float(g.tyLS[i]), float(g.yNoise[i]), float(g.yScale[i]), None if g.U is None else g.U[:, :, i].copy())


def conditionalITE(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT):
    """conditionalITE(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT) -> MeanITE (n,), CovITE (n, n)
    (src/estimation.jl:36-50; no jitter, like the reference)."""
    g = _single(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y)
    M, Cv = _ite_distributions(g, doT, 0.0)
    return M[0], Cv[0]


def _ite_distributions(g: GPSLCObject, doT, pred_noise, want_cov=True):
    n, S = g.getN(), g.getNumPosteriorSamples()
    ctx = g.ctx()
    M = np.empty((S, n), order="F")
    Cv = np.empty((S, n, n), order="F") if want_cov else None
    st = ctx.lib.gpslc_ite_distributions(ctx.h, S, *g._params(), float(doT), float(pred_noise), _p(M), _p(Cv))
    ctx.check(st)
    return M, Cv


def ITEDistributions(g: GPSLCObject, doT):
    """MeanITEs (S, n), CovITEs (S, n, n) incl. + I*predictionCovarianceNoise (src/estimation.jl:66-86)."""
    return _ite_distributions(g, doT, g.hyperparams.predictionCovarianceNoise)


def conditionalSATE(MeanITE, CovITE):
    """src/estimation.jl:116-121 (pure reduction; kept on the host for API parity)."""
    n = MeanITE.shape[0]
    return float(np.sum(MeanITE) / n), float(np.sum(CovITE) / n ** 2)


def SATEDistributions(g: GPSLCObject, doT):
    """MeanSATEs (S,), VarSATEs (S,) (src/estimation.jl:127-140) — O(N^2) per sample on the GPU,
    without materialising CovITE."""
    m, v, _ = predict(g, [doT])
    return m[:, 0].copy(), v[:, 0].copy()


def SATEsamples(MeanSATEs, VarSATEs, nSamplesPerMixture, z=None, seed=0):
    """src/estimation.jl:148-163 (variance passed as sigma, :159)."""
    lib = _lib.load()
    m = np.ascontiguousarray(MeanSATEs, dtype=np.float64)
    v = np.ascontiguousarray(VarSATEs, dtype=np.float64)
    S = m.shape[0]
    out = np.empty(S * nSamplesPerMixture)
    zz = None if z is None else np.ascontiguousarray(z, dtype=np.float64)
    st = lib.gpslc_sate_samples(_p(m), _p(v), S, int(nSamplesPerMixture), int(seed), _p(zz), _p(out))
    if st != 0:
        raise GPSLCError(st, "gpslc_sate_samples")
    return out


def predict(g: GPSLCObject, doTs: Sequence[float], want_mean_ite=False, spp=0, z=None, seed=0,
            want_draws=False, devices: Optional[Sequence[int]] = None):
    """The ensemble entry point (gpslc_predict): returns MeanSATE (S, L), VarSATE (S, L) and, when
    asked, MeanITE (n, S, L) / draws (L, n, S*spp).  ``devices`` = a list of GPU indices shards the posterior samples
    over one context per entry through ``gpslc_predict_multi`` (same results, bit for bit)."""
    n, S = g.getN(), g.getNumPosteriorSamples()
    doTs = np.ascontiguousarray(np.atleast_1d(np.asarray(doTs, dtype=np.float64)))
    L = doTs.shape[0]
    ctx = g.ctx() if devices is None else g.ctxs(devices)[0]
    ms = np.empty((S, L), order="F")
    vs = np.empty((S, L), order="F")
    mi = np.empty((n, S, L), order="F") if want_mean_ite else None
    dr = np.empty((L, n, S * spp), order="F") if want_draws else None
    zz = None
    if z is not None:
        zz = _f(z)
        if zz.shape != (n, spp, S, L):
            raise AssertionError(f"z must be (n, spp, S, L) = {(n, spp, S, L)}, got {zz.shape}")
    if devices is None:
        st = ctx.lib.gpslc_predict(ctx.h, S, *g._params(), L, _p(doTs), float(g.hyperparams.predictionCovarianceNoise),
                                   int(spp), int(seed), _p(zz), _p(ms), _p(vs), _p(mi), _p(dr))
    else:
        cs = g.ctxs(devices)
        hs = (C.c_void_p * len(cs))(*[c.h for c in cs])
        st = ctx.lib.gpslc_predict_multi(len(cs), hs, S, *g._params(), L, _p(doTs),
                                         float(g.hyperparams.predictionCovarianceNoise), int(spp), int(seed), _p(zz),
                                         _p(ms), _p(vs), _p(mi), _p(dr), None)
    ctx.check(st)
    if want_draws:
        return ms, vs, mi, dr
    return ms, vs, mi


def ITEsamples(g_or_means, doT_or_covs, nSamplesPerMixture, z=None, seed=0):
    """ITEsamples: n x (S*spp) draws, column order sample-outer / draw-inner (src/estimation.jl:95-109).
    Called as ITEsamples(g, doT, spp): the factor of CovITE + jitter is computed once per sample on the GPU."""
    g, doT = g_or_means, doT_or_covs
    zz = None
    if z is not None:   # z given in the reference's (n, S*spp) column order
        n, S = g.getN(), g.getNumPosteriorSamples()
        # column j*spp + d  ->  [i, d, j] under a column-major reshape
        zz = np.asarray(z, dtype=np.float64).reshape(n, nSamplesPerMixture, S, order="F")[:, :, :, None]
    _, _, _, dr = predict(g, [doT], spp=nSamplesPerMixture, z=zz, seed=seed, want_draws=True)
    return np.asfortranarray(dr[0])


# ------------------------------------------------------------------------------------------
# src/driver.jl, src/prediction.jl
# ------------------------------------------------------------------------------------------

def sampleITE(g: GPSLCObject, doT, samplesPerPosterior=10, z=None, seed=0):
    """sampleITE(g, doT; samplesPerPosterior=10) -> n x (S*spp) (src/driver.jl:86-89)."""
    return ITEsamples(g, doT, samplesPerPosterior, z=z, seed=seed)


def sampleSATE(g: GPSLCObject, doT, samplesPerPosterior=10, z=None, seed=0):
    """sampleSATE(g, doT; samplesPerPosterior=10) -> (S*spp,) (src/driver.jl:108-111)."""
    m, v = SATEDistributions(g, doT)
    return SATEsamples(m, v, samplesPerPosterior, z=z, seed=seed)


def doTRange(minDoT, maxDoT, fidelity):
    """``minDoT:(|maxDoT-minDoT|/fidelity):maxDoT`` (src/prediction.jl:24-28)."""
    delta = abs(maxDoT - minDoT)
    step = delta / fidelity
    if step == 0:
        raise ValueError("range step cannot be zero")
    if maxDoT < minDoT:
        return np.zeros(0)
    npts = int(np.floor((maxDoT - minDoT) / step + 1e-9)) + 1
    return minDoT + step * np.arange(npts)


def predictCounterfactualEffects(g: GPSLCObject, nSamplesPerMixture, fidelity=100, minDoT=None, maxDoT=None,
                                 z=None, seed=0, devices: Optional[Sequence[int]] = None):
    """predictCounterfactualEffects(g, spp; fidelity, minDoT, maxDoT) -> (ite [L, n, S*spp], doTrange)
    (src/prediction.jl:23-36).  All levels share one factorisation of A per posterior sample; ``devices`` shards the
    posterior samples over several GPUs (``gpslc_predict_multi``)."""
    lo = float(np.min(g.T)) if minDoT is None else float(minDoT)
    hi = float(np.max(g.T)) if maxDoT is None else float(maxDoT)
    rng = doTRange(lo, hi, fidelity)
    zz = None
    if z is not None:   # (L, n, S*spp) in the reference's order
        n, S = g.getN(), g.getNumPosteriorSamples()
        L = rng.shape[0]
        zz = np.asarray(z, dtype=np.float64).reshape(L, n, nSamplesPerMixture, S, order="F")
        zz = np.transpose(zz, (1, 2, 3, 0))
    _, _, _, dr = predict(g, rng, spp=nSamplesPerMixture, z=zz, seed=seed, want_draws=True, devices=devices)
    return dr, rng


def mvnDraw(cov, z, covscale=None, ctx: Optional[Context] = None):
    """chol(covscale_s * cov) z_s per column of ``z`` (gpslc_mvn_draw): Gen's `mvnormal(zeros(n), uCov)` — the auxiliary vector
    of `elliptical_slice(trace, :U => k => :U, zeros(n), uCov)` (src/inference.jl:48-54) and the prior draw of
    generateUfromSigmaU (src/model_likelihood.jl:4-10) — with the host's normals.  ``cov=None`` re-uses the covariance cached in
    ``ctx`` (mvnLogpdf caches SigmaU once per data set)."""
    z = _f(z)
    if z.ndim == 1:
        z = z[:, None]
    n, S = z.shape
    covf = None
    if cov is not None:
        cov = np.asarray(cov, dtype=np.float64)
        covf = cov if cov.flags.f_contiguous or cov.flags.c_contiguous else np.ascontiguousarray(cov)  # symmetric
    cs = None if covscale is None else np.ascontiguousarray(np.atleast_1d(covscale), dtype=np.float64)
    ctx = ctx or Context(n, 0, 0)
    out = np.empty((n, S), order="F")
    ctx.check(ctx.lib.gpslc_mvn_draw(ctx.h, S, _p(covf), _p(cs), _p(z), _p(out)))
    return out


def summarizeEstimates(samples, credible_interval=0.90):
    """summarizeEstimates(samples; credible_interval=0.90) (src/driver.jl:129-149): Individual, Mean,
    LowerBound, UpperBound per row of the n x m sample matrix, computed on the GPU (Julia's type-7 quantile: in-LDS
    sort per individual up to 16384 samples per row, exact radix select beyond)."""
    s = _f(np.atleast_2d(samples))
    n, m = s.shape
    mean, lo, hi = np.empty(n), np.empty(n), np.empty(n)
    ctx = _kernel_ctx()
    ctx.check(ctx.lib.gpslc_summarize(ctx.h, _p(s), n, m, float(credible_interval), _p(mean), _p(lo), _p(hi)))
    return {"Individual": np.arange(1, n + 1), "Mean": mean, "LowerBound": lo, "UpperBound": hi}


# ------------------------------------------------------------------------------------------
# src/model_likelihood.jl :Y node
# ------------------------------------------------------------------------------------------

def yLogpdf(g: GPSLCObject, X_override=None, Y_override=None):
    """log N(y; 0, Ycov) for every parameter set of ``g`` (the score the :Y address contributes to a
    Gen trace; src/model_likelihood.jl:83-120).  ``y`` is ``Y_override`` when given (the value Gen hands to a
    distribution's logpdf), else the data Y the node is constrained to."""
    S = g.getNumPosteriorSamples()
    ctx = g.ctx()
    out = np.empty(S)
    Xo = None if X_override is None else _f(X_override).reshape(g.getN(), g.getNX(), order="F")
    Yo = None if Y_override is None else _f(Y_override, (g.getN(),))
    U, uy, xy, ty, ys, yn = g._params()
    st = ctx.lib.gpslc_y_logpdf(ctx.h, S, U, _p(Xo), _p(Yo), uy, xy, ty, ys, yn, _p(out))
    ctx.check(st)
    return out


def gpLogpdf(F, LS, scale, noise, target, ctx: Optional[Context] = None):
    """log N(target; 0, processCov(rbfKernelLog(F, F, LS), scale, noise)) on the GPU, for one parameter set
    (F (n, nF), LS (nF,), scalars) or S of them (F (n, nF, S) or shared (n, nF); LS (nF, S); scale, noise (S,);
    target (n, S) or shared (n,)): the :X => k => :X, :T / :logitT and :Y node scores of the reference's Gen
    models (src/model_likelihood.jl:13-120)."""
    scale = np.atleast_1d(np.asarray(scale, dtype=np.float64))
    noise = np.atleast_1d(np.asarray(noise, dtype=np.float64))
    S = scale.shape[0]
    target = _f(target)
    n = target.shape[0]
    t_shared = 1 if target.ndim == 1 else 0
    if F is None:
        nF, Fa, f_shared, ls = 0, None, 1, None
    else:
        Fa = _f(F)
        if Fa.ndim == 1:
            Fa = Fa[:, None]
        nF = Fa.shape[1]
        f_shared = 1 if Fa.ndim == 2 else 0
        ls = np.asfortranarray(np.asarray(LS, dtype=np.float64).reshape(nF, S, order="F"))
    own = ctx is None
    ctx = ctx or Context(n, 0, 0)
    if own:
        ctx.set_data(None, np.zeros(n), np.zeros(n))
    out = np.empty(S)
    st = ctx.lib.gpslc_gp_logpdf(ctx.h, S, nF, _p(Fa), f_shared, _p(ls), _p(scale), _p(noise), _p(target), t_shared,
                                 _p(out))
    ctx.check(st)
    return out


def _marshal_nodes(nodes, ctx):
    """gpslc_node array for a sequence of (F, LS, scale, noise, target); arrays shared by several nodes of the call (the
    same feature block under several parameter values, one target) are converted and addressed once."""
    cnt = len(nodes)
    arr = (_lib.Node * max(cnt, 1))()
    keep = []
    memo = {}

    def conv(a, shape=None):
        key = id(a)
        hit = memo.get(key)
        if hit is None:
            b = _f(a, shape)
            if b.ndim == 1 and shape is None:
                b = b[:, None]
            hit = memo[key] = (b, b.ctypes.data)
            keep.append(a)
        return hit

    for i, (F, LS, scale, noise, target) in enumerate(nodes):
        tg, tgp = conv(target, (ctx.n,))
        if F is None or (isinstance(F, np.ndarray) and F.size == 0) or (not isinstance(F, np.ndarray) and np.asarray(F).size == 0):
            Fp, lsp, nF = None, None, 0
        else:
            Fa, Fp = conv(F)
            nF = Fa.shape[1]
            ls = np.ascontiguousarray(LS, dtype=np.float64).reshape(nF)
            keep.append(ls)
            lsp = ls.ctypes.data
        nd = arr[i]
        nd.nF = nF
        nd.F = Fp
        nd.ls = lsp
        nd.scale = float(scale)
        nd.noise = float(noise)
        nd.target = tgp
    return arr, (keep, memo)


def nodesLogpdf(nodes, ctx: Context, fail_value=None):
    """The fused whole-model score (gpslc_nodes_logpdf): ``nodes`` is a sequence of (F, LS, scale, noise, target)
    — F (n, nF) or None, LS (nF,), scalars, target (n,) — one per Gen address to re-score (:X => k => :X, :T /
    :logitT, :Y with F = [U | X | T]; src/model_likelihood.jl:13-120).  One call, and for n <= 640 one kernel
    launch, whatever the nodes' feature counts.  Returns the log-densities (len(nodes),).
    A covariance that is not positive definite raises PosDefException (PDMats' behaviour inside Gen's mvnormal);
    with ``fail_value`` given, the failing nodes (gpslc_last_info) get that value instead and the others keep
    their scores — what a batch of independent MH proposals needs."""
    cnt = len(nodes)
    arr, _keep = _marshal_nodes(nodes, ctx)
    out = np.empty(cnt)
    st = ctx.lib.gpslc_nodes_logpdf(ctx.h, cnt, C.cast(arr, C.c_void_p), _p(out))
    if st > 0 and fail_value is not None:
        out[ctx.last_info(cnt) != 0] = fail_value
        return out
    ctx.check(st)
    return out


def nodesDraw(nodes, ctx: Context):
    """Draws from the nodes' Gaussian priors (gpslc_nodes_draw): ``nodes`` as for nodesLogpdf, with standard normals z
    in the target slot; returns chol(K) z per node as the columns of an (n, len(nodes)) array — the `mvnormal(zeros(n),
    cov)` of an elliptical slice's auxiliary vector (src/inference.jl:225-232) or of a prior draw, with the
    random numbers still drawn on the host (any n: the single-workgroup kernels up to n = 640, the batched tiled
    factorisation + the predictive-draw kernel beyond)."""
    cnt = len(nodes)
    arr, _keep = _marshal_nodes(nodes, ctx)
    out = np.empty((ctx.n, cnt), order="F")
    ctx.check(ctx.lib.gpslc_nodes_draw(ctx.h, cnt, C.cast(arr, C.c_void_p), _p(out), None))
    return out


def mvnLogpdf(cov, x, covscale=None, ctx: Optional[Context] = None):
    """log N(x_s; 0, covscale_s * cov): the :U => u => :U node scores (uCov = SigmaU * uNoise,
    src/model_likelihood.jl:4-10, src/model_prior.jl:27-30).  ``cov=None`` re-uses the factor cached in
    ``ctx`` by an earlier call (SigmaU is constant for a data set)."""
    x = _f(x)
    if x.ndim == 1:
        x = x[:, None]
    n, S = x.shape
    covf = None
    if cov is not None:
        cov = np.asarray(cov, dtype=np.float64)
        covf = cov if cov.flags.f_contiguous or cov.flags.c_contiguous else np.ascontiguousarray(cov)  # symmetric
    cs = None if covscale is None else np.ascontiguousarray(np.atleast_1d(covscale), dtype=np.float64)
    ctx = ctx or Context(n, 0, 0)
    out = np.empty(S)
    st = ctx.lib.gpslc_mvn_logpdf(ctx.h, S, _p(covf), _p(cs), _p(x), _p(out))
    ctx.check(st)
    return out
