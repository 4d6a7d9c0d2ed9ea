"""Host-side posterior inference for the no-covariates continuous-treatment model, scoring every
Gaussian-process node on the GPU (BASELINE config 0).

Mirrors, for ``CausalGPSLCNoCovRealT`` (src/model.jl:45-57):

    getPriorParameters / getHyperParameters      src/hyperparameters.jl:38-70, 85-102
    generateSigmaU, prepareData                  src/utils.jl:17-33, src/data.jl:20-70
    Posterior(priorparams, nothing, T, Y, ...)   src/inference.jl:62-102  (MH within Gibbs + elliptical slice)
    paramProposal                                src/proposal.jl:32-41    (moment-matched InvGamma drift)
    gpslc                                        src/driver.jl:27-33, 59-69

The Markov chain itself is sequential scalar control flow (the reference drives it through Gen's
interpreter); it stays on the host.  What each step costs — the mvnormal scores of :T, :Y and :U => k => :U —
is evaluated by libgpslc_hip.so (gpslc_gp_logpdf / gpslc_mvn_logpdf).  Only the nodes an address touches are
re-scored (the reference re-executes the whole model body per `mh`).

Not bit-comparable with the reference: Gen 0.4.4 and Julia's RNG are unavailable here, so the chain uses
NumPy's Philox generator and the textbook algorithms (Metropolis-Hastings ratio with the asymmetric InvGamma
proposal; elliptical slice sampling, Murray et al. 2010, with the likelihood = the :T and :Y scores).  The
reference's own acceptance test for this path is statistical (test/driver.jl:45-52) and is reproduced in
tests/test_gpu_neec.py.  The other seven model variants are not built yet (NotImplementedError).
"""
from __future__ import annotations

import csv
import math
from typing import Optional

import numpy as np

from . import api
from .api import Context, GPSLCObject, HyperParameters


def getPriorParameters() -> dict:
    """src/hyperparameters.jl:38-70"""
    p = {}
    for name in ("uNoise", "xNoise", "tNoise", "yNoise", "xScale", "tScale", "yScale",
                 "uxLS", "utLS", "xtLS", "uyLS", "xyLS", "tyLS"):
        p[name + "Shape"] = 4.0
        p[name + "Scale"] = 4.0
    p["sigmaUNoise"] = 1.0e-13
    p["sigmaUCov"] = 1.0
    p["drift"] = 0.5
    return p


def getHyperParameters() -> HyperParameters:
    """src/hyperparameters.jl:85-102"""
    return HyperParameters()


def generateSigmaU(nIndividualsArray, eps=1e-13, cov=1.0):
    """src/utils.jl:17-33"""
    n = int(sum(nIndividualsArray))
    S = np.eye(n)
    i = 0
    for m in nIndividualsArray:
        S[i:i + m, i:i + m] = cov
        i += m
    S[np.diag_indices(n)] = 1 + eps
    return S


def removeAdjacent(v):
    """src/utils.jl:39-52"""
    out = []
    for e in v:
        if not out or e != out[-1]:
            out.append(e)
    return out


def prepareData(data, confounderEps=1.0e-13, confounderCov=1.0):
    """src/data.jl:20-70: CSV path (or dict of columns) -> SigmaU, obj, X, T, Y; rows sorted by `obj`
    (stable), covariates = every column that is not T / Y / obj."""
    if isinstance(data, str):
        with open(data, newline="") as f:
            rows = list(csv.DictReader(f))
        cols = {k: [r[k] for r in rows] for k in rows[0].keys()}
    else:
        cols = {k: list(v) for k, v in data.items()}
    n = len(cols["T"])
    order = np.arange(n)
    SigmaU = obj = None
    if "obj" in cols:
        order = np.array(sorted(range(n), key=lambda i: cols["obj"][i]))   # DataFrames.sort! is stable
        obj = [cols["obj"][i] for i in order]
        counts = {}
        for o in obj:
            counts[o] = counts.get(o, 0) + 1
        SigmaU = generateSigmaU([counts[o] for o in removeAdjacent(obj)], confounderEps, confounderCov)

    def num(col):
        vals = [cols[col][i] for i in order]
        if all(str(v).lower() in ("true", "false") for v in vals):
            return np.array([str(v).lower() == "true" for v in vals])
        return np.array([float(v) for v in vals])

    T, Y = num("T"), num("Y").astype(np.float64)
    xcols = [c for c in cols if c not in ("T", "Y", "obj")]
    X = np.column_stack([num(c).astype(np.float64) for c in xcols]) if xcols else None
    return SigmaU, obj, X, T, Y


# ---- scalar densities (Gen's inv_gamma(shape, scale)) -------------------------------------------------

def _invgamma_logpdf(x, shape, scale):
    if not x > 0:
        return -math.inf
    return shape * math.log(scale) - math.lgamma(shape) - (shape + 1.0) * math.log(x) - scale / x


def _proposal_params(cur, variance):
    """src/proposal.jl:32-41: InvGamma centred at `cur` with the given variance."""
    shape = (cur * cur / variance) + 2.0
    return shape, cur * (shape - 1.0)


def toMatrixModel(Ucols, n, nU):
    """What `toMatrix(U, n, nU)` (src/utils.jl:60-64) makes of the nU traced vectors inside the MODEL
    (src/model_likelihood.jl:7): permutedims(hcat(U...)) reshaped column-major to (n, nU), i.e. the vectors
    interleaved for nU >= 2 (derived from the source, unexecuted).  extractParameters
    (src/utils.jl:103-106) does NOT apply it, and neither does the prediction path."""
    H = np.column_stack(Ucols)            # n x nU, H[:, u] = traced vector u
    return np.asfortranarray(H.T.reshape(-1, order="F").reshape(n, nU, order="F"))


class _NoCovRealTChain:
    """State + node scores of CausalGPSLCNoCovRealT (src/model.jl:45-57)."""

    SCALARS = ("uNoise", "tNoise", "yNoise", "tyLS", "tScale", "yScale")

    def __init__(self, priorparams, SigmaU, T, Y, nU, rng, device=0):
        self.pp, self.SigmaU, self.T, self.Y, self.nU, self.rng = priorparams, SigmaU, T, Y, nU, rng
        self.n = len(Y)
        self.ctx = Context(self.n, 0, 0, device=device)
        self.ctx.set_data(None, np.zeros(self.n), np.zeros(self.n))
        api.mvnLogpdf(SigmaU, np.zeros((self.n, 0)), ctx=self.ctx)        # factor SigmaU once (cached)
        self.Lsig = np.linalg.cholesky(SigmaU)                            # for the slice's auxiliary draw
        ig = lambda name: priorparams[name + "Scale"] / rng.gamma(priorparams[name + "Shape"])   # noqa: E731
        # generate(): latent addresses from the prior (src/inference.jl:75), :T and :Y constrained
        self.v = {k: ig(k) for k in self.SCALARS}
        self.v["utLS"] = np.array([ig("utLS") for _ in range(nU)])
        self.v["uyLS"] = np.array([ig("uyLS") for _ in range(nU)])
        self.U = [math.sqrt(self.v["uNoise"]) * (self.Lsig @ rng.standard_normal(self.n)) for _ in range(nU)]
        self.s_u = self.score_u()
        self.s_t = self.score_t()
        self.s_y = self.score_y()

    # node scores (GPU)
    def score_u(self, uNoise=None, U=None):
        U = self.U if U is None else U
        un = self.v["uNoise"] if uNoise is None else uNoise
        return float(np.sum(api.mvnLogpdf(None, np.column_stack(U), covscale=np.full(self.nU, un), ctx=self.ctx)))

    def _umodel(self, U=None):
        return toMatrixModel(self.U if U is None else U, self.n, self.nU)

    def score_t(self, v=None, U=None):   # generateRealTfromU, src/model_likelihood.jl:55-60
        v = v or self.v
        return float(api.gpLogpdf(self._umodel(U), v["utLS"], v["tScale"], v["tNoise"], self.T, ctx=self.ctx)[0])

    def score_y(self, v=None, U=None):   # generateYfromUT, src/model_likelihood.jl:94-101
        v = v or self.v
        F = np.column_stack([self._umodel(U), self.T])
        return float(api.gpLogpdf(F, np.concatenate([v["uyLS"], [v["tyLS"]]]), v["yScale"], v["yNoise"], self.Y,
                                  ctx=self.ctx)[0])

    # which node scores an address touches
    TOUCH = {"uNoise": "u", "tNoise": "t", "yNoise": "y", "tyLS": "y", "tScale": "t", "yScale": "y",
             "utLS": "t", "uyLS": "y"}

    def mh(self, name, k=None):
        """One `mh(trace, paramProposal, (drift, addr))` (src/inference.jl:78-89)."""
        prior = name
        cur = self.v[name] if k is None else self.v[name][k]
        sh, sc = _proposal_params(cur, self.pp["drift"])
        new = sc / self.rng.gamma(sh)
        shb, scb = _proposal_params(new, self.pp["drift"])
        v2 = dict(self.v)
        if k is None:
            v2[name] = new
        else:
            arr = self.v[name].copy()
            arr[k] = new
            v2[name] = arr
        node = self.TOUCH[name]
        old_s = {"u": self.s_u, "t": self.s_t, "y": self.s_y}[node]
        try:
            new_s = (self.score_u(uNoise=new) if node == "u" else self.score_t(v2) if node == "t" else self.score_y(v2))
        except api.PosDefException:
            return False
        log_a = (new_s - old_s
                 + _invgamma_logpdf(new, self.pp[prior + "Shape"], self.pp[prior + "Scale"])
                 - _invgamma_logpdf(cur, self.pp[prior + "Shape"], self.pp[prior + "Scale"])
                 + _invgamma_logpdf(cur, shb, scb) - _invgamma_logpdf(new, sh, sc))
        if math.log(self.rng.random()) < log_a:
            self.v = v2
            if node == "u":
                self.s_u = new_s
            elif node == "t":
                self.s_t = new_s
            else:
                self.s_y = new_s
            return True
        return False

    def elliptical_slice(self, k):
        """`elliptical_slice(trace, :U => k => :U, zeros(n), uCov)` (src/inference.jl:92-98)."""
        nu = math.sqrt(self.v["uNoise"]) * (self.Lsig @ self.rng.standard_normal(self.n))
        log_y = self.s_t + self.s_y + math.log(self.rng.random())
        theta = self.rng.uniform(0.0, 2.0 * math.pi)
        lo, hi = theta - 2.0 * math.pi, theta
        f = self.U[k]
        for _ in range(200):
            prop = f * math.cos(theta) + nu * math.sin(theta)
            U2 = list(self.U)
            U2[k] = prop
            try:
                st, sy = self.score_t(U=U2), self.score_y(U=U2)
            except api.PosDefException:
                st = sy = -math.inf
            if st + sy > log_y:
                self.U, self.s_t, self.s_y = U2, st, sy
                self.s_u = self.score_u()
                return
            if theta < 0:
                lo = theta
            else:
                hi = theta
            theta = self.rng.uniform(lo, hi)
        # bracket collapsed onto the current state: keep it

    def snapshot(self):
        return {"uNoise": self.v["uNoise"], "tNoise": self.v["tNoise"], "yNoise": self.v["yNoise"],
                "tyLS": self.v["tyLS"], "tScale": self.v["tScale"], "yScale": self.v["yScale"],
                "utLS": self.v["utLS"].copy(), "uyLS": self.v["uyLS"].copy(), "U": [u.copy() for u in self.U]}


def Posterior(priorparams, X, T, Y, nU, nOuter, nMHInner, nESInner, seed=1234, device=0):
    """Posterior(priorparams, nothing, T::ContinuousTreatment, Y, nU, nOuter, nMHInner, nESInner)
    (src/inference.jl:62-102).  Returns the list of nOuter posterior samples (dicts keyed like the trace)."""
    if X is not None or nU is None or np.asarray(T).dtype == np.bool_:
        raise NotImplementedError("only CausalGPSLCNoCovRealT (latent confounders, no covariates, continuous "
                                  "treatment) is built so far; see DESIGN.md")
    rng = np.random.Generator(np.random.Philox(seed))
    ch = _NoCovRealTChain(priorparams, priorparams["SigmaU"], np.asarray(T, float), np.asarray(Y, float), nU, rng,
                          device=device)
    samples = []
    for _ in range(nOuter):
        for _ in range(nMHInner):
            ch.mh("uNoise"); ch.mh("tNoise"); ch.mh("yNoise"); ch.mh("tyLS")        # noqa: E702
            for k in range(nU):
                ch.mh("utLS", k); ch.mh("uyLS", k)                                      # noqa: E702
            ch.mh("tScale"); ch.mh("yScale")                                            # noqa: E702
        for _ in range(nESInner):
            for k in range(nU):
                ch.elliptical_slice(k)
        samples.append(ch.snapshot())
    return samples


def gpslc(data, hyperparams: Optional[HyperParameters] = None, priorparams: Optional[dict] = None, seed=1234,
          device=0) -> GPSLCObject:
    """gpslc(filename or columns; hyperparams, priorparams) (src/driver.jl:27-33): prepareData, run the chain,
    and return the GPSLCObject holding the retained posterior samples nBurnIn:stepSize:nOuter (burn-in index
    inclusive, src/estimation.jl:72,78) in the flat layout the prediction path consumes."""
    hp = hyperparams or getHyperParameters()
    pp = dict(priorparams or getPriorParameters())
    SigmaU, obj, X, T, Y = prepareData(data, pp["sigmaUNoise"], pp["sigmaUCov"])
    pp["SigmaU"] = SigmaU                                   # src/driver.jl:61
    post = Posterior(pp, X, T, Y, hp.nU if SigmaU is not None else None, hp.nOuter, hp.nMHInner, hp.nESInner,
                     seed=seed, device=device)
    keep = post[hp.nBurnIn - 1:hp.nOuter:hp.stepSize]
    S, n, nU = len(keep), len(Y), hp.nU
    U = np.zeros((n, nU, S), order="F")
    for s, smp in enumerate(keep):
        for u in range(nU):
            U[:, u, s] = smp["U"][u]                        # extractParameters: no interleave (src/utils.jl:103-106)
    g = GPSLCObject(None, T, Y, U, np.column_stack([smp["uyLS"] for smp in keep]), None,
                    np.array([smp["tyLS"] for smp in keep]), np.array([smp["yNoise"] for smp in keep]),
                    np.array([smp["yScale"] for smp in keep]), hyperparams=hp, device=device)
    g.posteriorSamples = post
    g.obj = obj
    g.SigmaU = SigmaU
    return g
