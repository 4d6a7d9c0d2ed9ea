"""Host-side posterior inference for the no-covariates continuous-treatment model, scoring every
Gaussian-process node on the GPU (BASELINE config 0).

Mirrors, for the four models with latent confounders — ``CausalGPSLCRealT`` / ``NoCovRealT`` (src/model.jl:11-27,
45-57, chains src/inference.jl:4-102) and ``CausalGPSLCBinaryT`` / ``NoCovBinaryT`` (src/model.jl:73-89, 109-120,
chains src/inference.jl:169-302):

    getPriorParameters / getHyperParameters      src/hyperparameters.jl:38-70, 85-102
    generateSigmaU, prepareData                  src/utils.jl:17-33, src/data.jl:20-70
    Posterior(priorparams, X | nothing, T, Y, ...) src/inference.jl:4-59, 62-102  (MH within Gibbs + elliptical slice)
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
tests/test_gpu_neec.py.  In the four no-U models the reference never constrains ``:X => k => :X`` (its
trace holds a prior draw of X, src/inference.jl:117-123); here X is the data, as in prediction.
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


def _parse_csv_column(vals):
    """The element type CSV.jl's inference would give a column: int, else float, else the strings themselves."""
    vals = list(vals)
    if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in vals):
        return [int(v) for v in vals]
    if all(isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool) for v in vals):
        return [float(v) for v in vals]
    strs = [str(v).strip() for v in vals]
    try:
        return [int(v) for v in strs]
    except ValueError:
        pass
    try:
        return [float(v) for v in strs]
    except ValueError:
        return strs


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
        # CSV.jl has already typed the column when DataFrames.sort! (src/data.jl:24) sees it: Int64 when every
        # value parses as an integer, else Float64 when every value parses as a number, else String — numeric
        # labels therefore sort numerically (1, 2, ..., 10), not lexicographically (1, 10, 11, ..., 2)
        keys = _parse_csv_column(cols["obj"])
        order = np.array(sorted(range(n), key=lambda i: keys[i]))          # DataFrames.sort! is stable
        obj = [keys[i] for i in order]
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


class _RealTChain:
    """State + node scores of the four models with latent confounders: CausalGPSLCRealT / NoCovRealT
    (src/model.jl:11-27, 45-57) and, with ``binary=True``, CausalGPSLCBinaryT / NoCovBinaryT (:73-89, 109-120)."""

    def __init__(self, priorparams, SigmaU, X, T, Y, nU, rng, device=0, binary=False):
        self.binary = binary          # CausalGPSLCBinaryT / NoCovBinaryT (src/model.jl:73-89, 109-120)
        self.Tb = np.asarray(T).astype(bool) if binary else None
        T = np.asarray(T, dtype=np.float64)   # Bool treatments promote to 0.0 / 1.0 in the :Y kernel
        nU = nU or 0                      # nU::Nothing -> the NoU models (src/model.jl:28-41, 58-67, 91-106, 122-131)
        self.pp, self.SigmaU, self.X, self.T, self.Y, self.nU, self.rng = priorparams, SigmaU, X, T, Y, nU, rng
        self.n = len(Y)
        self.nX = 0 if X is None else X.shape[1]
        self.ctx = Context(self.n, 0, 0, device=device)
        self.ctx.set_data(None, np.zeros(self.n), np.zeros(self.n))
        if nU:
            api.mvnLogpdf(SigmaU, np.zeros((self.n, 0)), ctx=self.ctx)    # SigmaU handed over once (cached in the ctx: scores
                                                                          # AND the slices' auxiliary draws, _u_draw)
        ig = lambda name: priorparams[name + "Scale"] / rng.gamma(priorparams[name + "Shape"])   # noqa: E731
        # generate(): latent addresses from the prior (src/inference.jl:20, :75); :T, :Y (and :X => k => :X) constrained
        v = {k: ig(k) for k in ("yNoise", "tyLS", "yScale")}
        if nU or self.nX:                 # the :T node has hyper-parameters only when something feeds it
            v.update({k: ig(k) for k in ("tNoise", "tScale")})
        if nU:
            v["uNoise"] = ig("uNoise")
            v["utLS"] = np.array([ig("utLS") for _ in range(nU)])
            v["uyLS"] = np.array([ig("uyLS") for _ in range(nU)])
        if self.nX:
            for name in ("xtLS", "xyLS"):
                v[name] = np.array([ig(name) for _ in range(self.nX)])
        if self.nX and nU:
            v["uxLS"] = np.array([[ig("uxLS") for _ in range(self.nX)] for _ in range(nU)])   # [u][k]: :uxLS => u => k
            for name in ("xNoise", "xScale"):
                v[name] = np.array([ig(name) for _ in range(self.nX)])
        self.v = v
        self.U = [self._u_draw(rng.standard_normal(self.n)) for _ in range(nU)]
        if binary:                        # :logitT from its prior, the :T => i => :T nodes are constrained
            self.logitT = self._t_draw(rng.standard_normal(self.n))
        self.s_u = self.score_u()
        self.s_x, self.s_t, self.s_y = self.score_xty()     # s_x: array over k (empty without covariates)
        self.s_b = self.score_b() if binary else 0.0

    # ---- binary treatments: :logitT ~ mvnormal(0, logitTCov), :T => i => :T ~ bernoulli(expit(logitT_i))
    def _t_features(self, U=None, v=None):
        """Features and lengthscales of the :T / :logitT node: [U, X], [U], [X] or nothing (T ~ N(0, I))."""
        v = v or self.v
        ls = ([v["utLS"]] if self.nU else []) + ([v["xtLS"]] if self.nX else [])
        if not ls:
            return np.zeros((self.n, 0)), np.zeros(0)
        if U is None:
            return self._cur()["Ft"], np.concatenate(ls)
        cols = ([self._umodel(U)] if self.nU else []) + ([self.X] if self.nX else [])
        return np.column_stack(cols), np.concatenate(ls)

    def _u_draw(self, z):
        """chol(SigmaU * uNoise) z — Gen's `mvnormal(zeros(n), uCov)` (generateUfromSigmaU, src/model_likelihood.jl:4-10; the
        auxiliary vector of the :U => k => :U slices, src/inference.jl:48-54) with the host's normals, from the covariance the
        ctx caches (gpslc_mvn_draw)."""
        return api.mvnDraw(None, z, covscale=[self.v["uNoise"]], ctx=self.ctx)[:, 0]

    def _t_draw(self, z):
        """chol(logitTCov) z — Gen's `mvnormal(zeros(n), logitTCov)` (src/model_likelihood.jl:25-33; the slice's auxiliary
        vector, src/inference.jl:225-232) with the host's normals: on the GPU, by the call that factors the node's covariance
        (gpslc_nodes_draw: one workgroup up to n = 640, the batched tiled factorisation beyond).  Without features the
        covariance is the identity (generateBinaryTfromPrior, src/model_prior.jl:195-200): the draw is z itself."""
        F, ls = self._t_features()
        if not F.shape[1]:
            return np.array(z, dtype=np.float64)
        return api.nodesDraw([(F, ls, self.v["tScale"], self.v["tNoise"], z)], self.ctx)[:, 0]

    def score_b(self, logitT=None):
        """sum_i log bernoulli(T_i; expit(logitT_i)) (generateBinaryT, src/model_prior.jl:21-24) — host scalar work."""
        l = self.logitT if logitT is None else logitT
        return float(-np.sum(np.logaddexp(0.0, np.where(self.Tb, -l, l))))

    def elliptical_slice_logitT(self):
        """`elliptical_slice(trace, :logitT, zeros(n), logitTCov)` (src/inference.jl:232)."""
        nu = self._t_draw(self.rng.standard_normal(self.n))
        log_y = self.s_b + math.log(self.rng.random())
        theta = self.rng.uniform(0.0, 2.0 * math.pi)
        lo, hi = theta - 2.0 * math.pi, theta
        f = self.logitT
        for _ in range(200):
            prop = f * math.cos(theta) + nu * math.sin(theta)
            sb = self.score_b(prop)
            if sb > log_y:
                self.logitT, self.s_b = prop, sb
                self.s_t = self.score_t()
                return
            if theta < 0:
                lo = theta
            else:
                hi = theta
            theta = self.rng.uniform(lo, hi)

    # ---- node scores (GPU) ------------------------------------------------------------------------
    def score_u(self, uNoise=None, U=None):
        """sum_k log N(U_k; 0, uNoise * SigmaU) (generateUfromSigmaU, src/model_likelihood.jl:4-10).  With
        Sigma = SigmaU fixed for the data set, log N(x; 0, c Sigma) = -(n log 2 pi + n log c + logdet Sigma + q / c) / 2
        with q = x' Sigma^-1 x: the GPU is asked once per value of U (covscale = 1, cached factor of SigmaU) for the
        quadratic forms, and a move of uNoise alone — one per sweep — is scored on the host from them."""
        if not self.nU:
            return 0.0
        un = self.v["uNoise"] if uNoise is None else uNoise
        n, l2pi = self.n, math.log(2.0 * math.pi)
        if getattr(self, "_ld_sigma", None) is None:        # logdet SigmaU from the score of x = 0
            self._ld_sigma = -2.0 * float(api.mvnLogpdf(None, np.zeros((n, 1)), covscale=np.ones(1), ctx=self.ctx)[0]) - n * l2pi
        Ucur = self.U if U is None else U
        c = getattr(self, "_q_cache", None)
        if U is not None or c is None or c[0] is not self.U:
            lp1 = api.mvnLogpdf(None, np.column_stack(Ucur), covscale=np.ones(self.nU), ctx=self.ctx)
            q = -2.0 * lp1 - n * l2pi - self._ld_sigma
            if U is None:
                self._q_cache = (self.U, q)
        else:
            q = c[1]
        return float(np.sum(-0.5 * (n * l2pi + n * math.log(un) + self._ld_sigma + q / un)))

    def _umodel(self, U=None):
        if U is None:
            return self._cur()["Um"]
        return toMatrixModel(U, self.n, self.nU)

    def _cur(self):
        """Feature blocks of the CURRENT latent state — the model-side U matrix, the :T node's [U | X] and the :Y node's
        [U | X | T], Fortran-ordered as the C ABI wants them — built once per value of self.U (the hyper-parameter
        sweeps score dozens of proposals on the same U; a slice move replaces the list self.U)."""
        c = getattr(self, "_cur_cache", None)
        if c is None or c["key"] is not self.U:
            Um = toMatrixModel(self.U, self.n, self.nU) if self.nU else None
            tx = ([Um] if self.nU else []) + ([self.X] if self.nX else [])
            Ft = np.asfortranarray(np.column_stack(tx)) if tx else np.zeros((self.n, 0))
            Fy = np.asfortranarray(np.column_stack(tx + [self.T]))
            c = self._cur_cache = {"key": self.U, "Um": Um, "Ft": Ft, "Fy": Fy}
        return c

    def _xcol(self, k):
        """Covariate k as ONE array object for the whole chain (the nodes of a call that share it are marshalled once)."""
        c = getattr(self, "_xcols", None)
        if c is None:
            c = self._xcols = [np.ascontiguousarray(self.X[:, j], dtype=np.float64) for j in range(self.nX)]
        return c[k]

    def _uxls_model(self, v):
        """`toMatrix(uxLS, nX, nU)` of src/model_prior.jl:110 -> (nX, nU); row k feeds X node k."""
        return toMatrixModel([v["uxLS"][u] for u in range(self.nU)], self.nX, self.nU)

    def score_x(self, v=None, U=None, only=None):
        """:X => k => :X for every covariate in ONE batched call (generateXfromU, src/model_likelihood.jl:13-22)."""
        if not self.nX or not self.nU:    # NoU models: X ~ N(0, I), no hyper-parameters (src/model_prior.jl:175-181)
            return np.zeros(0)
        v = v or self.v
        ls = self._uxls_model(v)                                  # (nX, nU)
        ks = range(self.nX) if only is None else [only]
        out = api.gpLogpdf(self._umodel(U), np.column_stack([ls[k] for k in ks]), v["xScale"][list(ks)],
                           v["xNoise"][list(ks)], self.X[:, list(ks)], ctx=self.ctx)
        return out if only is None else float(out[0])

    def score_t(self, v=None, U=None):   # generateRealTfromUX / fromU, src/model_likelihood.jl:36-44, 55-60
        v = v or self.v
        F, ls = self._t_features(U, v)
        target = self.logitT if self.binary else self.T      # :logitT for binary treatments, :T otherwise
        if not F.shape[1]:                # generateRealTfromPrior / BinaryTfromPrior: N(0, I), src/model_prior.jl:187-200
            return float(-0.5 * (target @ target) - 0.5 * self.n * math.log(2.0 * math.pi))
        return float(api.gpLogpdf(F, ls, v["tScale"], v["tNoise"], target, ctx=self.ctx)[0])

    def score_y(self, v=None, U=None):   # generateYfromUXT / UT, src/model_likelihood.jl:83-101
        v = v or self.v
        cols = ([self._umodel(U)] if self.nU else []) + ([self.X] if self.nX else []) + [self.T]
        ls = np.concatenate(([v["uyLS"]] if self.nU else []) + ([v["xyLS"]] if self.nX else []) + [[v["tyLS"]]])
        return float(api.gpLogpdf(np.column_stack(cols), ls, v["yScale"], v["yNoise"], self.Y, ctx=self.ctx)[0])

    def _xty_nodes(self, v, U):
        """Node descriptors of everything a move of U touches: the nX `:X => k => :X` nodes, `:T` / `:logitT` (when
        something feeds it) and `:Y`.  Returns (nodes, nx, has_t)."""
        nodes = []
        Um = self._umodel(U) if self.nU else None
        nx = self.nX if (self.nX and self.nU) else 0
        if nx:
            ls = self._uxls_model(v)
            for k in range(self.nX):
                nodes.append((Um, ls[k], v["xScale"][k], v["xNoise"][k], self._xcol(k)))
        Ft, lst = self._t_features(U, v)
        target = self.logitT if self.binary else self.T
        has_t = Ft.shape[1] > 0
        if has_t:
            nodes.append((Ft, lst, v["tScale"], v["tNoise"], target))
        if U is None:
            Fy = self._cur()["Fy"]
        else:
            Fy = np.column_stack(([Um] if self.nU else []) + ([self.X] if self.nX else []) + [self.T])
        lsy = np.concatenate(([v["uyLS"]] if self.nU else []) + ([v["xyLS"]] if self.nX else []) + [[v["tyLS"]]])
        nodes.append((Fy, lsy, v["yScale"], v["yNoise"], self.Y))
        return nodes, nx, has_t

    def _xty_split(self, out, nx, has_t):
        target = self.logitT if self.binary else self.T
        sx = np.array(out[:nx], dtype=np.float64)
        st = float(out[nx]) if has_t else float(-0.5 * (target @ target) - 0.5 * self.n * math.log(2.0 * math.pi))
        return sx, st, float(out[-1])

    def score_xty(self, v=None, U=None):
        """Every node a move of U touches — the nX `:X => k => :X` nodes, `:T` / `:logitT` and `:Y` — in ONE fused
        call (gpslc_nodes_logpdf): what a Gen `update` of `:U => k => :U` re-scores (src/inference.jl:48-54)."""
        v = v or self.v
        nodes, nx, has_t = self._xty_nodes(v, U)
        return self._xty_split(api.nodesLogpdf(nodes, self.ctx), nx, has_t)

    def score_xty_many(self, Us, v=None):
        """The same for several candidate values of U in ONE fused call (len(Us) x (nX + 2) nodes, one workgroup each);
        a candidate with a covariance that is not positive definite scores -inf on that node.  Returns a list of
        (s_x, s_t, s_y)."""
        v = v or self.v
        allnodes, per = [], 0
        for U in Us:
            nodes, nx, has_t = self._xty_nodes(v, U)
            per = len(nodes)
            allnodes += nodes
        out = api.nodesLogpdf(allnodes, self.ctx, fail_value=-math.inf)
        return [self._xty_split(out[i * per:(i + 1) * per], nx, has_t) for i in range(len(Us))]

    # which node an address touches
    TOUCH = {"uNoise": "u", "tNoise": "t", "yNoise": "y", "tyLS": "y", "tScale": "t", "yScale": "y",
             "utLS": "t", "uyLS": "y", "uxLS": "x", "xNoise": "x", "xScale": "x", "xtLS": "t", "xyLS": "y"}

    def _draw(self, name, i=None, j=None):
        """The random numbers of one MH move: the InvGamma drift proposal (src/proposal.jl:32-41) and the acceptance
        uniform, in that order.  Both depend only on the address's current value, so a sweep can draw them for all
        its addresses up front, in the reference's address order, whatever the schedule of the scores."""
        cur = self.v[name] if i is None else (self.v[name][i] if j is None else self.v[name][i][j])
        sh, sc = _proposal_params(cur, self.pp["drift"])
        return sc / self.rng.gamma(sh), math.log(self.rng.random())

    def _propose(self, name, i=None, j=None, draw=None, base=None):
        """The proposal half of one `mh(trace, paramProposal, (drift, addr))`.  ``base``: the parameter values the move
        starts from (default: the current state; a speculated state in `mh_speculative`)."""
        vb = self.v if base is None else base
        cur = vb[name] if i is None else (vb[name][i] if j is None else vb[name][i][j])
        sh, sc = _proposal_params(cur, self.pp["drift"])
        new, log_unif = draw if draw is not None else self._draw(name, i, j)
        shb, scb = _proposal_params(new, self.pp["drift"])
        v2 = dict(vb)
        if i is not None:
            arr = vb[name].copy()
            if j is None:
                arr[i] = new
            else:
                arr[i][j] = new
            v2[name] = arr
        else:
            v2[name] = new
        node = self.TOUCH[name]
        xk = (j if name == "uxLS" else i) if node == "x" else None
        log_q = (_invgamma_logpdf(new, self.pp[name + "Shape"], self.pp[name + "Scale"])
                 - _invgamma_logpdf(cur, self.pp[name + "Shape"], self.pp[name + "Scale"])
                 + _invgamma_logpdf(cur, shb, scb) - _invgamma_logpdf(new, sh, sc))
        return {"name": name, "i": i, "j": j, "new": new, "v2": v2, "node": node, "xk": xk, "log_q": log_q,
                "log_unif": log_unif}

    def _old_score(self, pr):
        return self.s_x[pr["xk"]] if pr["node"] == "x" else {"u": self.s_u, "t": self.s_t, "y": self.s_y}[pr["node"]]

    def _decide(self, pr, new_s):
        """The accept / reject half: `new_s` = the touched node's score under the proposal (-inf: not positive
        definite, which Gen would have thrown on; here the move is rejected)."""
        if not pr["log_unif"] < new_s - self._old_score(pr) + pr["log_q"]:
            return False
        name, i, j, new = pr["name"], pr["i"], pr["j"], pr["new"]
        if i is None:
            self.v[name] = new
        else:                             # element-wise: other entries of the array may have moved in the same batch
            arr = self.v[name].copy()
            if j is None:
                arr[i] = new
            else:
                arr[i][j] = new
            self.v[name] = arr
        if pr["node"] == "u":
            self.s_u = new_s
        elif pr["node"] == "t":
            self.s_t = new_s
        elif pr["node"] == "y":
            self.s_y = new_s
        else:
            self.s_x = self.s_x.copy()
            self.s_x[pr["xk"]] = new_s
        return True

    def mh(self, name, i=None, j=None, draw=None):
        """One `mh(trace, paramProposal, (drift, addr))` (src/inference.jl:22-45, :78-89): `i`, `j` as in
        getProposalAddress (src/proposal.jl:7-24), 0-based here.  One score call for the one node it touches."""
        pr = self._propose(name, i, j, draw)
        try:
            if pr["node"] == "u":
                new_s = self.score_u(uNoise=pr["new"])
            elif pr["node"] == "t":
                new_s = self.score_t(pr["v2"])
            elif pr["node"] == "y":
                new_s = self.score_y(pr["v2"])
            else:
                new_s = self.score_x(pr["v2"], only=pr["xk"])
        except api.PosDefException:
            new_s = -math.inf
        return self._decide(pr, new_s)

    def _gp_node(self, pr):
        """(F, LS, scale, noise, target) of the node a proposal touches, under the proposal's parameters."""
        v2 = pr["v2"]
        if pr["node"] == "x":
            k = pr["xk"]
            return (self._umodel(), self._uxls_model(v2)[k], v2["xScale"][k], v2["xNoise"][k], self._xcol(k))
        if pr["node"] == "t":
            F, ls = self._t_features(None, v2)
            return (F, ls, v2["tScale"], v2["tNoise"], self.logitT if self.binary else self.T)
        ls = np.concatenate(([v2["uyLS"]] if self.nU else []) + ([v2["xyLS"]] if self.nX else []) + [[v2["tyLS"]]])
        return (self._cur()["Fy"], ls, v2["yScale"], v2["yNoise"], self.Y)

    def mh_batch(self, addrs, draws=None):
        """MH moves on addresses that touch DIFFERENT nodes, scored in one fused call (gpslc_nodes_logpdf: one
        workgroup per node, one launch).  Given U (and logitT) the model's density factorises over its nodes and no
        two nodes share a hyper-parameter, so these moves neither see nor affect one another: taking them together
        is the same Markov kernel as taking them one after the other."""
        props = [self._propose(*(tuple(a) + (None,) * (3 - len(a))), None if draws is None else draws[a]) for a in addrs]
        assert len({(p_["node"], p_["xk"]) for p_ in props}) == len(props), "addresses of one batch must touch distinct nodes"
        gp = [p_ for p_ in props if p_["node"] != "u"]
        scores = {}
        if gp:
            out = api.nodesLogpdf([self._gp_node(p_) for p_ in gp], self.ctx, fail_value=-math.inf)
            for p_, sc in zip(gp, out):
                scores[id(p_)] = float(sc)
        acc = 0
        for p_ in props:
            if p_["node"] == "u":
                try:
                    new_s = self.score_u(uNoise=p_["new"])
                except api.PosDefException:
                    new_s = -math.inf
            else:
                new_s = scores[id(p_)]
            acc += self._decide(p_, new_s)
        return acc

    def mh_speculative(self, segments, draws):
        """Several consecutive MH moves of every per-node address chain in ONE fused call.  For a segment a_0 .. a_{D-1}
        of a chain the proposal of a_l is scored under every accept / reject outcome of a_0 .. a_{l-1} (2^D - 1 nodes
        per chain: an accepted move changes the parameters the next one starts from); the outcomes are then decided in
        order with the pre-drawn uniforms, each against the score its predecessor left — the decisions, and therefore
        the chain, are those of the one-move-at-a-time schedule, in ceil(len / D) launches instead of len."""
        nodes, index, plans = [], {}, []
        for seg in segments:
            tree = {}

            def build(level, prefix, base):
                a = seg[level]
                pr = self._propose(*(tuple(a) + (None,) * (3 - len(a))), draw=draws[a], base=base)
                tree[prefix] = pr
                if pr["node"] != "u":
                    index[id(pr)] = len(nodes)
                    nodes.append(self._gp_node(pr))
                if level + 1 < len(seg):
                    build(level + 1, prefix + (0,), base)
                    build(level + 1, prefix + (1,), pr["v2"])

            build(0, (), self.v)
            plans.append((seg, tree))
        out = api.nodesLogpdf(nodes, self.ctx, fail_value=-math.inf) if nodes else []
        acc = 0
        for seg, tree in plans:
            prefix = ()
            for _ in seg:
                pr = tree[prefix]
                if pr["node"] == "u":
                    try:
                        new_s = self.score_u(uNoise=pr["new"])
                    except api.PosDefException:
                        new_s = -math.inf
                else:
                    new_s = float(out[index[id(pr)]])
                ok = self._decide(pr, new_s)
                acc += ok
                prefix += (1 if ok else 0,)
        return acc

    mh_depth = 2        # consecutive moves of a chain scored speculatively per fused call (2^depth - 1 nodes per chain)

    slice_depth = 8     # candidates of a slice's all-rejected path scored per fused call

    def elliptical_slice(self, k, depth=None):
        """`elliptical_slice(trace, :U => k => :U, zeros(n), uCov)` (src/inference.jl:48-54, :92-98).
        The shrinking bracket depends only on the angles tried, not on their scores: the first ``depth`` candidates of
        the all-rejected path are therefore known up front and are scored in ONE fused call (depth x (nX + 2) nodes, one
        workgroup each); the first one above the slice level is the move, exactly as if they had been tried one by
        one — the generator is rewound to where the one-by-one loop would have left it, so ``depth`` changes the number
        of launches (≈ 10 tries per slice on IHDP), not the chain.  ``depth=1`` is the reference's schedule."""
        rng = self.rng
        depth = self.slice_depth if depth is None else depth
        nu = self._u_draw(rng.standard_normal(self.n))
        log_y = float(np.sum(self.s_x)) + self.s_t + self.s_y + math.log(rng.random())
        theta = rng.uniform(0.0, 2.0 * math.pi)
        lo, hi = theta - 2.0 * math.pi, theta
        f = self.U[k]
        tries = 0
        while tries < 200:
            d = max(1, min(depth, 200 - tries))
            state = rng.bit_generator.state            # generator right after `theta` was drawn
            thetas, brackets = [theta], []
            t, l, h = theta, lo, hi
            for _ in range(d - 1):                     # the path taken if every candidate so far is rejected
                if t < 0:
                    l = t
                else:
                    h = t
                brackets.append((l, h))
                t = rng.uniform(l, h)
                thetas.append(t)
            cands = []
            for t in thetas:
                U2 = list(self.U)
                U2[k] = f * math.cos(t) + nu * math.sin(t)
                cands.append(U2)
            scores = self.score_xty_many(cands) if d > 1 else [self._score_xty_or_inf(cands[0])]
            for j, (sx, st, sy) in enumerate(scores):
                if float(np.sum(sx)) + st + sy > log_y:
                    rng.bit_generator.state = state    # rewind, then consume exactly the j draws the one-by-one loop made
                    for (l, h) in brackets[:j]:
                        rng.uniform(l, h)
                    self.U, self.s_x, self.s_t, self.s_y = cands[j], sx, st, sy
                    self.s_u = self.score_u()
                    return
            # all d rejected: shrink past the last one and go on
            t = thetas[-1]
            if brackets:
                lo, hi = brackets[-1]
            if t < 0:
                lo = t
            else:
                hi = t
            theta = rng.uniform(lo, hi)
            tries += d
        # bracket collapsed onto the current state: keep it

    def _score_xty_or_inf(self, U2):
        try:
            return self.score_xty(U=U2)
        except api.PosDefException:
            return np.zeros(0), -math.inf, -math.inf

    def sweep_addresses(self):
        """The addresses of one inner sweep in the reference's order (src/inference.jl:23-44 / :78-89 with U,
        :126-139 / :324-337 NoU with covariates, :158-160 / :372-374 NoU NoCov)."""
        if not self.nU and not self.nX:
            return [("yNoise",), ("tyLS",), ("yScale",)]
        out = [("uNoise",)] if self.nU else []
        out += [("tNoise",), ("yNoise",), ("tyLS",)]
        for k in range(self.nU):
            out += [("utLS", k), ("uyLS", k)]
            out += [("uxLS", k, l) for l in range(self.nX)]
        for k in range(self.nX):
            if self.nU:
                out += [("xNoise", k), ("xtLS", k), ("xyLS", k), ("xScale", k)]
            else:
                out += [("xtLS", k), ("xyLS", k)]
        return out + [("tScale",), ("yScale",)]

    def sweep_mh(self, batched=True, depth=None):
        """One inner sweep.  ``batched=False``: address by address, one score call each — the reference's schedule.
        ``batched=True`` (default): the sweep is split into its per-node address chains (the :Y chain yNoise, tyLS,
        uyLS.., xyLS.., yScale; the :T chain; one chain per :X => k => :X; uNoise for the U prior), each kept in
        the reference's order, and step t of every chain is scored in ONE fused call (`mh_batch`): 2 + nU + nX + 1
        calls per sweep instead of one per address (IHDP, nU = 1, nX = 6: 10 instead of 38).  Moves on different
        nodes commute (see `mh_batch`), so this is the same transition kernel — and since an address's random
        numbers depend only on its own current value, they are drawn up front in the reference's address order:
        both schedules produce the SAME chain, bit for bit (tests/test_gpu_neec.py).  ``depth`` (default `mh_depth`)
        consecutive moves of every chain are scored per call under all their accept / reject outcomes
        (`mh_speculative`): 5 launches per IHDP sweep at the default depth 2 (deeper trees cost more host staging than
        they save in launches: tools/bench_slice_depth.py)."""
        addrs = self.sweep_addresses()
        draws = {a: self._draw(*a) for a in addrs}
        if not batched:
            for a in addrs:
                self.mh(*(tuple(a) + (None,) * (3 - len(a))), draw=draws[a])
            return
        chains = {}
        for a in addrs:
            node = self.TOUCH[a[0]]
            key = (node, (a[2] if a[0] == "uxLS" else a[1]) if node == "x" else None)
            chains.setdefault(key, []).append(a)
        d = max(1, int(self.mh_depth if depth is None else depth))
        for t in range(0, max(len(c) for c in chains.values()), d):
            segs = [c[t:t + d] for c in chains.values() if t < len(c)]
            if d == 1:
                self.mh_batch([sg[0] for sg in segs], draws)
            else:
                self.mh_speculative(segs, draws)

    def snapshot(self):
        out = {k: (val.copy() if isinstance(val, np.ndarray) else val) for k, val in self.v.items()}
        out["U"] = [u.copy() for u in self.U]
        if self.binary:
            out["logitT"] = self.logitT.copy()
        return out


def Posterior(priorparams, X, T, Y, nU, nOuter, nMHInner, nESInner, seed=1234, device=0):
    """Posterior(priorparams, X or nothing, T::ContinuousTreatment, Y, nU, nOuter, nMHInner, nESInner)
    (src/inference.jl:4-59 with covariates, :62-102 without).  Returns the list of nOuter posterior samples
    (dicts keyed like the trace)."""
    if nU is not None and priorparams.get("SigmaU") is None:
        raise TypeError("nU given but priorparams['SigmaU'] is nothing: no matching Posterior method "
                        "(pass HyperParameters(nU=None) for data without object labels)")
    rng = np.random.Generator(np.random.Philox(seed))
    Xa = None if X is None else np.asarray(X, float).reshape(len(Y), -1)
    binary = np.asarray(T).dtype == np.bool_
    ch = _RealTChain(priorparams, priorparams["SigmaU"], Xa, np.asarray(T), np.asarray(Y, float), nU, rng,
                     device=device, binary=binary)
    samples = []
    no_u_no_cov = nU is None and Xa is None           # one MH sweep per outer iteration, no slice (:157-163, :371-377)
    for _ in range(nOuter):
        for _ in range(1 if no_u_no_cov else nMHInner):
            ch.sweep_mh()
        for _ in range(0 if no_u_no_cov else nESInner):
            if binary:
                ch.elliptical_slice_logitT()          # src/inference.jl:232, :348
            for k in range(nU or 0):
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
    nU = hp.nU if SigmaU is not None else None
    if SigmaU is None:                                      # GPSLCObject constructors, src/types.jl:277-289
        import dataclasses
        hp = dataclasses.replace(hp, nU=None)
    post = Posterior(pp, X, T, Y, nU, hp.nOuter, hp.nMHInner, hp.nESInner, seed=seed, device=device)
    keep = post[hp.nBurnIn - 1:hp.nOuter:hp.stepSize]
    S, n = len(keep), len(Y)
    U = uyLS = None
    if nU:
        U = np.zeros((n, nU, S), order="F")
        for s, smp in enumerate(keep):
            for u in range(nU):
                U[:, u, s] = smp["U"][u]                    # extractParameters: no interleave (src/utils.jl:103-106)
        uyLS = np.column_stack([smp["uyLS"] for smp in keep])
    xyLS = None if X is None else np.column_stack([smp["xyLS"] for smp in keep])
    g = GPSLCObject(X, np.asarray(T, dtype=np.float64), Y, U, uyLS, xyLS,
                    np.array([smp["tyLS"] for smp in keep]), np.array([smp["yNoise"] for smp in keep]),
                    np.array([smp["yScale"] for smp in keep]), hyperparams=hp, device=device)
    g.posteriorSamples = post
    g.obj = obj
    g.SigmaU = SigmaU
    return g
