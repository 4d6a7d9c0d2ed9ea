"""CPU restatement of the CausalGPSLC.jl prediction / :Y-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported CPU baseline.  The product
path (``causalgpslc.jl_amd``) never routes through this file and fails loudly
when the HIP library is missing.

PINNING STATUS.  The reference is Julia; no Julia runtime exists in this
pipeline, so the reference itself cannot be executed.  This restatement is
pinned against every known-answer test the reference holds for the path
(tests/test_oracle_reference_answers.py lists them with file:line), but those
only cover the kernel matrix, ``processCov``, the exact-zero identities at
n = 1, the jitter placement and ``generateSigmaU``.  No reference test pins a
non-trivial MeanITE / CovITE / SATE / log-density value, so for N > 1 the
numerics are **parity unpinned** by the reference; they are cross-checked here
by two independent restatements (literal, structured) and an extended-precision
evaluation (``*_longdouble``).

All citations are relative to /root/reference/.

Two forms are provided:

* ``literal``  – the reference's arithmetic in the reference's own order
  (5 log-kernels, exp*scale, symmetric-indefinite solves = LAPACK dsysv, the
  same routine family Julia's ``Symmetric \\`` dispatches to, four C blocks,
  upper-triangle symmetrisation + jitter, one Cholesky per draw).
* ``structured`` – one Cholesky of A per posterior sample + Schur complement,
  the algorithm the HIP path runs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import scipy.linalg as sla

# ----------------------------------------------------------------------------
# kernel.jl
# ----------------------------------------------------------------------------


def rbf_kernel_log_scalar(xi, xip, ls):
    """src/kernel.jl:13-19  ``-sum((Xi .- Xiprime).^2 ./ LS.^2)`` (no 1/2, /LS^2)."""
    xi = np.atleast_1d(np.asarray(xi, dtype=np.float64))
    xip = np.atleast_1d(np.asarray(xip, dtype=np.float64))
    ls = np.asarray(ls, dtype=np.float64)
    if ls.ndim != 0 and ls.shape[0] != xi.shape[0]:
        raise AssertionError("vector lengthscale doesn't match individual")
    terms = (xi - xip) ** 2 / ls ** 2
    acc = 0.0
    for t in terms:  # sequential sum over the (short) feature dimension
        acc += float(t)
    return -acc


def _as_2d(X):
    X = np.asarray(X)
    if X.dtype == np.bool_:
        X = X.astype(np.float64)
    X = X.astype(np.float64, copy=False)
    if X.ndim == 1:
        X = X[:, None]
    return X


def rbf_kernel_log(X1, X2, ls):
    """src/kernel.jl:24-32 (matrix/vector method) and :34-42 (vector-of-vectors).

    Dense n x n log-Gram, both triangles.  Same per-entry arithmetic as the
    double loop over ``rbfKernelLogScalar`` (difference, square, divide by LS^2,
    sequential sum over the feature dimension, negate), vectorised over (i, ip).
    """
    A = _as_2d(X1)
    B = _as_2d(X2)
    if A.shape != B.shape:
        raise AssertionError("X1 and X2 are different sizes!")
    n, d = A.shape
    ls = np.asarray(ls, dtype=np.float64)
    if ls.ndim == 0:
        ls2 = np.full(d, float(ls) ** 2)
    else:
        if ls.shape[0] != d:
            raise AssertionError("vector lengthscale doesn't match individual")
        ls2 = ls.astype(np.float64) ** 2
    acc = np.zeros((n, n))
    for k in range(d):
        diff = A[:, k][:, None] - B[:, k][None, :]
        acc += diff * diff / ls2[k]
    return -acc


def logit(p):
    """src/kernel.jl:46"""
    return math.log(p / (1 - p))


def expit(x):
    """src/kernel.jl:49"""
    return math.exp(x) / (1.0 + math.exp(x))


def process_cov(logcov, scale, noise=None):
    """src/kernel.jl:53-55 (with noise) and :57-59 (without)."""
    out = np.exp(np.asarray(logcov, dtype=np.float64)) * scale
    if noise is not None:
        out = out + np.eye(out.shape[0]) * noise
    return out


# ----------------------------------------------------------------------------
# utils.jl / hyperparameters.jl pieces that define the input contract
# ----------------------------------------------------------------------------


def generate_sigma_u(n_individuals, eps=1e-13, cov=1.0):
    """src/utils.jl:17-33"""
    n = int(sum(n_individuals))
    S = np.eye(n)
    i = 0
    for m in n_individuals:
        S[i:i + m, i:i + m] = cov
        i += m
    S[np.diag_indices(n)] = 1 + eps
    return S


PREDICTION_COVARIANCE_NOISE = 1e-10  # src/hyperparameters.jl:92


def num_posterior_samples(n_burn_in=10, step_size=1, n_outer=24):
    """src/utils.jl:156-161 / src/estimation.jl:72: length(nBurnIn:stepSize:nOuter),
    burn-in index inclusive (defaults -> 15)."""
    if n_outer < n_burn_in:
        return 0
    return (n_outer - n_burn_in) // step_size + 1


@dataclass
class PosteriorSample:
    """What ``extractParameters`` (src/utils.jl:92-124) returns for one sample."""
    uyLS: Optional[np.ndarray]   # (nU,) or None
    xyLS: Optional[np.ndarray]   # (nX,) or None
    tyLS: float
    yNoise: float
    yScale: float
    U: Optional[np.ndarray]      # (n, nU) or None


# ----------------------------------------------------------------------------
# likelihood.jl  (literal)
# ----------------------------------------------------------------------------


def _sym_solve(A, B):
    """Julia 1.7 ``Symmetric(A) \\ B`` -> Bunch-Kaufman (dsytrf/dsytrs); LAPACK
    dsysv is the same factorisation + solve."""
    return sla.solve(A, B, assume_a="sym", lower=False, check_finite=False)


def likelihood_distribution(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT):
    """src/likelihood.jl:8-52 (U+X), :55-94 (U only), :97-136 (X only), :139-174 (T only).

    The four methods differ only in which log-kernels are summed.
    """
    Y = np.asarray(Y, dtype=np.float64)
    n = Y.shape[0]
    Tm = _as_2d(T)
    assert Tm.shape[0] == n
    base = np.zeros((n, n))
    if U is not None:
        Um = _as_2d(U)
        assert Um.shape[0] == n and np.atleast_1d(uyLS).shape[0] == Um.shape[1]
        base = base + rbf_kernel_log(Um, Um, np.atleast_1d(uyLS))      # :24
    if X is not None:
        Xm = _as_2d(X)
        assert Xm.shape[0] == n and np.atleast_1d(xyLS).shape[0] == Xm.shape[1]
        base = base + rbf_kernel_log(Xm, Xm, np.atleast_1d(xyLS))      # :25
    doTv = np.full((n, 1), float(doT))
    tyCovLog = rbf_kernel_log(Tm, Tm, tyLS)                            # :26
    tyCovLogS = rbf_kernel_log(Tm, doTv, tyLS)                         # :27
    tyCovLogSS = rbf_kernel_log(doTv, doTv, tyLS)                      # :28

    CovWW = process_cov(base + tyCovLog, yScale, 0.0)                  # :30-31
    CovWWp = CovWW + yNoise * np.eye(n)                                # :32
    CovWWs = process_cov(base + tyCovLogS, yScale, 0.0)                # :35
    CovWsWs = process_cov(base + tyCovLogSS, yScale, 0.0)              # :38-39

    CovWWpInvCovWW = _sym_solve(CovWWp, CovWW)                         # :42
    CovWWpInvCovWWs = _sym_solve(CovWWp, CovWWs)                       # :43

    CovC11 = CovWW - CovWW @ CovWWpInvCovWW                            # :46
    CovC12 = CovWWs - CovWW @ CovWWpInvCovWWs                          # :47
    CovC21 = CovWWs.T - CovWWs.T @ CovWWpInvCovWW                      # :48
    CovC22 = CovWsWs - CovWWs.T @ CovWWpInvCovWWs                      # :49
    return Y, CovWW, CovWWs, CovWWp, CovC11, CovC12, CovC21, CovC22


# ----------------------------------------------------------------------------
# estimation.jl (literal)
# ----------------------------------------------------------------------------


def conditional_ite(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT):
    """src/estimation.jl:36-50"""
    Y, CovWW, CovWWs, CovWWp, C11, C12, C21, C22 = likelihood_distribution(
        uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT)
    MeanITE = (CovWWs.T - CovWW) @ _sym_solve(CovWWp, Y)               # :46
    CovITE = C11 - C12 - C21 + C22                                     # :47
    return MeanITE, CovITE


def _symmetric_upper(M):
    """LinearAlgebra.Symmetric(M): the upper triangle wins (src/estimation.jl:82)."""
    Uq = np.triu(M)
    return Uq + np.triu(M, 1).T


def ite_distributions(samples: Sequence[PosteriorSample], X, T, Y, doT,
                      pred_noise=PREDICTION_COVARIANCE_NOISE):
    """src/estimation.jl:66-86.  ``samples`` is the already burn-in/step filtered list."""
    n = np.asarray(Y).shape[0]
    S = len(samples)
    MeanITEs = np.zeros((S, n))
    CovITEs = np.zeros((S, n, n))
    for idx, p in enumerate(samples):
        m, C = conditional_ite(p.uyLS, p.xyLS, p.tyLS, p.yNoise, p.yScale, p.U, X, T, Y, doT)
        MeanITEs[idx] = m
        CovITEs[idx] = _symmetric_upper(C) + np.eye(n) * pred_noise    # :82
    return MeanITEs, CovITEs


def ite_samples(MeanITEs, CovITEs, spp, z):
    """src/estimation.jl:95-109.  Gen ``mvnormal(mean, cov)`` = mean + chol_lower(cov) * z
    (Distributions/PDMats ``unwhiten``); ``z`` (n, S*spp) are the standard normals,
    column order: posterior sample outer, draw inner (:100-107).  One Cholesky PER DRAW
    in the reference; the factor is identical across the spp draws of a sample."""
    S, n = MeanITEs.shape
    out = np.zeros((n, S * spp))
    i = 0
    for j in range(S):
        L = np.linalg.cholesky(CovITEs[j])   # raises LinAlgError ~ PosDefException
        for _ in range(spp):
            out[:, i] = MeanITEs[j] + L @ z[:, i]
            i += 1
    return out


def conditional_sate(MeanITE, CovITE):
    """src/estimation.jl:116-121"""
    n = MeanITE.shape[0]
    return float(np.sum(MeanITE) / n), float(np.sum(CovITE) / n ** 2)


def sate_distributions(samples, X, T, Y, doT, pred_noise=PREDICTION_COVARIANCE_NOISE):
    """src/estimation.jl:127-140 (CovITE here already carries the jitter)."""
    MeanITEs, CovITEs = ite_distributions(samples, X, T, Y, doT, pred_noise)
    S = MeanITEs.shape[0]
    ms = np.zeros(S)
    vs = np.zeros(S)
    for i in range(S):
        ms[i], vs[i] = conditional_sate(MeanITEs[i], CovITEs[i])
    return ms, vs


def sate_samples(MeanSATEs, VarSATEs, spp, z):
    """src/estimation.jl:148-163.  ``normal(mean, var)``: the VARIANCE is passed where
    Gen's normal expects a standard deviation (:159) -> sample = mean + var * z."""
    S = len(MeanSATEs)
    out = np.zeros(S * spp)
    i = 0
    for j in range(S):
        for _ in range(spp):
            out[i] = MeanSATEs[j] + VarSATEs[j] * z[i]
            i += 1
    return out


# ----------------------------------------------------------------------------
# driver.jl / prediction.jl (literal)
# ----------------------------------------------------------------------------


def sample_ite(samples, X, T, Y, doT, spp, z, pred_noise=PREDICTION_COVARIANCE_NOISE):
    """src/driver.jl:86-89"""
    M, C = ite_distributions(samples, X, T, Y, doT, pred_noise)
    return ite_samples(M, C, spp, z)


def sample_sate(samples, X, T, Y, doT, spp, z, pred_noise=PREDICTION_COVARIANCE_NOISE):
    """src/driver.jl:108-111"""
    m, v = sate_distributions(samples, X, T, Y, doT, pred_noise)
    return sate_samples(m, v, spp, z)


def do_t_range(min_do_t, max_do_t, fidelity):
    """src/prediction.jl:24-28: ``minDoT:(|max-min|/fidelity):maxDoT``.
    Julia's float range has floor((max-min)/step)+1 points start + i*step (with
    its twice-precision fix-up the end point is hit exactly when it is a multiple)."""
    delta = abs(max_do_t - min_do_t)
    step = delta / fidelity
    if step == 0:
        raise ValueError("step cannot be zero")  # Julia: ArgumentError
    if max_do_t < min_do_t:
        return np.zeros(0)
    npts = int(math.floor((max_do_t - min_do_t) / step + 1e-9)) + 1
    return min_do_t + step * np.arange(npts)


def predict_counterfactual_effects(samples, X, T, Y, spp, z, fidelity=100,
                                   min_do_t=None, max_do_t=None,
                                   pred_noise=PREDICTION_COVARIANCE_NOISE):
    """src/prediction.jl:23-36.  ``z`` (L, n, S*spp).  Returns ite (L, n, S*spp), doTrange."""
    Tn = np.asarray(T, dtype=np.float64)
    lo = float(Tn.min()) if min_do_t is None else float(min_do_t)
    hi = float(Tn.max()) if max_do_t is None else float(max_do_t)
    rng = do_t_range(lo, hi, fidelity)
    n = np.asarray(Y).shape[0]
    ite = np.zeros((len(rng), n, len(samples) * spp))
    for i, doT in enumerate(rng):
        ite[i] = sample_ite(samples, X, T, Y, doT, spp, z[i], pred_noise)
    return ite, rng


def julia_quantile(v, p):
    """Statistics.quantile(v, p) of Julia 1.7 (type 7, alpha = beta = 1): sort, aleph = n p + (1 - p),
    j = clamp(trunc(aleph), 1, n-1), gamma = clamp(aleph - j, 0, 1), v[j] + gamma (v[j+1] - v[j])."""
    v = np.sort(np.asarray(v, dtype=np.float64))
    n = v.shape[0]
    if n == 1:
        return float(v[0])
    aleph = n * p + (1.0 - p)
    j = min(max(int(aleph), 1), n - 1)
    gam = min(max(aleph - j, 0.0), 1.0)
    return float(v[j - 1] + gam * (v[j] - v[j - 1]))


def summarize_estimates(samples, credible_interval=0.90):
    """src/driver.jl:129-149: mean(samples, dims=2) and the two quantiles per individual."""
    lo = (1 - credible_interval) / 2
    hi = 1 - lo
    samples = np.asarray(samples, dtype=np.float64)
    return (samples.mean(axis=1),
            np.array([julia_quantile(r, lo) for r in samples]),
            np.array([julia_quantile(r, hi) for r in samples]))


# ----------------------------------------------------------------------------
# model_likelihood.jl :Y node
# ----------------------------------------------------------------------------


def y_cov(uyLS, xyLS, tyLS, yScale, yNoise, U, X, T):
    """Ycov of src/model_likelihood.jl:83-91 / 94-101 / 104-111 / 114-120."""
    Tm = _as_2d(T)
    n = Tm.shape[0]
    acc = np.zeros((n, n))
    if U is not None:
        acc = acc + rbf_kernel_log(_as_2d(U), _as_2d(U), np.atleast_1d(uyLS))
    if X is not None:
        acc = acc + rbf_kernel_log(_as_2d(X), _as_2d(X), np.atleast_1d(xyLS))
    acc = acc + rbf_kernel_log(Tm, Tm, tyLS)
    return process_cov(acc, yScale, yNoise)


def y_logpdf(uyLS, xyLS, tyLS, yScale, yNoise, U, X, T, Y):
    """Score contribution of the ``:Y`` node: logpdf(MvNormal(0, Symmetric(Ycov)), Y)
    = -1/2 (n log 2pi + logdet + Y' Ycov^-1 Y)  (Gen mvnormal -> Distributions/PDMats)."""
    C = y_cov(uyLS, xyLS, tyLS, yScale, yNoise, U, X, T)
    Y = np.asarray(Y, dtype=np.float64)
    L = np.linalg.cholesky(C)
    zz = sla.solve_triangular(L, Y, lower=True, check_finite=False)
    n = Y.shape[0]
    return float(-0.5 * (n * math.log(2 * math.pi) + 2 * np.sum(np.log(np.diag(L))) + zz @ zz))


# ----------------------------------------------------------------------------
# structured restatement (one Cholesky per posterior sample + Schur form)
# ----------------------------------------------------------------------------


def _base_and_e(p: PosteriorSample, X, T, dtype=np.float64):
    Tm = _as_2d(T).astype(dtype)
    n = Tm.shape[0]
    lux = np.zeros((n, n), dtype=dtype)
    if p.U is not None:
        Um = _as_2d(p.U).astype(dtype)
        for k in range(Um.shape[1]):
            d = Um[:, k][:, None] - Um[:, k][None, :]
            lux += d * d / dtype(np.atleast_1d(p.uyLS)[k]) ** 2
    if X is not None:
        Xm = _as_2d(X).astype(dtype)
        for k in range(Xm.shape[1]):
            d = Xm[:, k][:, None] - Xm[:, k][None, :]
            lux += d * d / dtype(np.atleast_1d(p.xyLS)[k]) ** 2
    Bm = dtype(p.yScale) * np.exp(-lux)
    dt = Tm[:, 0][:, None] - Tm[:, 0][None, :]
    E = np.exp(-(dt * dt) / dtype(p.tyLS) ** 2)
    return Bm, E


def _chol_lower(A):
    """Column Cholesky that works for any float dtype (np.linalg has no longdouble)."""
    if A.dtype == np.float64:
        return np.linalg.cholesky(A)
    n = A.shape[0]
    L = np.zeros_like(A)
    A = A.copy()
    for j in range(n):
        d = A[j, j]
        if not d > 0:
            raise np.linalg.LinAlgError(f"not positive definite at pivot {j + 1}")
        d = np.sqrt(d)
        L[j, j] = d
        if j + 1 < n:
            col = A[j + 1:, j] / d
            L[j + 1:, j] = col
            A[j + 1:, j + 1:] -= np.outer(col, col)
    return L


def _tri_solve_lower(L, B):
    if L.dtype == np.float64:
        return sla.solve_triangular(L, B, lower=True, check_finite=False)
    n = L.shape[0]
    Xs = np.array(B, dtype=L.dtype, copy=True)
    for i in range(n):
        if i:
            Xs[i] = Xs[i] - L[i, :i] @ Xs[:i]
        Xs[i] = Xs[i] / L[i, i]
    return Xs


def structured_ite(p: PosteriorSample, X, T, Y, doT, dtype=np.float64):
    """MeanITE / CovITE (no jitter) via K = B∘E, Ks = diag(r) B, Kss = B:
    D = Ks' - K = B∘(r_j - e_ij);  Delta = B∘(e_ij - r_i - r_j + 1);
    MeanITE = D A^-1 Y;  CovITE = Delta - (L^-1 D')' (L^-1 D')."""
    Bm, E = _base_and_e(p, X, T, dtype)
    Tv = _as_2d(T).astype(dtype)[:, 0]
    n = Tv.shape[0]
    r = np.exp(-((Tv - dtype(doT)) ** 2) / dtype(p.tyLS) ** 2)
    A = Bm * E + dtype(p.yNoise) * np.eye(n, dtype=dtype)
    L = _chol_lower(A)
    D = Bm * (r[None, :] - E)
    Delta = Bm * (E - r[:, None] - r[None, :] + dtype(1))
    z = _tri_solve_lower(L, np.asarray(Y, dtype=dtype))
    V = _tri_solve_lower(L, D.T.copy())
    mean = V.T @ z
    cov = Delta - V.T @ V
    return mean, cov


def structured_sate(p: PosteriorSample, X, T, Y, do_ts, pred_noise=PREDICTION_COVARIANCE_NOISE,
                    dtype=np.float64):
    """O(N^2)-per-level SATE mean/variance, exactly the quantities the HIP path forms:
    c = r∘bsum - ksum (column sums of D), sumDelta = sum K - 2 r.bsum + sum B,
    z = L^-1 Y, w = L^-1 c;  MeanSATE = w.z / n;  VarSATE = (sumDelta - w.w + n eps) / n^2.
    Also returns the :Y log-density pieces (logdet, quad)."""
    Bm, E = _base_and_e(p, X, T, dtype)
    Tv = _as_2d(T).astype(dtype)[:, 0]
    n = Tv.shape[0]
    K = Bm * E
    A = K + dtype(p.yNoise) * np.eye(n, dtype=dtype)
    L = _chol_lower(A)
    bsum = Bm.sum(axis=0)
    ksum = K.sum(axis=0)
    z = _tri_solve_lower(L, np.asarray(Y, dtype=dtype))
    ms, vs = [], []
    for doT in np.atleast_1d(do_ts):
        r = np.exp(-((Tv - dtype(doT)) ** 2) / dtype(p.tyLS) ** 2)
        c = r * bsum - ksum
        sum_delta = (ksum.sum() - dtype(2) * (r @ bsum)) + bsum.sum()
        w = _tri_solve_lower(L, c)
        ms.append((w @ z) / n)
        vs.append((sum_delta - w @ w + n * dtype(pred_noise)) / dtype(n) ** 2)
    logdet = 2 * np.sum(np.log(np.diag(L)))
    return np.array(ms, dtype=dtype), np.array(vs, dtype=dtype), logdet, z @ z


def literal_sate_longdouble(p: PosteriorSample, X, T, Y, doT,
                            pred_noise=PREDICTION_COVARIANCE_NOISE):
    """Extended-precision (x87 80-bit) evaluation of the literal 4-block formula with a
    Cholesky-based solve; small N only.  Used to set defensible tolerances."""
    ld = np.longdouble
    mean, cov = structured_ite(p, X, T, Y, doT, dtype=ld)
    n = mean.shape[0]
    cov = _symmetric_upper(cov) + np.eye(n, dtype=ld) * ld(pred_noise)
    return mean, cov, np.sum(mean) / n, np.sum(cov) / ld(n) ** 2


# ----------------------------------------------------------------------------
# counter-based normals shared with the HIP path (Philox4x32-10 + Box-Muller)
# ----------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al. 2011).  counter (..., 4) uint32, key (2,) uint32."""
    c = np.asarray(counter, dtype=np.uint64).copy()
    k0, k1 = int(key[0]), int(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PHILOX_M0 * c[..., 0]
        p1 = _PHILOX_M1 * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        n0 = hi1 ^ c[..., 1] ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c[..., 3] ^ np.uint64(k1)
        n3 = lo0
        c = np.stack([n0, n1, n2, n3], axis=-1)
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def philox_normals(seed: int, stream: int, count: int):
    """``count`` N(0,1) doubles for (seed, stream): element e uses counter
    (e>>1 low, e>>1 high, stream low, stream high), key = seed split in two words;
    the 4 output words make two 53-bit-ish uniforms (u = ((hi<<21 ^ lo>>11)+0.5)/2^53 for
    u1 from words 0,1 and u2 from words 2,3); Box-Muller: even e -> r cos, odd e -> r sin."""
    e = np.arange(count, dtype=np.uint64)
    pair = e >> np.uint64(1)
    ctr = np.stack([pair & np.uint64(0xFFFFFFFF), pair >> np.uint64(32),
                    np.full(count, stream & 0xFFFFFFFF, dtype=np.uint64),
                    np.full(count, (stream >> 32) & 0xFFFFFFFF, dtype=np.uint64)], axis=-1)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = philox4x32_10(ctr, key).astype(np.uint64)
    a = (w[:, 0] << np.uint64(21)) ^ (w[:, 1] >> np.uint64(11))
    b = (w[:, 2] << np.uint64(21)) ^ (w[:, 3] >> np.uint64(11))
    u1 = (a.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = (b.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * math.pi * u2
    return np.where((e & np.uint64(1)) == 0, rad * np.cos(ang), rad * np.sin(ang))


# ----------------------------------------------------------------------------
# the other Gaussian-process nodes of the Gen models
# ----------------------------------------------------------------------------


def mvnormal_logpdf(x, cov):
    """Gen ``mvnormal(zeros(n), cov)`` score = logpdf(MvNormal(0, Symmetric(cov)), x) (Distributions/PDMats:
    Cholesky, -1/2 (n log 2pi + logdet + x' cov^-1 x))."""
    x = np.asarray(x, dtype=np.float64)
    L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64))
    z = sla.solve_triangular(L, x, lower=True, check_finite=False)
    n = x.shape[0]
    return float(-0.5 * (n * math.log(2 * math.pi) + 2 * np.sum(np.log(np.diag(L))) + z @ z))


def x_node_logpdf(U, uxLS_k, xScale_k, xNoise_k, X_k):
    """:X => k => :X of generateXfromU (src/model_likelihood.jl:13-22):
    xCov_k = processCov(rbfKernelLog(U, U, uxLS[k, :]), xScale[k], xNoise[k])."""
    return mvnormal_logpdf(X_k, process_cov(rbf_kernel_log(U, U, uxLS_k), xScale_k, xNoise_k))


def t_node_logpdf(U, X, utLS, xtLS, tScale, tNoise, T):
    """:T (real) / :logitT (binary) of generate{Real,Binary}Tfrom{UX,U,X}
    (src/model_likelihood.jl:25-80): cov = processCov(utCovLog + xtCovLog, tScale, tNoise)."""
    n = np.asarray(T).shape[0]
    acc = np.zeros((n, n))
    if U is not None:
        acc = acc + rbf_kernel_log(U, U, utLS)
    if X is not None:
        acc = acc + rbf_kernel_log(X, X, xtLS)
    return mvnormal_logpdf(T, process_cov(acc, tScale, tNoise))


def u_node_logpdf(SigmaU, uNoise, U_k):
    """:U => u => :U of generateUfromSigmaU (src/model_likelihood.jl:4-10): uCov = SigmaU * uNoise."""
    return mvnormal_logpdf(U_k, np.asarray(SigmaU) * uNoise)
