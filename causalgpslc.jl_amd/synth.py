"""Deterministic synthetic workloads: data set + posterior pack.

X ~ N(0,1) (n x D); objects of 16 instances; continuous T ~ N(0,1) or Bernoulli(0.5);
Y = sin(T) + 0.5 X[:,0] + u_obj + 0.3 eps.  Posterior pack per sample: U[:, k, s] = object-level
N(0,1) value broadcast to its instances + 1e-6 eps (what elliptical slice sampling under SigmaU
produces, src/utils.jl:17-33); uyLS, xyLS, tyLS, yScale, yNoise ~ InvGamma(4, 4) i.i.d. (the
reference's priors, src/hyperparameters.jl:38-70) floored at 0.25 to keep cond(A) bounded.
Generator: numpy Philox, seed 1234 (the reference's test seed, test/runtests.jl:18).
"""
from __future__ import annotations

import numpy as np


def _invgamma(rng, shape, scale, size):
    return np.maximum(scale / rng.gamma(shape, 1.0, size=size), 0.25)


def make_dataset(n, D, binary_t=False, seed=1234, obj_size=16):
    rng = np.random.Generator(np.random.Philox(seed))
    X = rng.standard_normal((n, D)) if D > 0 else None
    nobj = (n + obj_size - 1) // obj_size
    obj = np.repeat(np.arange(nobj), obj_size)[:n]
    u_obj = rng.standard_normal(nobj)[obj]
    if binary_t:
        T = (rng.random(n) < 0.5).astype(np.float64)
    else:
        T = rng.standard_normal(n)
    Y = np.sin(T) + (0.5 * X[:, 0] if D > 0 else 0.0) + u_obj + 0.3 * rng.standard_normal(n)
    return X, T, Y, obj


def make_posterior(n, D, K, S, obj, seed=1234):
    """Returns dict(U (n,K,S) | None, uyLS (K,S) | None, xyLS (D,S) | None, tyLS, yNoise, yScale (S,))."""
    rng = np.random.Generator(np.random.Philox(seed + 1))
    out = {}
    if K > 0:
        nobj = int(obj.max()) + 1
        base = rng.standard_normal((nobj, K, S))
        U = base[obj] + 1e-6 * rng.standard_normal((n, K, S))
        out["U"] = np.asfortranarray(U)
        out["uyLS"] = np.asfortranarray(_invgamma(rng, 4.0, 4.0, (K, S)))
    else:
        out["U"] = None
        out["uyLS"] = None
    out["xyLS"] = np.asfortranarray(_invgamma(rng, 4.0, 4.0, (D, S))) if D > 0 else None
    out["tyLS"] = _invgamma(rng, 4.0, 4.0, S)
    out["yNoise"] = _invgamma(rng, 4.0, 4.0, S)
    out["yScale"] = _invgamma(rng, 4.0, 4.0, S)
    return out


def levels(T, L):
    """L equally spaced intervention levels over [min T, max T] (src/prediction.jl:24-28); L = 1 -> the
    midpoint; binary T with L = 2 -> {0, 1}."""
    lo, hi = float(np.min(T)), float(np.max(T))
    if L == 1:
        return np.array([0.5 * (lo + hi)])
    return lo + (hi - lo) / (L - 1) * np.arange(L)
