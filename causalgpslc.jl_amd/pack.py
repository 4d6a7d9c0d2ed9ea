"""Posterior pack: a flat little-endian binary of the data and the retained posterior samples
(SURVEY.md §8f next-2).  It replaces Julia's `Serialization` of a GPSLCObject (src/io.jl:14-34) for the
prediction path: everything `extractParameters` (src/utils.jl:92-124) yields for
i in nBurnIn:stepSize:nOuter, stacked, in Julia's own column-major order — so the Julia-side writer is a
handful of `write(io, ...)` calls and the reader needs no Julia.

The reader and writer are the C functions of the library (gpslc_pack_save / gpslc_pack_read_header /
gpslc_pack_load, include/gpslc_hip.h) so that any host language binds the same code; this module only
allocates the NumPy buffers.

Layout (all little-endian):
    8 bytes   magic  b"GPSLCPK1"
    6 x int64 n, nX, nU, S, binaryT (0/1), reserved (0)
    7 x f64   hyperparams: nU (or -1), nOuter, nMHInner, nESInner, nBurnIn, stepSize, predictionCovarianceNoise
    f64 arrays, column-major, in this order:
        X[n, nX]  T[n]  Y[n]  U[n, nU, S]  uyLS[nU, S]  xyLS[nX, S]  tyLS[S]  yNoise[S]  yScale[S]
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .api import GPSLCObject, HyperParameters, _p

MAGIC = b"GPSLCPK1"


def _raise(st, what):
    if st == -1005:
        raise OSError(f"{what}: file cannot be opened, read or written")
    if st == -1006:
        raise ValueError(f"{what}: not a GPSLC posterior pack (bad magic, truncated or trailing bytes)")
    raise _lib.GPSLCError(st, what)


def saveGPSLCObject(g: GPSLCObject, path: str, binary_t: bool = False) -> None:
    """saveGPSLCObject(g, filename) (src/io.jl:14-19) in the flat pack format."""
    lib = _lib.load()
    hp = g.hyperparams
    h = _lib.PackHeader(g.getN(), g.getNX(), g.getNU(), g.getNumPosteriorSamples(), 1 if binary_t else 0, 0)
    for i, v in enumerate((-1.0 if hp.nU is None else float(hp.nU), hp.nOuter, hp.nMHInner, hp.nESInner, hp.nBurnIn,
                           hp.stepSize, hp.predictionCovarianceNoise)):
        h.hyper[i] = float(v)
    arrs = [None if a is None else np.asfortranarray(a, dtype=np.float64)
            for a in (g.X, g.T, g.Y, g.U, g.uyLS, g.xyLS, g.tyLS, g.yNoise, g.yScale)]
    st = lib.gpslc_pack_save(os.fsencode(path), C.byref(h), *[_p(a) for a in arrs])
    if st != 0:
        _raise(st, f"gpslc_pack_save({path})")


def readPackHeader(path: str) -> dict:
    """Sizes and hyper-parameters of a pack without reading its arrays."""
    lib = _lib.load()
    h = _lib.PackHeader()
    st = lib.gpslc_pack_read_header(os.fsencode(path), C.byref(h))
    if st != 0:
        _raise(st, f"gpslc_pack_read_header({path})")
    return {"n": h.n, "nX": h.nX, "nU": h.nU, "S": h.S, "binary_t": bool(h.binary_t), "hyper": list(h.hyper)}


def loadGPSLCObject(path: str, device: int = 0, samples: Optional[Tuple[int, int]] = None,
                    fp32_kernel: bool = False) -> GPSLCObject:
    """loadGPSLCObject(filename) (src/io.jl:29-34) from the flat pack format.  ``samples = (s0, s1)`` loads only
    that contiguous block of posterior samples — what one rank of a sharded prediction needs."""
    lib = _lib.load()
    hd = readPackHeader(path)
    n, nX, nU, S = hd["n"], hd["nX"], hd["nU"], hd["S"]
    s0, s1 = (0, S) if samples is None else (int(samples[0]), int(samples[1]))
    if not 0 <= s0 <= s1 <= S:
        raise IndexError(f"sample block [{s0}, {s1}) outside 0..{S}")
    Sl = s1 - s0

    def buf(*shape, samples=False):
        lead = shape[:-1] if samples else shape      # an empty sample block keeps its (zero-length) arrays
        return np.empty(shape, order="F") if all(d > 0 for d in lead) else None

    X, T, Y = buf(n, nX), buf(n), buf(n)
    U, uyLS, xyLS = buf(n, nU, Sl, samples=True), buf(nU, Sl, samples=True), buf(nX, Sl, samples=True)
    tyLS, yNoise, yScale = buf(Sl, samples=True), buf(Sl, samples=True), buf(Sl, samples=True)
    st = lib.gpslc_pack_load(os.fsencode(path), s0, s1, *[_p(a) for a in (X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale)])
    if st != 0:
        _raise(st, f"gpslc_pack_load({path})")
    hpv = hd["hyper"]
    hp = HyperParameters(None if hpv[0] < 0 else int(hpv[0]), int(hpv[1]), int(hpv[2]), int(hpv[3]), int(hpv[4]),
                         int(hpv[5]), hpv[6])
    g = GPSLCObject(X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale, hyperparams=hp, device=device,
                    fp32_kernel=fp32_kernel)
    g.binary_t = hd["binary_t"]
    return g
