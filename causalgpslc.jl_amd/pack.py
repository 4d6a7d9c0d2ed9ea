"""Posterior pack: a flat little-endian binary of the data and the retained posterior samples
(SURVEY.md §8f next-2).  It replaces Julia's `Serialization` of a GPSLCObject (src/io.jl:14-34) for the
prediction path: everything `extractParameters` (src/utils.jl:92-124) yields for
i in nBurnIn:stepSize:nOuter, stacked, in Julia's own column-major order — so the Julia-side writer is a
handful of `write(io, ...)` calls and the reader needs no Julia.

Layout (all little-endian):
    8 bytes   magic  b"GPSLCPK1"
    6 x int64 n, nX, nU, S, binaryT (0/1), reserved (0)
    7 x f64   hyperparams: nU (or -1), nOuter, nMHInner, nESInner, nBurnIn, stepSize, predictionCovarianceNoise
    f64 arrays, column-major, in this order:
        X[n, nX]  T[n]  Y[n]  U[n, nU, S]  uyLS[nU, S]  xyLS[nX, S]  tyLS[S]  yNoise[S]  yScale[S]
"""
from __future__ import annotations

import struct

import numpy as np

from .api import GPSLCObject, HyperParameters

MAGIC = b"GPSLCPK1"


def saveGPSLCObject(g: GPSLCObject, path: str, binary_t: bool = False) -> None:
    """saveGPSLCObject(g, filename) (src/io.jl:14-19) in the flat pack format."""
    n, nX, nU, S = g.getN(), g.getNX(), g.getNU(), g.getNumPosteriorSamples()
    hp = g.hyperparams
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<6q", n, nX, nU, S, 1 if binary_t else 0, 0))
        f.write(struct.pack("<7d", -1.0 if hp.nU is None else float(hp.nU), hp.nOuter, hp.nMHInner, hp.nESInner,
                            hp.nBurnIn, hp.stepSize, hp.predictionCovarianceNoise))
        for a in (g.X, g.T, g.Y, g.U, g.uyLS, g.xyLS, g.tyLS, g.yNoise, g.yScale):
            if a is not None:
                f.write(np.asfortranarray(a, dtype="<f8").tobytes(order="F"))


def loadGPSLCObject(path: str, device: int = 0) -> GPSLCObject:
    """loadGPSLCObject(filename) (src/io.jl:29-34) from the flat pack format."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not a GPSLC posterior pack")
        n, nX, nU, S, binary_t, _ = struct.unpack("<6q", f.read(48))
        hpv = struct.unpack("<7d", f.read(56))

        def rd(*shape):
            cnt = int(np.prod(shape))
            if cnt == 0:
                return None
            buf = f.read(8 * cnt)
            if len(buf) != 8 * cnt:
                raise ValueError("truncated posterior pack")
            return np.frombuffer(buf, dtype="<f8").reshape(shape, order="F").copy(order="F")

        X = rd(n, nX)
        T, Y = rd(n), rd(n)
        U, uyLS, xyLS = rd(n, nU, S), rd(nU, S), rd(nX, S)
        tyLS, yNoise, yScale = rd(S), rd(S), rd(S)
        if f.read(1):
            raise ValueError("trailing bytes in posterior pack")
    hp = HyperParameters(None if hpv[0] < 0 else int(hpv[0]), int(hpv[1]), int(hpv[2]), int(hpv[3]), int(hpv[4]),
                         int(hpv[5]), hpv[6])
    g = GPSLCObject(X, T, Y, U, uyLS, xyLS, tyLS, yNoise, yScale, hyperparams=hp, device=device)
    g.binary_t = bool(binary_t)
    return g
