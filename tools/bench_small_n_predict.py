#!/usr/bin/env python3
"""Where the time of predictCounterfactualEffects goes at the reference's own size (n = 150, 91 posterior samples x 101
levels x 100 draws: a 1.1 GB draw tensor): the same call with the outputs left in HBM (gpslc_predict_dev) against the
host-pointer form that also brings the tensor to pageable host memory."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402

n, S, L, spp, nU = 150, 91, 101, 100, 2
rng = np.random.default_rng(0)
X, T, Y, objid = gp.synth.make_dataset(n, 0, binary_t=False, seed=3)
post = gp.synth.make_posterior(n, 0, nU, S, objid, seed=3)
g = gp.GPSLCObject(None, T, Y, post["U"], post["uyLS"], None, post["tyLS"], post["yNoise"], post["yScale"])
doTs = np.linspace(T.min(), T.max(), L)
for _ in range(2):
    t0 = time.perf_counter()
    out = gp.predict(g, doTs, spp=spp, seed=7, want_draws=True)
    t1 = time.perf_counter()
print(f"host-pointer form (draw tensor to pageable host memory): {t1 - t0:.3f} s")

dev = torch.device("cuda", 0)
ctx = g.ctx()
to_dev = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1, order="F"))).to(dev)
ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
packs = [to_dev(a) for a in (g.U, g.uyLS, g.xyLS, g.tyLS, g.yScale, g.yNoise)]
ddo = to_dev(doTs)
ms = torch.empty(S * L, dtype=torch.float64, device=dev)
vs = torch.empty(S * L, dtype=torch.float64, device=dev)
dr = torch.empty(L * n * S * spp, dtype=torch.float64, device=dev)
for _ in range(3):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    st = ctx.lib.gpslc_predict_dev(ctx.h, S, *[ptr(t) for t in packs], L, ptr(ddo), 1e-10, spp, 7, None, ptr(ms), ptr(vs), None, ptr(dr))
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    assert st == 0, st
print(f"device form (everything stays in HBM): {t1 - t0:.3f} s = {S * L / (t1 - t0):.0f} (sample, level) units/s")
pin = torch.empty(L * n * S * spp, dtype=torch.float64).pin_memory()
torch.cuda.synchronize(dev)
t0 = time.perf_counter(); pin.copy_(dr); torch.cuda.synchronize(dev); t1 = time.perf_counter()
print(f"1.1 GB device -> pinned host: {t1 - t0:.3f} s ({dr.numel() * 8 / (t1 - t0) / 1e9:.1f} GB/s)")

# the host-pointer form writing into a PINNED destination (what a caller with a registered buffer gets)
out = pin.numpy()
U, uy, xy, ty, ys, yn = g._params()
msn, vsn = np.empty(S * L), np.empty(S * L)
for _ in range(2):
    t0 = time.perf_counter()
    st = ctx.lib.gpslc_predict(ctx.h, S, U, uy, xy, ty, ys, yn, L, doTs.ctypes.data_as(C.c_void_p), 1e-10, spp, 7, None,
                               msn.ctypes.data_as(C.c_void_p), vsn.ctypes.data_as(C.c_void_p), None, out.ctypes.data_as(C.c_void_p))
    t1 = time.perf_counter()
    assert st == 0, st
print(f"host-pointer form into a pinned destination: {t1 - t0:.3f} s")
