#!/usr/bin/env python3
"""In-kernel phase stamps of the diagonal-block factorisation (measurement build): s_memtime of waves 0 and 1 of workgroup 0 at
the phase boundaries of every step of diag_potrf_inv_la_kernel, for a launch of `batch` matrices at size N (the first column of a
factorisation: gpslc_y_logpdf on N = 1024).  Usage: potrf_stamps.py [N batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causalgpslc_jl_amd as gp          # noqa: E402
from causalgpslc_jl_amd import synth    # noqa: E402

gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")
n, S = (int(a) for a in (sys.argv[1:3] + ["1024", "8192"][len(sys.argv) - 1:]))
X, T, Y, obj = synth.make_dataset(n, 4)
post = synth.make_posterior(n, 4, 1, S, obj)
g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
gp.predict(g, [0.3])
gp.predict(g, [0.3])
lib = g.ctx().lib
buf = (C.c_ulonglong * 80)()
lib.gpslc_diag_potrf_stamps.argtypes = [C.c_void_p]
assert lib.gpslc_diag_potrf_stamps(C.cast(buf, C.c_void_p)) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(40, 2).astype(np.int64)
t0 = st[0].min()
names = ["image loaded", "barrier", "block 0 factorised (wave 0)", "barrier"]
for p in range(8):
    names += [f"step {p} phase 1 done", "barrier A", f"step {p} phase 2 done", "barrier B"]
print(f"N={n}, {S} matrices per launch; shader clocks (s_memtime) since the first stamp: wave 0 / wave 1 of workgroup 0 (the first workgroup of its CU: its first block and first step also pay the instruction-cache misses)")
for i, nm in enumerate(names[:36]):
    print(f"{nm:34s} {st[i, 0] - t0:9d} {st[i, 1] - t0:9d}")
print("total", (st[35].max() - t0))
