#!/bin/bash
# round 6, call 39: W = D L^-T of unit B as one persistent launch of row tasks (wsolve_rows_kernel) against one launch per tile
# column (measurement build, GPSLC_W_ROWS = 0): identical draw tensors first (SHA-256), then units/s at N = 4096 / 2048 / 1024
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c39; mkdir -p $O
for w in 1 0; do
GPSLC_W_ROWS=$w timeout -k 10 300 python3 - > $O/sha_$w.txt 2>&1 <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
import causalgpslc_jl_amd as gp
from causalgpslc_jl_amd import synth
gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")
for n, S, L in ((700, 5, 3), (1024, 3, 2), (2048, 2, 2), (4096, 2, 1)):
    X, T, Y, obj = synth.make_dataset(n, 8)
    post = synth.make_posterior(n, 8, 2, S, obj, seed=99)
    g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
    ms, vs, mi, dr = gp.predict(g, synth.levels(T, L), spp=4, seed=3, want_draws=True)
    print(n, S, L, hashlib.sha256(np.ascontiguousarray(dr).tobytes()).hexdigest()[:16], hashlib.sha256(np.ascontiguousarray(mi).tobytes()).hexdigest()[:16])
PY
done
cmp $O/sha_1.txt $O/sha_0.txt && echo "draw tensors identical" ; cat $O/sha_1.txt | grep -v amdgpu
for rep in 1 2; do
for w in 1 0; do
GPSLC_W_ROWS=$w timeout -k 10 300 python3 tools/bench_unit_b.py --diag-lib 4096 8 16 10 2> $O/err.txt | tail -1 | sed "s/^/rows launch=$w: /"
done; done
for w in 1 0; do
GPSLC_W_ROWS=$w timeout -k 10 300 python3 tools/bench_unit_b.py --diag-lib 4096 64 1 10 2> $O/err.txt | tail -1 | sed "s/^/rows launch=$w: /"
GPSLC_W_ROWS=$w timeout -k 10 300 python3 tools/bench_unit_b.py --diag-lib 2048 16 16 10 2> $O/err.txt | tail -1 | sed "s/^/rows launch=$w: /"
GPSLC_W_ROWS=$w timeout -k 10 300 python3 tools/bench_unit_b.py --diag-lib 1024 32 32 10 2> $O/err.txt | tail -1 | sed "s/^/rows launch=$w: /"
done
