#!/bin/bash
# round 6, call 44: MeanITE of a level sweep with 128 levels per pass (ite_mean_mfma_kernel<F, 128>) against passes of 64
# (measurement build, GPSLC_ITE_NL128 = 0): level-sweep tests first, identical MeanITE (SHA-256), then samples/s
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c44; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_estimation.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for w in 1 0; do
GPSLC_ITE_NL128=$w timeout -k 10 300 python3 - > $O/sha_$w.txt 2>&1 <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
import causalgpslc_jl_amd as gp
from causalgpslc_jl_amd import synth
gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")
for n, D, K, S, L in ((700, 8, 2, 5, 65), (1024, 4, 1, 3, 101), (1000, 11, 1, 2, 128), (2048, 8, 2, 2, 130), (300, 2, 0, 4, 200), (520, 14, 2, 2, 97)):
    X, T, Y, obj = synth.make_dataset(n, D)
    post = synth.make_posterior(n, D, K, S, obj, seed=99)
    g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
    ms, vs, mi = gp.predict(g, synth.levels(T, L), want_mean_ite=True)
    print(n, D, K, S, L, hashlib.sha256(np.ascontiguousarray(mi).tobytes()).hexdigest()[:16], float(np.abs(mi).max()))
PY
done
cmp $O/sha_1.txt $O/sha_0.txt && echo "MeanITE identical"; grep -v amdgpu $O/sha_1.txt
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for L in 65 101 128; do
for w in 1 0; do
GPSLC_ITE_NL128=$w timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 4096 --levels $L > $O/n1024_l${L}_$w.json 2> $O/err.txt; val $O/n1024_l${L}_$w.json "N=1024 L=$L passes of $(( 64 + 64 * w ))"
done; done
for w in 1 0; do
GPSLC_ITE_NL128=$w timeout -k 10 300 $B --levels 101 > $O/n4096_l101_$w.json 2> $O/err.txt; val $O/n4096_l101_$w.json "N=4096 L=101 passes of $(( 64 + 64 * w ))"
done
