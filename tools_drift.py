import sys, numpy as np
sys.path.insert(0, '.')
import causalgpslc_jl_amd as gp
from causalgpslc_jl_amd import synth
for (n, D, K, binary) in [(4096, 8, 2, False), (16384, 16, 4, True)]:
    S = 8 if n == 4096 else 2
    X, T, Y, obj = synth.make_dataset(n, D, binary_t=binary)
    post = synth.make_posterior(n, D, K, S, obj)
    doTs = np.array([0.0, 1.0]) if binary else synth.levels(T, 2)
    out = {}
    for f32 in (False, True):
        g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"], fp32_kernel=f32)
        out[f32] = gp.predict(g, doTs, want_mean_ite=True)
    m64, v64, i64 = out[False]; m32, v32, i32 = out[True]
    print(f"N={n} D={D} nU={K} binary={binary}: max rel drift MeanSATE {np.max(np.abs(m32-m64)/np.abs(m64)):.2e}, VarSATE {np.max(np.abs(v32-v64)/np.abs(v64)):.2e}, MeanITE {np.max(np.abs(i32-i64))/np.max(np.abs(i64)):.2e}")
