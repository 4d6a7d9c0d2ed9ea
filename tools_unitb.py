import sys, time, numpy as np
sys.path.insert(0, '.')
import causalgpslc_jl_amd as gp
from causalgpslc_jl_amd import synth
n, D, K = 4096, 8, 2
X, T, Y, obj = synth.make_dataset(n, D)
for (S, L, spp) in [(64, 1, 8), (32, 2, 8)]:
    post = synth.make_posterior(n, D, K, S, obj)
    g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
    doTs = synth.levels(T, L)
    gp.predict(g, doTs, spp=spp, seed=1, want_draws=True)
    t0 = time.perf_counter()
    ms, vs, mi, dr = gp.predict(g, doTs, spp=spp, seed=1, want_draws=True)
    dt = time.perf_counter() - t0
    print(f"S={S} L={L} spp={spp}: {dt*1e3:.1f} ms -> {S*L/dt:.1f} unit-B/s ({S*L*160.4e9/dt/1e12:.1f} TF incl. unit A + copies); finite={np.isfinite(dr).all()}")
