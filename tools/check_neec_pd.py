import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import causalgpslc_jl_amd as gp
import gpslc_oracle as orc
neec = os.path.join(ROOT, "tests", "golden", "neec", "NEEC_sampled.csv")
hp = gp.getHyperParameters()
hp.nOuter, hp.nU, hp.nMHInner, hp.nESInner = 100, 2, 3, 5
g = gp.gpslc(neec, hyperparams=hp, seed=1234)
S = gp.getNumPosteriorSamples(g)
print("S", S, "yNoise min", g.yNoise.min(), "yScale max", g.yScale.max(), "tyLS range", g.tyLS.min(), g.tyLS.max(), "uyLS min", g.uyLS.min())
rng = gp.doTRange(float(g.T.min()), float(g.T.max()), 100)
try:
    gp.predict(g, rng, spp=1, seed=1, want_draws=True)
    print("no failure")
except gp.PosDefException as e:
    info = g.ctx().last_info(S)
    bad = np.nonzero(info)[0]
    print("failing samples", bad, info[bad])
    s = int(bad[0])
    p = orc.PosteriorSample(g.uyLS[:, s], None, float(g.tyLS[s]), float(g.yNoise[s]), float(g.yScale[s]), g.U[:, :, s])
    print("params", p.uyLS, p.tyLS, p.yNoise, p.yScale)
    nfail = 0
    mins = []
    for l, d in enumerate(rng):
        M, Cv = orc.ite_distributions([p], None, g.T, g.Y, float(d))
        ev = np.linalg.eigvalsh(Cv[0])
        mins.append(ev[0])
        try:
            np.linalg.cholesky(Cv[0])
        except np.linalg.LinAlgError:
            nfail += 1
    print("oracle (literal restatement) Cholesky failures over the 101 levels for that sample:", nfail, "min eig range", min(mins), max(mins))
    # which levels fail on the GPU for this sample
    from causalgpslc_jl_amd.sharded import slice_object
    g1 = slice_object(g, s, s + 1)
    gf = 0
    for l, d in enumerate(rng):
        try:
            gp.predict(g1, [d], spp=1, seed=1, want_draws=True)
        except gp.PosDefException:
            gf += 1
    print("GPU failures over the 101 levels for that sample:", gf)
    # which level, and how far is the GPU CovITE from the literal one
    for l, d in enumerate(rng):
        try:
            gp.predict(g1, [d], spp=1, seed=1, want_draws=True)
        except gp.PosDefException as e2:
            print("level", l, "doT", d, "info", e2.info)
            Mg, Cg = gp.ITEDistributions(g1, float(d))
            M, Cv = orc.ite_distributions([p], None, g.T, g.Y, float(d))
            Cg0 = Cg[0]
            print("max |C_gpu - C_lit|", np.max(np.abs(Cg0 - Cv[0])), "asym", np.max(np.abs(Cg0 - Cg0.T)), "max|C|", np.max(np.abs(Cv[0])))
            print("eig min gpu", np.linalg.eigvalsh((Cg0 + Cg0.T) / 2)[:3], "lit", np.linalg.eigvalsh(Cv[0])[:3])
            ms, Cs = orc.structured_ite(p, None, g.T, g.Y, float(d))
            Cs = Cs + 1e-10 * np.eye(len(ms))
            print("structured numpy: max |C_str - C_lit|", np.max(np.abs(Cs - Cv[0])), "eig min", np.linalg.eigvalsh((Cs + Cs.T) / 2)[:3])
            try:
                np.linalg.cholesky(Cg0); print("numpy cholesky of the GPU CovITE: ok")
            except np.linalg.LinAlgError:
                print("numpy cholesky of the GPU CovITE: FAILS too")
            # hypothesis: the breakdown comes from the inverse-based panel solves of the tiled factorisation
            x = np.random.default_rng(0).standard_normal(150)
            try:
                v = gp.mvnLogpdf(Cg0, x)
                print("single-workgroup (column-operation) factorisation of the GPU CovITE: ok, logpdf", v, "ref", orc.mvnormal_logpdf(x, Cg0))
            except gp.PosDefException as e3:
                print("single-workgroup factorisation FAILS: info", e3.info)
            big = np.eye(700)
            big[:150, :150] = Cg0
            try:
                v = gp.mvnLogpdf(big, np.concatenate([x, np.zeros(550)]))
                print("tiled factorisation (n = 700 embedding): ok", v)
            except gp.PosDefException as e3:
                print("tiled factorisation (n = 700 embedding) FAILS: info", e3.info)
            break
