"""The N > 1 path: world_size-2 gloo process group, posterior samples sharded over ranks, one all_gather at
the end.  On CPU (-m "not gpu") the per-rank compute is injected (the oracle stands in for the HIP entry point —
there is no GPU here); what is under test is the partition, the padding, the gather, and each rank loading only
its own block of the posterior pack.  The -m gpu test runs the real HIP entry point under the same process group
(two ranks sharing device 0; RCCL refuses two ranks on one device, so the group stays gloo)."""
import os
import sys

import numpy as np
import pytest

import cases
import gpslc_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_compute(g_local, doTs):
    S = g_local.getNumPosteriorSamples()
    ms = np.zeros((S, len(doTs)))
    vs = np.zeros((S, len(doTs)))
    for s in range(S):
        p = orc.PosteriorSample(None if g_local.uyLS is None else g_local.uyLS[:, s],
                                None if g_local.xyLS is None else g_local.xyLS[:, s],
                                float(g_local.tyLS[s]), float(g_local.yNoise[s]), float(g_local.yScale[s]),
                                None if g_local.U is None else g_local.U[:, :, s])
        m, v, _, _ = orc.structured_sate(p, g_local.X, g_local.T, g_local.Y, doTs)
        ms[s], vs[s] = m, v
    return ms, vs


def _oracle_compute_with_mean(g_local, doTs):
    ms, vs = _oracle_compute(g_local, doTs)
    S, n = g_local.getNumPosteriorSamples(), g_local.getN()
    mi = np.zeros((n, S, len(doTs)))
    for s in range(S):
        p = orc.PosteriorSample(None if g_local.uyLS is None else g_local.uyLS[:, s],
                                None if g_local.xyLS is None else g_local.xyLS[:, s],
                                float(g_local.tyLS[s]), float(g_local.yNoise[s]), float(g_local.yScale[s]),
                                None if g_local.U is None else g_local.U[:, :, s])
        for l, d in enumerate(doTs):
            mi[:, s, l] = orc.conditional_ite(p.uyLS, p.xyLS, p.tyLS, p.yNoise, p.yScale, p.U, g_local.X, g_local.T,
                                              g_local.Y, d)[0]
    return ms, vs, mi


def _worker_pack(rank, world, port, S, out_dir):
    """Every rank loads ONLY its block of the pack file and gathers SATE + MeanITE."""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s0, s1 = gp.shard_range(S, world, rank)
    g_local = gp.loadGPSLCObject(os.path.join(out_dir, "g.pk"), samples=(s0, s1))
    assert gp.getNumPosteriorSamples(g_local) == s1 - s0
    ms, vs, mi = gp.predict_sharded(g_local, np.array([0.3, 0.6]), compute=_oracle_compute_with_mean,
                                    gather_mean_ite=True, samples=(S, s0, s1))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ms=ms, vs=vs, mi=mi)
    dist.barrier()
    dist.destroy_process_group()


def _worker_hip(rank, world, port, S, out_dir):
    """The real HIP entry point under a process group: both ranks on device 0, gloo collectives."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ms, vs, mi = gp.predict_sharded_pack(os.path.join(out_dir, "g.pk"), np.array([0.3, 0.6]), gather_mean_ite=True,
                                         fp32_kernel=(S == 3))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ms=ms, vs=vs, mi=mi)
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, S, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.make_case(20, "UX", False, S=S, seed=3)
    g = cases.gpslc_object(gp, c)
    ms, vs = gp.predict_sharded(g, c["doTs"], compute=_oracle_compute)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ms=ms, vs=vs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [5, 2, 1])   # uneven shards, one sample each, fewer samples than ranks
def test_world_size_2_gloo_matches_single_process(tmp_path, S):
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    port = 29500 + (os.getpid() + S) % 2000
    mp.spawn(_worker, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    c = cases.make_case(20, "UX", False, S=S, seed=3)
    g = cases.gpslc_object(gp, c)
    ref_m, ref_v = _oracle_compute(g, c["doTs"])
    for r in range(2):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert np.array_equal(d["ms"], ref_m) and np.array_equal(d["vs"], ref_v)


@pytest.mark.parametrize("S", [5, 1])
def test_world_size_2_gloo_pack_blocks_and_mean_ite_gather(tmp_path, S):
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    c = cases.make_case(20, "UX", False, S=S, seed=4)
    g = cases.gpslc_object(gp, c)
    gp.saveGPSLCObject(g, str(tmp_path / "g.pk"))
    port = 31500 + (os.getpid() + S) % 2000
    mp.spawn(_worker_pack, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    ref = _oracle_compute_with_mean(g, np.array([0.3, 0.6]))
    for r in range(2):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for k, x in zip(("ms", "vs", "mi"), ref):
            assert np.array_equal(d[k], x), k


@pytest.mark.gpu
@pytest.mark.parametrize("S", [5, 3, 1])     # S = 3 runs the mixed-precision flag through the sharded path
def test_world_size_2_real_hip_path_on_one_device(tmp_path, S):
    """predict_sharded_pack with the default (HIP) compute under a 2-rank group: each rank loads its block of the
    pack, factorises on the device of its LOCAL_RANK, keeps the block in HBM and gathers.  Result == the
    single-process HIP prediction, bit for bit (same kernels, same per-sample arithmetic)."""
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    c = cases.make_case(150, "UX", False, S=S, seed=6)
    g = cases.gpslc_object(gp, c, fp32_kernel=(S == 3))
    gp.saveGPSLCObject(g, str(tmp_path / "g.pk"))
    port = 33500 + (os.getpid() + S) % 2000
    mp.spawn(_worker_hip, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    ms, vs, mi = gp.predict(g, np.array([0.3, 0.6]), want_mean_ite=True)
    for r in range(2):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert np.array_equal(d["ms"], ms) and np.array_equal(d["vs"], vs) and np.array_equal(d["mi"], mi)
    # and the library really used the flag: fp32 kernel build differs from the fp64 one in the last digits
    if S == 3:
        g64 = cases.gpslc_object(gp, c)
        ms64, _, _ = gp.predict(g64, np.array([0.3, 0.6]))
        assert not np.array_equal(ms64, ms) and np.allclose(ms64, ms, rtol=1e-5)


# ---- gather policies for the large outputs ("gather to rank 0 (or all-gather) ... optionally meanITE and draws")

def _oracle_compute_with_draws(g_local, doTs, spp, z_local):
    """CPU stand-in with draws: (ms, vs, mi, draws (L, n, S_r*spp)) — the reference's sampleITE per level
    (src/estimation.jl:95-109: mean + chol(CovITE + jitter) z, sample outer / draw inner)."""
    ms, vs, mi = _oracle_compute_with_mean(g_local, doTs)
    S, n, L = g_local.getNumPosteriorSamples(), g_local.getN(), len(doTs)
    dr = np.zeros((L, n, S * spp))
    for s in range(S):
        p = orc.PosteriorSample(None if g_local.uyLS is None else g_local.uyLS[:, s],
                                None if g_local.xyLS is None else g_local.xyLS[:, s],
                                float(g_local.tyLS[s]), float(g_local.yNoise[s]), float(g_local.yScale[s]),
                                None if g_local.U is None else g_local.U[:, :, s])
        for l, d in enumerate(doTs):
            M, Cv = orc.ite_distributions([p], g_local.X, g_local.T, g_local.Y, float(d), pred_noise=1e-6)
            Lc = np.linalg.cholesky(Cv[0])
            for k in range(spp):
                dr[l, :, s * spp + k] = M[0] + Lc @ z_local[:, k, s, l]
    return ms, vs, mi, dr


def _worker_modes(rank, world, port, S, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = cases.make_case(12, "UX", False, S=S, seed=8)
    g = cases.gpslc_object(gp, c)
    doTs, spp = np.array([0.2, 0.7]), 3
    z = np.random.default_rng(0).standard_normal((12, spp, S, 2))
    out = {}
    r = gp.predict_sharded_full(g, doTs, compute=_oracle_compute_with_draws, sate="root", mean_ite="root",
                                draws="root", spp=spp, z=z, root=1)
    out["root_has"] = np.array([r.meanSATE is not None, r.meanITE is not None, r.draws is not None])
    if rank == 1:
        out.update(ms=r.meanSATE, vs=r.varSATE, mi=r.meanITE.numpy(), dr=r.draws.numpy())
    r = gp.predict_sharded_full(g, doTs, compute=_oracle_compute_with_draws, sate="all", mean_ite="local",
                                draws="local", spp=spp, z=z)
    out.update(l_ms=r.meanSATE, l_mi=r.meanITE.numpy(), l_dr=r.draws.numpy(), l_rng=np.array([r.s0, r.s1]))
    r = gp.predict_sharded_full(g, doTs, compute=_oracle_compute_with_draws, sate="none", mean_ite="none",
                                draws="all", spp=spp, z=z, to_host=True)
    out.update(a_dr=r.draws, n_ms=r.meanSATE, none_mi=np.array([r.meanITE is None]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [5, 1])
def test_world_size_2_gloo_gather_policies_root_local_all(tmp_path, S):
    """MeanITE and draws: "root" delivers the whole tensor to the root rank ONLY (one gather), "local" leaves each rank
    its own block, "all" is the explicit all_gather; SATE "root" / "none" likewise.  Uneven and empty shards."""
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    port = 35500 + (os.getpid() + S) % 2000
    mp.spawn(_worker_modes, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    c = cases.make_case(12, "UX", False, S=S, seed=8)
    g = cases.gpslc_object(gp, c)
    doTs, spp = np.array([0.2, 0.7]), 3
    z = np.random.default_rng(0).standard_normal((12, spp, S, 2))
    ms, vs, mi, dr = _oracle_compute_with_draws(g, doTs, spp, z)
    d0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    d1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    assert not d0["root_has"].any() and d1["root_has"].all()          # root = 1: rank 0 received nothing
    for k, x in zip(("ms", "vs", "mi", "dr"), (ms, vs, mi, dr)):
        assert np.array_equal(d1[k], x), k
    for d in (d0, d1):
        a, b = d["l_rng"]
        assert np.array_equal(d["l_ms"], ms)                           # SATE "all": every rank
        assert np.array_equal(d["l_mi"], mi[:, a:b, :]) and np.array_equal(d["l_dr"], dr[:, :, a * spp:b * spp])
        assert np.array_equal(d["a_dr"], dr) and d["none_mi"][0]
        assert np.array_equal(d["n_ms"], ms[a:b])                      # SATE "none": the rank's own block


def _worker_hip_modes(rank, world, port, S, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.random.default_rng(1).standard_normal((150, 2, S, 2))
    r = gp.predict_sharded_pack(os.path.join(out_dir, "g.pk"), np.array([0.3, 0.6]), mean_ite="root", draws="root",
                                spp=2, z=z, to_host=True)
    out = {"has": np.array([r.meanITE is not None, r.draws is not None]), "ms": r.meanSATE, "vs": r.varSATE}
    # the library's own normals (Philox, seed 11): every rank places its block in the ensemble (gpslc_set_ensemble)
    rs = gp.predict_sharded_pack(os.path.join(out_dir, "g.pk"), np.array([0.3, 0.6]), draws="root", spp=2, seed=11,
                                 to_host=True)
    if rank == 0:
        out.update(mi=r.meanITE, dr=r.draws, dr_seeded=rs.draws)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("S", [5, 1])
def test_world_size_2_real_hip_path_root_gather_of_mean_ite_and_draws(tmp_path, S):
    """The real HIP entry point under a 2-rank group (both on device 0): MeanITE and the draw tensor are gathered to
    rank 0 only and equal the single-process prediction with the same normals bit for bit."""
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    c = cases.make_case(150, "UX", False, S=S, seed=6)
    g = cases.gpslc_object(gp, c)
    gp.saveGPSLCObject(g, str(tmp_path / "g.pk"))
    port = 37500 + (os.getpid() + S) % 2000
    mp.spawn(_worker_hip_modes, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    z = np.random.default_rng(1).standard_normal((150, 2, S, 2))
    ms, vs, mi, dr = gp.predict(g, np.array([0.3, 0.6]), want_mean_ite=True, spp=2, z=z, want_draws=True)
    d0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    d1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    assert d0["has"].all() and not d1["has"].any()
    assert np.array_equal(d0["mi"], mi) and np.array_equal(d0["dr"], dr)
    for d in (d0, d1):
        assert np.array_equal(d["ms"], ms) and np.array_equal(d["vs"], vs)
    # seeded draws do not depend on the sharding: two ranks = one process, bit for bit (seed + rank did not)
    dr_seeded = gp.predict(g, np.array([0.3, 0.6]), spp=2, seed=11, want_draws=True)[3]
    assert np.array_equal(d0["dr_seeded"], dr_seeded)


def test_shard_range_partitions_exactly():
    import causalgpslc_jl_amd as gp
    for S in (0, 1, 7, 8, 8192):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, b = gp.shard_range(S, world, r)
                assert 0 <= a <= b <= S
                cover += list(range(a, b))
            assert cover == list(range(S))
            sizes = [gp.shard_range(S, world, r)[1] - gp.shard_range(S, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
