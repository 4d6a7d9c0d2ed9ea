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
