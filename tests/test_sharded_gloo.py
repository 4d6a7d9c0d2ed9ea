"""The N > 1 path on CPU: world_size-2 gloo process group, posterior samples sharded over ranks, one
all_gather at the end.  The per-rank compute is injected (the oracle stands in for the HIP entry point —
there is no GPU here); what is under test is the partition, the padding and the gather."""
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
