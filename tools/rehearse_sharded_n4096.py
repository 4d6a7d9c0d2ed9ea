#!/usr/bin/env python3
"""BASELINE config 3 shape through the sharded path on ONE GPU: N = 4096, D = 8, nU = 2, S posterior samples x 64
intervention levels, two gloo ranks sharing device 0 (RCCL refuses two ranks per device), every rank loading only its
block of the posterior pack; the gathered SATE arrays are compared bit for bit with the single-process prediction."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch.distributed as dist
    import causalgpslc_jl_amd as gp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hd = gp.readPackHeader(path)
    T = gp.loadGPSLCObject(path, samples=(0, 0)).T
    doTs = gp.synth.levels(T, 64)
    t0 = time.perf_counter()
    ms, vs = gp.predict_sharded_pack(path, doTs)
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"2 ranks on one GPU (gloo): {hd['S']} samples x 64 levels in {dt:.2f} s", flush=True)
    np.savez(os.path.join(out, f"r{rank}.npz"), ms=ms, vs=vs)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import causalgpslc_jl_amd as gp
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    n, D, K = 4096, 8, 2
    X, T, Y, obj = gp.synth.make_dataset(n, D)
    post = gp.synth.make_posterior(n, D, K, S, obj)
    g = gp.GPSLCObject(X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "g.pk")
        gp.saveGPSLCObject(g, path)
        mp.spawn(worker, args=(2, 29650, path, d), nprocs=2, join=True)
        t0 = time.perf_counter()
        ms, vs, _ = gp.predict(g, gp.synth.levels(T, 64))
        print(f"single process: {time.perf_counter() - t0:.2f} s", flush=True)
        for r in range(2):
            z = np.load(os.path.join(d, f"r{r}.npz"))
            assert np.array_equal(z["ms"], ms) and np.array_equal(z["vs"], vs), r
    print("sharded == single process, bit for bit")
