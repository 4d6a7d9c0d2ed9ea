#!/usr/bin/env python3
"""bench.py — posterior samples/s of the GP hot path (Gram build + Cholesky + predict) at N = 4096.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" processes one batch of `--samples-per-step` posterior samples per GPU of the BASELINE
config "Synthetic N=4096 D=8 nU=2" (SURVEY.md §8d unit A: Gram build + potrf(A) + alpha +
MeanITE + SATE mean/variance for L = 1 intervention level).  Inputs are resident in HBM before the
timed region (gpslc_predict_dev takes device pointers; torch only provides device memory and the
process group).  Posterior samples shard over ranks with no data-path collective; one all_gather of
the (S x L) SATE arrays closes each step (weak scaling: per-GPU work is fixed).

The JSON line carries
  roofline      the dominant kernel (tile_gemm_nt_kernel<1, 0>, the f64-MFMA tile update): algorithmic
                flop (textbook count: diagonal tiles half, augmented rows by their live rows) / HIP-event
                time of every launch inside the timed region, against the fp64 matrix peak (78.6 TFLOP/s,
                AMD spec; the guides list no f64 MFMA rate — DESIGN.md §4 has the measured micro-benchmark:
                76 TFLOP/s register-only); `traffic` = HBM bytes per launch from the committed PMC passes;
  cpu_baseline  the literal CPU restatement of the reference algorithm (oracle/, NumPy/OpenBLAS) timed
                on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--nu", type=int, default=2)
    ap.add_argument("--levels", type=int, default=1)
    ap.add_argument("--samples-per-step", type=int, default=1024)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--no-mean-ite", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-units", type=int, default=2)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--binary-t", action="store_true", help="Bernoulli(0.5) treatments (BASELINE config 5 shape)")
    ap.add_argument("--fp32-kernel", action="store_true", help="mixed precision: RBF evaluation in fp32 (config 5)")
    return ap.parse_args()


def cpu_baseline(n, D, K, units, X, T, Y, post, doT):
    """Literal CPU restatement (oracle) timed on the host cores: `units` (sample, level) units."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gpslc_oracle as orc
    import numpy as np
    t0 = time.perf_counter()
    for s in range(units):
        p = orc.PosteriorSample(post["uyLS"][:, s] if K else None, post["xyLS"][:, s] if D else None,
                                float(post["tyLS"][s]), float(post["yNoise"][s]), float(post["yScale"][s]),
                                post["U"][:, :, s] if K else None)
        M, Cv = orc.ite_distributions([p], X, T, Y, doT)
        orc.conditional_sate(M[0], Cv[0])
    dt = time.perf_counter() - t0
    # the same unit with the structured algorithm the GPU path uses (one Cholesky, augmented right-hand sides):
    # what a CPU gains from the algorithm alone — reported beside the literal restatement, not instead of it
    t1 = time.perf_counter()
    for s in range(units):
        p = orc.PosteriorSample(post["uyLS"][:, s] if K else None, post["xyLS"][:, s] if D else None,
                                float(post["tyLS"][s]), float(post["yNoise"][s]), float(post["yScale"][s]),
                                post["U"][:, :, s] if K else None)
        orc.structured_sate(p, X, T, Y, np.array([doT]))
    dts = time.perf_counter() - t1
    try:
        from threadpoolctl import threadpool_info
        thr = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        thr = os.cpu_count() or 1
    return {"value": units / dt, "unit": "posterior samples/s", "cores": int(thr), "kind": "port",
            "sample": f"{units} (sample, level) units at N={n} D={D} nU={K}: literal restatement of the reference "
                      f"algorithm (5 kernel builds, 3 symmetric-indefinite solves, 4 GEMMs; NumPy/OpenBLAS), "
                      f"{dt:.1f} s wall; host has {os.cpu_count()} logical cores",
            "structured_value": units / dts,
            "structured_note": "same units with the structured algorithm of the GPU path (one Cholesky + augmented "
                               "right-hand sides, SATE mean/variance) in NumPy/SciPy on the same cores"}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # functional rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0 and the process
    # group is gloo (RCCL refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("GPSLC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # GPSLC_BENCH_FORCE_DIST=1: run the collective path (RCCL init, all_gather, barrier, all_reduce) even with a
    # single rank — the way to exercise the real "nccl" calls on a one-GPU box (under torch.distributed.run)
    use_dist = world > 1 or os.environ.get("GPSLC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import causalgpslc_jl_amd as gp
    from causalgpslc_jl_amd import synth

    n, D, K, L = a.n, a.d, a.nu, a.levels
    Sr = a.samples_per_step
    X, T, Y, obj = synth.make_dataset(n, D, binary_t=a.binary_t)
    post = synth.make_posterior(n, D, K, Sr, obj, seed=1234 + 17 * rank)   # every rank owns different samples
    doT = synth.levels(T, L)

    def to_dev(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x.reshape(-1, order="F"))).to(dev)

    dX, dT, dY = to_dev(X), to_dev(T), to_dev(Y)
    dU, duy, dxy = to_dev(post["U"]), to_dev(post["uyLS"]), to_dev(post["xyLS"])
    dty, dys, dyn, ddo = to_dev(post["tyLS"]), to_dev(post["yScale"]), to_dev(post["yNoise"]), to_dev(doT)
    mS = torch.empty(Sr * L, dtype=torch.float64, device=dev)
    vS = torch.empty(Sr * L, dtype=torch.float64, device=dev)
    mI = None if a.no_mean_ite else torch.empty(n * Sr * L, dtype=torch.float64, device=dev)

    def ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    ctx = gp.Context(n, D, K, device=local_rank, profile=not a.no_profile, fp32_kernel=a.fp32_kernel)
    ctx.check(ctx.lib.gpslc_set_data_dev(ctx.h, ptr(dX), ptr(dT), ptr(dY)))
    ctx.set_tuning(a.max_batch, a.panel, a.streams)

    gathered_m = [torch.empty_like(mS) for _ in range(world)] if use_dist else None
    gathered_v = [torch.empty_like(vS) for _ in range(world)] if use_dist else None

    def step():
        st = ctx.lib.gpslc_predict_dev(ctx.h, Sr, ptr(dU), ptr(duy), ptr(dxy), ptr(dty), ptr(dys), ptr(dyn), L,
                                       ptr(ddo), 1e-10, 0, 0, None, ptr(mS), ptr(vS), ptr(mI), None)
        if not (st > 0 and os.environ.get("GPSLC_GEMM_DIAG")):   # diagnostic kernels produce garbage (non-PD)
            ctx.check(st)
        if use_dist:   # the single end-of-step collective: SATE summaries of every rank's shard
            if rehearsal:   # gloo: gather through host memory
                gm = [torch.empty(Sr * L, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(gm, mS.cpu())
                dist.all_gather(gm, vS.cpu())
            else:
                dist.all_gather(gathered_m, mS)
                dist.all_gather(gathered_v, vS)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    launches, kms, kflop = ctx.profile_get(0)      # tile_gemm_nt_kernel<1, 0, 0>: the dominant kernel
    launches1, kms1, kflop1 = ctx.profile_get(1)   # <1, 0, 1>: in-panel column update fused with the panel solve
    kname = "tile_gemm_nt_kernel<1, 0, 0> (f64 MFMA tile update: trailing updates of the blocked Cholesky)"
    kname1 = "tile_gemm_nt_kernel<1, 0, 1> (f64 MFMA tile update: in-panel column update fused with the panel solve)"
    if kms1 > kms:
        # small N (at most one panel of tile columns): the in-panel instantiation is the dominant kernel
        launches, kms, kflop, launches1, kms1, kflop1 = launches1, kms1, kflop1, launches, kms, kflop
        kname, kname1 = kname1, kname

    if rank == 0:
        total_samples = Sr * world * a.steps
        val = total_samples / dt
        # sanity: results are finite and the two SATE paths agree (mean of MeanITE == MeanSATE)
        ms_h = mS.cpu().numpy()
        diag = bool(os.environ.get("GPSLC_GEMM_DIAG"))   # timing-only diagnostic kernels: results are garbage
        assert diag or np.all(np.isfinite(ms_h)), "non-finite SATE in the benchmark output"
        if mI is not None and not diag:
            mi_h = mI.cpu().numpy().reshape(n, Sr, L, order="F")
            chk = np.max(np.abs(mi_h.mean(axis=0)[:, 0] - ms_h.reshape(Sr, L, order="F")[:, 0]))
            assert chk <= (1e-5 if a.fp32_kernel else 1e-8) * max(1.0, np.max(np.abs(ms_h))), chk
        out = {
            "metric": "posterior samples/sec (kernel+chol+predict) at N=%d" % n,
            "value": val, "unit": "posterior samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if not a.fp32_kernel else "f64 factorisation, f32 kernel build", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo — not a result)" if rehearsal else ""),
            "config": {"workload": f"Synthetic N={n} D={D} nU={K}, unit A (Gram build + potrf + alpha + MeanITE + "
                                   f"SATE mean/var), L={L} level(s), {Sr} posterior samples per GPU per step"
                                   + (", binary treatment" if a.binary_t else ""),
                       "samples_per_gpu_per_step": Sr, "levels": L, "mean_ite": not a.no_mean_ite,
                       "sharding": f"posterior samples over {world} rank(s), all_gather of SATE at step end"},
        }
        if launches > 0 and kms > 0:
            ach = kflop / (kms * 1e-3) / 1e12
            # HBM bytes per launch of this kernel from the committed PMC passes of this same command
            # (separate rocprofv3 --pmc runs, FETCH_SIZE doubled per the gfx950 correction; tools/profile_r01.sh)
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_tile_gemm.json")
            if os.path.exists(pmc) and (n, D, K, L, Sr) == (4096, 8, 2, 1, 1024) and a.max_batch == 0 and a.panel == 0:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_note": "bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_tile_gemm.md",
                               "kernel": kname,
                               "launches": int(launches), "avg_launch_ms": kms / launches,
                               "algorithmic_flop_per_launch": kflop / launches,
                               "share_of_step_time": kms * 1e-3 / dt}
            if launches1 > 0 and kms1 > 0:
                out["roofline"]["second_kernel"] = {
                    "kernel": kname1,
                    "achieved": kflop1 / (kms1 * 1e-3) / 1e12, "frac": kflop1 / (kms1 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    "launches": int(launches1), "avg_launch_ms": kms1 / launches1,
                    "share_of_step_time": kms1 * 1e-3 / dt}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, D, K, a.cpu_units, X, T, Y, post, float(doT[0]))
        print(json.dumps(out), flush=True)
    if use_dist:
        if rank == 0 and not rehearsal:   # the gathered shards are what a caller would consume: check rank 0's own
            assert torch.equal(gathered_m[0], mS) and torch.equal(gathered_v[0], vS)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
