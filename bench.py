#!/usr/bin/env python3
"""bench.py — posterior samples/s of the GP hot path (Gram build + Cholesky + predict) at N = 4096.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a process group in the environment (no WORLD_SIZE): this process only launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a CHILD (before
anything touches the GPU) and exits with its code; under torch.distributed.run every rank runs main().  The
JSON line is printed by rank 0 only when the world size equals --gpus; a mismatch is an error (exit 3), never a
silent 1-GPU number.

A "step" processes one batch of `--samples-per-step` posterior samples per GPU of the BASELINE config
"Synthetic N=4096 D=8 nU=2" (SURVEY.md §8d unit A: Gram build + potrf(A) + alpha + MeanITE + SATE mean/variance
for L = 1 intervention level).  Inputs are resident in HBM before the timed region (gpslc_predict_dev takes
device pointers; torch only provides device memory and the process group).  Posterior samples shard over ranks
with no data-path collective; one all_gather of the (S x L) SATE arrays closes each step (weak scaling).

The JSON line carries
  roofline      the dominant kernel (tile_gemm_nt_kernel<1, 0, 0>, the f64-MFMA tile update): algorithmic flop
                (textbook count: diagonal tiles half, augmented rows by their live rows) / HIP-event time of every
                launch inside the timed region, against the fp64 matrix peak (78.6 TFLOP/s, AMD spec; the guides
                list no f64 MFMA rate — DESIGN.md §4 has the measured micro-benchmark: 76 TFLOP/s register-only);
                `traffic` = HBM bytes per launch from the committed PMC passes of this command, null when the
                kernel source changed since they were taken (the summary stores the source's git blob hash);
  sate_rel_err  the metric's second half: MeanSATE / VarSATE of this run's first posterior samples against the
                literal CPU restatement of the reference algorithm (src/estimation.jl:36-50, 116-121) — the same
                units the cpu_baseline leg times; above the SURVEY §8d tolerance the run fails (exit 4);
  units         SURVEY §8d units B (full ITE covariance + factor per (sample, level)) and C (predictive draws),
                measured in this run after the timed region (rank 0, N = 1 only), each with its own bound;
  cpu_baseline  the literal CPU restatement (oracle/, NumPy/OpenBLAS) timed on this box's host cores on a bounded
                sample (rank 0, N = 1 only), with the structured algorithm beside it.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBPS = 8000.0
KERNEL_SRC = os.path.join(ROOT, "causalgpslc.jl_amd", "csrc", "k_tilegemm.hip")
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r02_pmc_tile_gemm.json")


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
    ap.add_argument("--cpu-units", type=int, default=3)
    ap.add_argument("--no-units", action="store_true", help="skip the unit B / unit C measurements")
    ap.add_argument("--unit-b-samples", type=int, default=8, help="posterior samples of the unit B / C measurement")
    ap.add_argument("--unit-b-levels", type=int, default=16, help="intervention levels per posterior sample (they share one factor of A)")
    ap.add_argument("--unit-b-spp", type=int, default=10, help="draws per (sample, level): the reference's default")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--binary-t", action="store_true", help="Bernoulli(0.5) treatments (BASELINE config 5 shape)")
    ap.add_argument("--fp32-kernel", action="store_true", help="mixed precision: RBF evaluation in fp32 (config 5)")
    ap.add_argument("--diag-lib", action="store_true",
                    help="measurement only: load libgpslc_hip_diag.so (make diag), the build in which the GPSLC_* "
                         "environment switches exist; never used for reported numbers")
    return ap.parse_args()


def git_blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def self_launch(a):
    """--gpus N > 1 outside a process group: start the N ranks as a child job and pass its exit code on.
    Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise it)."""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but this box has {have} GPU(s); refusing to run a "
              f"{have}-GPU job under an {a.gpus}-GPU label", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def blas_info():
    try:
        from threadpoolctl import threadpool_info
        infos = [i for i in threadpool_info() if i.get("user_api") == "blas"] or threadpool_info()
        thr = max([i.get("num_threads", 1) for i in infos] + [1])
        desc = "; ".join(f"{i.get('internal_api')} {i.get('version')} ({i.get('architecture', '?')}, "
                         f"{i.get('threading_layer', '?')}, {i.get('num_threads')} threads)" for i in infos)
        return int(thr), desc
    except Exception:
        return os.cpu_count() or 1, "unknown BLAS"


def cpu_baseline(n, D, K, units, X, T, Y, post, doT):
    """Literal CPU restatement (oracle) timed on the host cores: `units` (sample, level) units.  Returns the
    baseline record and the (MeanSATE, VarSATE) pairs it computed — the parity reference for sate_rel_err."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gpslc_oracle as orc
    import numpy as np

    def sample(s):
        return orc.PosteriorSample(post["uyLS"][:, s] if K else None, post["xyLS"][:, s] if D else None,
                                   float(post["tyLS"][s]), float(post["yNoise"][s]), float(post["yScale"][s]),
                                   post["U"][:, :, s] if K else None)
    ref = []
    t0 = time.perf_counter()
    for s in range(units):
        M, Cv = orc.ite_distributions([sample(s)], X, T, Y, doT)
        ref.append(orc.conditional_sate(M[0], Cv[0]))
    dt = time.perf_counter() - t0
    # the same units with the structured algorithm the GPU path uses (one Cholesky, augmented right-hand sides):
    # what a CPU gains from the algorithm alone — reported beside the literal restatement, not instead of it
    t1 = time.perf_counter()
    for s in range(units):
        orc.structured_sate(sample(s), X, T, Y, np.array([doT]))
    dts = time.perf_counter() - t1
    thr, blas = blas_info()
    rec = {"value": units / dt, "unit": "posterior samples/s", "cores": thr, "kind": "port",
           "sample": f"{units} (sample, level) units at N={n} D={D} nU={K}: literal restatement of the reference "
                     f"algorithm (5 kernel builds, 3 symmetric-indefinite solves, 4 GEMMs; NumPy/SciPy on {blas}), "
                     f"{dt:.1f} s wall, extrapolated from these units; host has {os.cpu_count()} logical cores",
           "structured_value": units / dts,
           "structured_note": "same units with the structured algorithm of the GPU path (one Cholesky + augmented "
                              "right-hand sides, SATE mean/variance) in NumPy/SciPy on the same cores"}
    return rec, ref


def measure_units(gp, synth, np, torch, a, dev, local_rank, X, T, Y, obj, dX, dT, dY):
    """SURVEY §8d units B and C on this GPU (after the timed region; inputs resident in HBM).

    Unit B is defined GIVEN the factor of A ("one (sample, level) full-ITE unit given L"): the measurement sweeps
    `--unit-b-levels` intervention levels per posterior sample, the shape of predictCounterfactualEffects
    (src/prediction.jl:30-33), so the unit-A work of a sample is shared by its levels; the single-level case
    (64 samples x 1 level, unit-A work of every sample included) is reported beside it."""
    n, D, K = a.n, a.d, a.nu
    spp = a.unit_b_spp
    ctx = gp.Context(n, D, K, device=local_rank, profile=True, fp32_kernel=a.fp32_kernel)

    def to_dev(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x.reshape(-1, order="F"))).to(dev)

    def ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    ctx.check(ctx.lib.gpslc_set_data_dev(ctx.h, ptr(dX), ptr(dT), ptr(dY)))

    def run_case(Sb, Lb):
        post = synth.make_posterior(n, D, K, Sb, obj, seed=4321)
        doT = synth.levels(T, Lb)
        packs = [to_dev(post[k]) for k in ("U", "uyLS", "xyLS", "tyLS", "yScale", "yNoise")]
        ddo = to_dev(doT)
        mI = torch.empty(n * Sb * Lb, dtype=torch.float64, device=dev)
        dr = torch.empty(n * Sb * Lb * spp, dtype=torch.float64, device=dev)

        def run():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ctx.lib.gpslc_predict_dev(ctx.h, Sb, *[ptr(t) for t in packs], Lb, ptr(ddo), 1e-10, spp, 7, None,
                                           None, None, ptr(mI), ptr(dr))
            ctx.check(st)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        run()                   # warm-up: workspace allocation, first touch
        ctx.profile_reset()
        dt = run()
        prof = ctx.profile_get(2)
        assert bool(torch.isfinite(dr).all()), "non-finite predictive draws"
        info = ctx.last_info(Sb)
        assert not info.any(), f"CovITE factorisation broke down: info = {info[info != 0][:4]}"
        return dt, prof

    flop_b = 7.0 / 3.0 * float(n) ** 3
    Sb, Lb = a.unit_b_samples, a.unit_b_levels
    dt, (draws_l, draws_ms, draws_n) = run_case(Sb, Lb)
    dt1, _ = run_case(64, 1)
    units = Sb * Lb
    out = {"B": {"what": "one (sample, level) unit given the factor of A: W = D L^-T, CovITE + jitter = Delta - W W', its "
                         "Cholesky (src/estimation.jl:36-50, 82, 95-109); the levels of a sample share its factor of A "
                         "(src/prediction.jl:30-33); the timed call also computes those factors, MeanITE and the draws",
                 "value": units / dt, "unit": "(sample, level) units/s", "samples": Sb, "levels": Lb, "spp": spp,
                 "ms": 1e3 * dt, "bound": "mfma", "algorithmic_flop_per_unit": flop_b,
                 "achieved": units * flop_b / dt / 1e12, "peak": FP64_PEAK_TFLOPS, "roofline_unit": "TFLOP/s",
                 "frac": units * flop_b / dt / 1e12 / FP64_PEAK_TFLOPS,
                 "ceiling_units_per_s": FP64_PEAK_TFLOPS * 1e12 / flop_b,
                 "single_level": {"samples": 64, "levels": 1, "value": 64 / dt1, "ms": 1e3 * dt1,
                                  "frac": 64 * flop_b / dt1 / 1e12 / FP64_PEAK_TFLOPS,
                                  "note": "every unit pays its own unit-A work (Gram + factor of A + MeanITE) here"}}}
    if draws_l > 0 and draws_ms > 0:
        sec = draws_ms * 1e-3
        bytes_draw = 4.0 * float(n) ** 2           # one draw on its own reads the factor once (trmv)
        out["C"] = {"what": "predictive draws mu + L_c z given both factors (src/estimation.jl:105 with the factor "
                            "computed once per unit): all spp draws of a unit in one pass over L_c, f64 MFMA",
                    "value": draws_n / sec, "unit": "draws/s", "launches": int(draws_l),
                    "avg_launch_ms": draws_ms / draws_l, "draws_per_unit": spp, "bound": "hbm",
                    "algorithmic_bytes_per_draw": bytes_draw,
                    "achieved": draws_n * bytes_draw / sec / 1e9, "peak": HBM_PEAK_GBPS, "roofline_unit": "GB/s",
                    "frac": draws_n * bytes_draw / sec / 1e9 / HBM_PEAK_GBPS,
                    "factor_stream_GBps": (draws_n / spp) * bytes_draw / sec / 1e9,
                    "note": "achieved = draws x 4N^2 B / kernel time (the per-draw figure of SURVEY §8d; it exceeds "
                            "the HBM peak as soon as the spp draws of a unit share one pass over L_c); "
                            "factor_stream_GBps = the bytes of L_c actually streamed (once per unit) / time"}
    ctx.close()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # functional rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0 and the process
    # group is gloo (RCCL refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("GPSLC_BENCH_REHEARSAL") == "1"
    if world != a.gpus and not rehearsal:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {a.gpus} "
                  f"(or run `python bench.py --gpus {a.gpus}` outside torch.distributed.run)", file=sys.stderr, flush=True)
        sys.exit(3)
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
    if a.diag_lib:
        gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace("libgpslc_hip.so", "libgpslc_hip_diag.so")

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
        ctx.check(ctx.lib.gpslc_predict_dev(ctx.h, Sr, ptr(dU), ptr(duy), ptr(dxy), ptr(dty), ptr(dys), ptr(dyn), L,
                                            ptr(ddo), 1e-10, 0, 0, None, ptr(mS), ptr(vS), ptr(mI), None))
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

    rc = 0
    if rank == 0:
        total_samples = Sr * world * a.steps
        val = total_samples / dt
        # sanity: results are finite and the two SATE paths agree (mean of MeanITE == MeanSATE)
        ms_h = mS.cpu().numpy()
        vs_h = vS.cpu().numpy()
        assert np.all(np.isfinite(ms_h)), "non-finite SATE in the benchmark output"
        if mI is not None:
            mi_h = mI.cpu().numpy().reshape(n, Sr, L, order="F")
            chk = np.max(np.abs(mi_h.mean(axis=0)[:, 0] - ms_h.reshape(Sr, L, order="F")[:, 0]))
            assert chk <= (1e-5 if a.fp32_kernel else 1e-8) * max(1.0, np.max(np.abs(ms_h))), chk
        out = {
            "metric": "posterior samples/sec (kernel+chol+predict) at N=%d; SATE rel-err vs CPU" % n,
            "value": val, "unit": "posterior samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if not a.fp32_kernel else "f64 factorisation, f32 kernel build",
            "data": "synthetic" + (" (MEASUREMENT BUILD libgpslc_hip_diag.so — not a result)" if a.diag_lib else "") + (" (REHEARSAL: all ranks on one GPU, gloo — not a result)" if rehearsal else ""),
            "config": {"workload": f"Synthetic N={n} D={D} nU={K}, unit A (Gram build + potrf + alpha + MeanITE + "
                                   f"SATE mean/var), L={L} level(s), {Sr} posterior samples per GPU per step"
                                   + (", binary treatment" if a.binary_t else ""),
                       "samples_per_gpu_per_step": Sr, "levels": L, "mean_ite": not a.no_mean_ite,
                       "sharding": f"posterior samples over {world} rank(s), all_gather of SATE at step end"},
        }
        if launches > 0 and kms > 0:
            ach = kflop / (kms * 1e-3) / 1e12
            # HBM bytes per launch of this kernel from the committed PMC passes of this same command (separate
            # rocprofv3 --pmc runs, FETCH_SIZE doubled per the gfx950 correction; tools/profile_r02.sh).  The
            # summary records the git blob hash of the kernel source it was taken from: a different source today
            # means the number no longer describes this kernel, and it is withheld.
            traffic, tnote = None, "no PMC summary for this configuration"
            if os.path.exists(PMC_SUMMARY) and (n, D, K, L, Sr) == (4096, 8, 2, 1, 1024) and a.max_batch == 0 and a.panel == 0:
                pm = json.load(open(PMC_SUMMARY))
                if pm.get("kernel_src_sha") == git_blob_sha(KERNEL_SRC):
                    traffic = pm.get("hbm_bytes_per_launch")
                    tnote = ("bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE), "
                             "profiles/r02_pmc_tile_gemm.md; kernel source hash matches")
                else:
                    tnote = "STALE: k_tilegemm.hip changed since profiles/r02_pmc_tile_gemm.json was taken; withheld"
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_note": tnote,
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
        if world == 1 and not a.no_units:
            out["units"] = measure_units(gp, synth, np, torch, a, dev, local_rank, X, T, Y, obj, dX, dT, dY)
            out["units"]["A"] = {"what": "the headline value: one posterior sample (Gram + potrf + alpha + MeanITE + SATE)",
                                 "value": val, "unit": "posterior samples/s",
                                 "algorithmic_flop_per_unit": float(n) ** 3 / 3.0 + float(n) ** 2 * (3 * (D + K + 1) + 4 + 5 * L),
                                 "ceiling_units_per_s": FP64_PEAK_TFLOPS * 1e12 /
                                 (float(n) ** 3 / 3.0 + float(n) ** 2 * (3 * (D + K + 1) + 4 + 5 * L))}
        if world == 1 and not a.no_cpu_baseline:
            rec, ref = cpu_baseline(n, D, K, a.cpu_units, X, T, Y, post, float(doT[0]))
            out["cpu_baseline"] = rec
            # the metric's second half: this run's GPU results for the very units the CPU leg computed
            ms2, vs2 = ms_h.reshape(Sr, L, order="F"), vs_h.reshape(Sr, L, order="F")
            em = max(abs(ms2[s, 0] - ref[s][0]) / abs(ref[s][0]) for s in range(a.cpu_units))
            ev = max(abs(vs2[s, 0] - ref[s][1]) / abs(ref[s][1]) for s in range(a.cpu_units))
            tol = 1e-4 if a.fp32_kernel else 1e-6
            ok = all(abs(ms2[s, 0] - ref[s][0]) <= tol * abs(ref[s][0]) + 1e-12 and
                     abs(vs2[s, 0] - ref[s][1]) <= tol * abs(ref[s][1]) + 1e-9 * float(post["yScale"][s])
                     for s in range(a.cpu_units))
            out["sate_rel_err"] = {"mean": em, "var": ev, "units": a.cpu_units, "tolerance": tol, "ok": ok,
                                   "reference": "literal CPU restatement (oracle.ite_distributions + conditional_sate: "
                                                "src/estimation.jl:36-50, 82, 116-121), fp64, same inputs",
                                   "rule": "|dMean| <= tol |ref| + 1e-12 and |dVar| <= tol |ref| + 1e-9 yScale (SURVEY §8d)"}
            if not ok:
                rc = 4
        print(json.dumps(out), flush=True)
    if use_dist:
        if rank == 0 and not rehearsal:   # the gathered shards are what a caller would consume: check rank 0's own
            assert torch.equal(gathered_m[0], mS) and torch.equal(gathered_v[0], vS)
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
