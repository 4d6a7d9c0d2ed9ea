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
  roofline      the dominant kernel (tile_gemm_nt_kernel<1, 0>, the f64-MFMA tile update): algorithmic flop
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
# Three figures (HBM bytes per posterior sample at config 2; Gram / MeanITE fp64-VALU issue time per sample) are MEASURED
# quantities: they live in profiles/r06_bench_constants.json, each stamped with the git blob hashes of the sources it was measured
# on (tools/collect_profiles.py), and are withheld — like roofline.traffic — when any of those sources has changed since.
CSRC = os.path.join(ROOT, "causalgpslc.jl_amd", "csrc")
KERNEL_SRC = os.path.join(CSRC, "k_tilegemm.hip")
def pmc_summary_path(stem):
    """the newest committed PMC summary profiles/rNN_<stem>.json"""
    return next((p for p in (os.path.join(ROOT, "profiles", f"{r}_{stem}.json") for r in ("r06", "r05", "r04"))
                 if os.path.exists(p)), os.path.join(ROOT, "profiles", f"r06_{stem}.json"))


BENCH_CONSTANTS = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_bench_constants.json") for r in ("r06", "r05"))
                        if os.path.exists(p)), os.path.join(ROOT, "profiles", "r06_bench_constants.json"))


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
    ap.add_argument("--task-tiles", type=int, default=-1,
                    help="gpslc_set_task_schedule: matrices of up to this many tiles per side take the persistent "
                         "factorisation launch (0 = one launch per tile column, -1 = the library's default)")
    ap.add_argument("--task-min-tiles", type=int, default=0, help="gpslc_set_task_schedule: smallest tile count that takes the "
                                                                  "persistent launch (0 = the library's default)")
    ap.add_argument("--task-min-matrices", type=int, default=0, help="gpslc_set_task_schedule: smallest chunk (matrices) that takes "
                                                                     "the persistent launch (0 = the library's default)")
    ap.add_argument("--task-group", type=int, default=0, help="gpslc_set_task_schedule: matrices per group of the task order")
    ap.add_argument("--no-mean-ite", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-units", type=int, default=3)
    ap.add_argument("--no-units", action="store_true", help="skip the unit B / unit C measurements")
    ap.add_argument("--unit-b-samples", type=int, default=8, help="posterior samples of the unit B / C measurement")
    ap.add_argument("--unit-b-levels", type=int, default=16, help="intervention levels per posterior sample (they share one factor of A)")
    ap.add_argument("--unit-b-spp", type=int, default=10, help="draws per (sample, level): the reference's default")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps steps each: `value` is the first, `value_runs` lists all (box-to-box spread is "
                         "about 1 %; this shows the run-to-run spread on one box)")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE configs[3] leg (64 intervention levels per sample)")
    ap.add_argument("--config4-levels", type=int, default=64)
    ap.add_argument("--config4-steps", type=int, default=2)
    ap.add_argument("--no-configs", action="store_true", help="skip the short timed runs of BASELINE configs[1] and configs[4]")
    ap.add_argument("--no-panel-leg", action="store_true", help="skip the short leg that times the panel schedule of rounds 1-5 "
                                                                  "beside the default one (roofline.panel_schedule)")
    ap.add_argument("--no-multi-abi", action="store_true", help="skip the timing of gpslc_predict_multi's host delivery")
    ap.add_argument("--binary-t", action="store_true", help="Bernoulli(0.5) treatments (BASELINE config 5 shape)")
    ap.add_argument("--fp32-kernel", action="store_true", help="mixed precision: RBF evaluation in fp32 (config 5)")
    ap.add_argument("--diag-lib", action="store_true",
                    help="measurement only: load libgpslc_hip_diag.so (make diag), the build in which the GPSLC_* "
                         "environment switches exist; never used for reported numbers")
    ap.add_argument("--lib", default="", help="measurement only: load this build of the library instead (kernel A/B variants, "
                                              "`make variant`); the JSON line is labelled as a measurement")
    ap.add_argument("--timing-only", action="store_true",
                    help="measurement only (timing-only kernel variants whose results are garbage by construction): ignore "
                         "the status of the calls and skip every result check; the line is labelled")
    return ap.parse_args([x for x in sys.argv[1:] if x != "--"])


def git_blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(default_config, stem="pmc_tile_gemm"):
    """HBM bytes per launch of the dominant kernel (stem: which kernel's summary) from the committed PMC passes of the default
    command, or None with the reason: the summary records the git blob hash of the kernel source it was taken from, and a
    different source today means the number no longer describes this kernel."""
    path = pmc_summary_path(stem)
    if not default_config or not os.path.exists(path):
        return None, "no PMC summary for this configuration"
    pm = json.load(open(path))
    if pm.get("kernel_src_sha") != git_blob_sha(KERNEL_SRC):
        return None, f"STALE: k_tilegemm.hip changed since {os.path.relpath(path, ROOT)} was taken; withheld"
    return pm.get("hbm_bytes_per_launch"), ("bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE), "
                                             f"{os.path.relpath(path, ROOT)}; kernel source hash matches")


def measured_constant(name):
    """(value, note) of a measured constant of profiles/r06_bench_constants.json, or (None, reason) when the file is missing or
    one of the sources the measurement was taken on has changed (git blob hash) — a remembered number must not describe a
    kernel it was not measured on."""
    if not os.path.exists(BENCH_CONSTANTS):
        return None, f"{os.path.relpath(BENCH_CONSTANTS, ROOT)} is missing"
    rec = json.load(open(BENCH_CONSTANTS)).get(name)
    if not rec:
        return None, f"{name} is not in {os.path.relpath(BENCH_CONSTANTS, ROOT)}"
    for rel, sha in rec.get("source_shas", {}).items():
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path) or git_blob_sha(path) != sha:
            return None, f"STALE: {rel} changed since {rec.get('from', 'the measurement')} was taken; withheld"
    return rec["value"], f"{rec.get('from', '')}; source hashes match ({', '.join(sorted(rec.get('source_shas', {})))})"


def count_gpus_sysfs():
    """GPUs of this box from the KFD topology (nodes with SIMDs), without loading torch or any HIP runtime in the
    launcher process; None when the topology is not readable."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            props = dict(l.split()[:2] for l in open(os.path.join(base, d, "properties")) if len(l.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


def self_launch(a):
    """--gpus N > 1 outside a process group: start the N ranks as a child job and pass its exit code on.
    This launcher never touches the GPU: the device count comes from sysfs (torch's device_count() as the fallback
    only — it does not initialise the GPU on this image)."""
    have = count_gpus_sysfs()
    if have is None:
        import torch
        have = torch.cuda.device_count()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but this box has {have} GPU(s); refusing to run a "
              f"{have}-GPU job under an {a.gpus}-GPU label", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # "--" before this script's own flags: torch.distributed.run's parser otherwise claims every flag that is a prefix of
    # one of its own (--n 1024 -> "ambiguous option: --nnodes, --nproc-per-node, ...", --nu 2 -> --numa-binding)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--"] + \
          [x for x in sys.argv[1:] if x != "--"]
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
    unit0 = None
    t0 = time.perf_counter()
    for s in range(units):
        M, Cv = orc.ite_distributions([sample(s)], X, T, Y, doT)
        ref.append(orc.conditional_sate(M[0], Cv[0]))
        if s == 0:
            unit0 = (np.array(M[0]), np.array(Cv[0]))     # literal MeanITE and CovITE + jitter of (sample 0, doT)
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
    return rec, ref, unit0


def structured_unit0(n, D, K, X, T, Y, post, doT, pred_noise=1e-10):
    """(MeanITE, CovITE + jitter) of (sample 0, doT) from the structured restatement: the parity reference of unit B
    when the literal leg did not run (--no-cpu-baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gpslc_oracle as orc
    import numpy as np
    p = orc.PosteriorSample(post["uyLS"][:, 0] if K else None, post["xyLS"][:, 0] if D else None,
                            float(post["tyLS"][0]), float(post["yNoise"][0]), float(post["yScale"][0]),
                            post["U"][:, :, 0] if K else None)
    m, cov = orc.structured_ite(p, X, T, Y, doT)
    return np.array(m), (cov + cov.T) / 2 + pred_noise * np.eye(n)


def unit_b_parity(np, M0, C0, mi_gpu, draw_gpu, z, ref_name):
    """One (sample, level) unit of the full-covariance path at the size the line quotes unit B on: MeanITE and one
    predictive draw with the caller's normals against chol(CovITE + jitter) of the CPU restatement
    (src/estimation.jl:95-109).  Bound on the draw: SURVEY §8d's 1e-8 ||L_c|| ||z|| where the conditioning permits it,
    i.e. max(1e-8, 1e-15 cond(CovITE + jitter)) — a Cholesky factor moves by cond x the 1e-15 relative rounding of
    forming CovITE (tests/test_gpu_estimation.py uses the same rule)."""
    ev = np.linalg.eigvalsh(C0)
    lam_max, lam_min = float(ev[-1]), float(ev[0])
    Lc = np.linalg.cholesky(C0)
    ref = M0 + Lc @ z
    cond = lam_max / max(lam_min, 1e-300)
    bound = max(1e-8, 1e-15 * cond) * np.sqrt(lam_max) * float(np.linalg.norm(z)) + 1e-9 * float(np.linalg.norm(ref))
    err = float(np.linalg.norm(draw_gpu - ref))
    merr = float(np.max(np.abs(mi_gpu - M0)))
    mbound = 1e-6 * float(np.max(np.abs(M0))) + 1e-12
    # the tight guard: wherever cond < 1e8 fp64 delivers far better than the conditioning-aware bound —
    # 1e-9 ||L_c|| ||z|| is still 1e4 x the observed error, and a CovITE factor that regressed by 1e4 fails it
    tight = (1e-9 * np.sqrt(lam_max) * float(np.linalg.norm(z)) + 1e-12 * float(np.linalg.norm(ref))) if cond < 1e8 else None
    return {"draw_err": err, "draw_bound": bound, "draw_tight_bound": tight, "mean_ite_err": merr, "mean_ite_bound": mbound,
            "cond": cond, "ok": bool(err <= bound and merr <= mbound and (tight is None or err <= tight)),
            "reference": ref_name,
            "rule": "||draw - (M + chol(C) z)|| <= max(1e-8, 1e-15 cond(C)) sqrt(lambda_max(C)) ||z|| + 1e-9 ||ref||; "
                    "max|MeanITE - ref| <= 1e-6 max|ref| + 1e-12; C = CovITE + 1e-10 I (src/estimation.jl:82)"}


def measure_units(gp, synth, np, torch, a, dev, local_rank, X, T, Y, obj, dX, dT, dY, post0=None, doT0=None, unit0=None,
                  unit0_name=""):
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

    def parity_case():
        """(sample 0 of the timed run, level doT0) through the full-covariance path with the caller's normals."""
        packs = [to_dev(np.ascontiguousarray(post0[k][..., 0:1])) for k in ("U", "uyLS", "xyLS", "tyLS", "yScale", "yNoise")]
        z = np.random.default_rng(2024).standard_normal(n)
        dz, ddo = to_dev(z), to_dev(np.array([doT0]))
        mI = torch.empty(n, dtype=torch.float64, device=dev)
        dr = torch.empty(n, dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.gpslc_predict_dev(ctx.h, 1, *[ptr(t) for t in packs], 1, ptr(ddo), 1e-10, 1, 0, ptr(dz),
                                            None, None, ptr(mI), ptr(dr)))
        torch.cuda.synchronize()
        assert not ctx.last_info(1).any(), "CovITE factorisation broke down in the parity unit"
        return unit_b_parity(np, unit0[0], unit0[1], mI.cpu().numpy(), dr.cpu().numpy(), z, unit0_name)

    flop_b = 7.0 / 3.0 * float(n) ** 3
    flop_a1 = float(n) ** 3 / 3.0 + float(n) ** 2 * (3 * (D + K + 1) + 4 + 5)      # unit A with one level
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
                 # sampleITE(g, doT) with ONE level is the reference's most common call (src/driver.jl:86-89): every unit then pays
                 # its own unit-A work, so its ceiling is 1 / (unit A + unit B) = 78.6 TF / (23.6 + 160.3 GF) = 427/s at N = 4096
                 "single_level": {"samples": 64, "levels": 1, "value": 64 / dt1, "ms": 1e3 * dt1,
                                  "frac": 64 * flop_b / dt1 / 1e12 / FP64_PEAK_TFLOPS,
                                  "algorithmic_flop_per_unit_incl_unit_a": flop_b + flop_a1,
                                  "ceiling_units_per_s_incl_unit_a": FP64_PEAK_TFLOPS * 1e12 / (flop_b + flop_a1),
                                  "frac_incl_unit_a": 64 * (flop_b + flop_a1) / dt1 / 1e12 / FP64_PEAK_TFLOPS,
                                  "note": "every unit pays its own unit-A work (Gram + factor of A + MeanITE) here: frac is against "
                                          "unit B's flop alone (the 490/s ceiling), frac_incl_unit_a against the work the call "
                                          "really does (VERDICT r05 item 7)"}}}
    if unit0 is not None:
        out["B"]["parity"] = parity_case()
    if draws_l > 0 and draws_ms > 0:
        sec = draws_ms * 1e-3
        bytes_unit = 4.0 * float(n) ** 2           # the lower triangle of L_c: N^2/2 doubles, streamed ONCE per unit
        units_c = draws_n / spp
        out["C"] = {"what": "predictive draws mu + L_c z given both factors (src/estimation.jl:105 with the factor "
                            "computed once per unit): all spp draws of a unit in one pass over L_c, f64 MFMA",
                    "value": draws_n / sec, "unit": "draws/s", "launches": int(draws_l),
                    "avg_launch_ms": draws_ms / draws_l, "draws_per_unit": spp, "bound": "hbm",
                    "algorithmic_bytes_per_unit": bytes_unit,
                    "achieved": units_c * bytes_unit / sec / 1e9, "peak": HBM_PEAK_GBPS, "roofline_unit": "GB/s",
                    "frac": units_c * bytes_unit / sec / 1e9 / HBM_PEAK_GBPS,
                    "per_draw_equivalent_GBps": draws_n * bytes_unit / sec / 1e9,
                    "note": "achieved = bytes of L_c actually streamed (4N^2 B once per unit, shared by its spp draws) / "
                            "kernel time; per_draw_equivalent_GBps = SURVEY §8d's per-draw figure (4N^2 B per draw), "
                            "which is not a traffic: it exceeds the HBM peak as soon as draws share a pass"}
    ctx.close()
    return out


def run_config(gp, synth, np, torch, dev, local_rank, n, D, K, S, L, binary_t, fp32, steps, warmup, label, literal_golden=None):
    """One short timed run of unit A (with MeanITE) on another BASELINE configuration, inputs resident in HBM; this
    run's first unit is checked against the structured oracle evaluated in fp64 (SURVEY §8d tolerances)."""
    X, T, Y, obj = synth.make_dataset(n, D, binary_t=binary_t)
    post = synth.make_posterior(n, D, K, S, obj, seed=99)
    doT = synth.levels(T, L) if not binary_t else np.array([1.0])[:L]

    def to_dev(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x.reshape(-1, order="F"))).to(dev)

    def ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    dX, dT, dY = to_dev(X), to_dev(T), to_dev(Y)
    packs = [to_dev(post[k]) for k in ("U", "uyLS", "xyLS", "tyLS", "yScale", "yNoise")]
    ddo = to_dev(doT)
    mS = torch.empty(S * L, dtype=torch.float64, device=dev)
    vS = torch.empty(S * L, dtype=torch.float64, device=dev)
    mI = torch.empty(n * S * L, dtype=torch.float64, device=dev)
    ctx = gp.Context(n, D, K, device=local_rank, profile=True, fp32_kernel=fp32)
    ctx.check(ctx.lib.gpslc_set_data_dev(ctx.h, ptr(dX), ptr(dT), ptr(dY)))

    def step():
        ctx.check(ctx.lib.gpslc_predict_dev(ctx.h, S, *[ptr(t) for t in packs], L, ptr(ddo), 1e-10, 0, 0, None,
                                            ptr(mS), ptr(vS), ptr(mI), None))
    for _ in range(warmup):
        step()
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = [ctx.profile_get(k) for k in (0, 1, 4)]
    ctx.close()
    flop = float(n) ** 3 / 3.0 + float(n) ** 2 * (3 * (D + K + 1) + 4 + 5 * L)
    val = S * steps / dt
    rec = {"workload": label, "value": val, "unit": "posterior samples/s", "samples_per_step": S, "steps": steps,
           "ms_per_step": 1e3 * dt / steps, "dtype": "f64 factorisation, f32 kernel build" if fp32 else "f64",
           "algorithmic_flop_per_unit": flop, "ceiling_units_per_s": FP64_PEAK_TFLOPS * 1e12 / flop,
           "frac_of_ceiling": val * flop / (FP64_PEAK_TFLOPS * 1e12)}
    if (n, D, K, L, S) == (1024, 4, 1, 1, 8192) and not fp32:
        # Config 2 moves ~39 MB of HBM traffic per posterior sample (PMC: FETCH_SIZE x 2 + WRITE_SIZE summed over the kernels of
        # an 8,192-sample chunk), eight times the 4.7 MB matrix: the left-looking schedule's bytes (profiles/r04_ab_experiments.md
        # §5).  The figure is read from the committed measurement of THIS chunk size and THESE sources, or withheld.
        bps, bnote = measured_constant("c2_hbm_bytes_per_sample")
        if bps is None:
            rec["hbm"] = {"bytes_per_unit": None, "frac": None, "note": bnote}
        else:
            rec["hbm"] = {"bytes_per_unit": bps, "achieved": val * bps / 1e9,
                          "peak": HBM_PEAK_GBPS, "roofline_unit": "GB/s", "frac": val * bps / 1e9 / HBM_PEAK_GBPS,
                          "ceiling_units_per_s": HBM_PEAK_GBPS * 1e9 / bps,
                          "note": "HBM bytes per sample from the PMC passes (" + bnote + ") x samples/s; frac_of_ceiling above "
                                  "is against the MFMA peak"}
    names = ("trailing_update_kernel", "fused_in_panel_kernel", "task_kernel")
    for nm, (ln, ms, fl) in zip(names, prof):
        if ln > 0 and ms > 0:
            rec[nm] = {"achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                       "share_of_time": ms * 1e-3 / dt, "roofline_unit": "TFLOP/s"}
    # parity of this run's first unit: structured oracle in fp64
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gpslc_oracle as orc
    p0 = orc.PosteriorSample(post["uyLS"][:, 0] if K else None, post["xyLS"][:, 0] if D else None,
                             float(post["tyLS"][0]), float(post["yNoise"][0]), float(post["yScale"][0]),
                             post["U"][:, :, 0] if K else None)
    rm, rv = orc.structured_sate(p0, X, T, Y, np.asarray(doT[:1], dtype=np.float64))[:2]
    rm, rv = float(np.ravel(rm)[0]), float(np.ravel(rv)[0])
    gm, gv = float(mS[0].item()), float(vS[0].item())
    tol = 1e-6
    rec["parity"] = {"mean_rel_err": abs(gm - rm) / abs(rm), "var_rel_err": abs(gv - rv) / abs(rv), "tolerance": tol,
                     "ok": bool(abs(gm - rm) <= tol * abs(rm) + 1e-12 and
                                abs(gv - rv) <= tol * abs(rv) + 1e-9 * float(post["yScale"][0])),
                     "reference": "structured CPU restatement in fp64 (oracle.structured_sate), first unit of this run"}
    if literal_golden:
        lit = literal_golden_check(gp, np, torch, dev, local_rank, n, D, K, X, T, Y, obj, binary_t, fp32, literal_golden)
        rec["parity"]["literal_golden"] = lit
        rec["parity"]["ok"] = bool(rec["parity"]["ok"] and lit.get("ok", True))
    return rec


def literal_golden_check(gp, np, torch, dev, local_rank, n, D, K, X, T, Y, obj, binary_t, fp32, path):
    """The committed LITERAL restatement of this configuration's shape (tests/golden/config5_literal.npz: one (sample, level) unit
    at N = 16384, 11 minutes of host CPU — tests/golden/make_golden_config5.py) against one more gpslc_predict_dev call on the
    golden's own inputs, in the mode this configuration is timed in.  The inputs are regenerated from the same seeds and their
    checksums compared with the ones stored beside the outputs; on a mismatch nothing is claimed."""
    from causalgpslc_jl_amd import synth
    gold = np.load(path)
    gn, gD, gK, gS = (int(v) for v in gold["shape"])
    if (gn, gD, gK) != (n, D, K) or gS != 1:
        return {"skipped": f"the golden holds N={gn} D={gD} nU={gK}"}
    post = synth.make_posterior(n, D, K, 1, obj)
    chk = np.array([X.sum(), T.sum(), Y.sum(), post["U"].sum(), post["uyLS"].sum(), post["xyLS"].sum(),
                    post["tyLS"][0], post["yNoise"][0], post["yScale"][0]])
    if not np.allclose(chk, gold["in_checksums"], rtol=1e-13, atol=0):
        return {"skipped": "the synthetic generator no longer reproduces the golden's inputs (checksums differ)"}

    def to_dev(x):
        return torch.from_numpy(np.ascontiguousarray(x.reshape(-1, order="F"))).to(dev)

    def ptr(t):
        return C.c_void_p(t.data_ptr())

    ctx = gp.Context(n, D, K, device=local_rank, fp32_kernel=fp32)
    dX, dT, dY = to_dev(X), to_dev(T), to_dev(Y)
    ctx.check(ctx.lib.gpslc_set_data_dev(ctx.h, ptr(dX), ptr(dT), ptr(dY)))
    packs = [to_dev(post[k]) for k in ("U", "uyLS", "xyLS", "tyLS", "yScale", "yNoise")]
    ddo = to_dev(np.array([float(gold["doT"])]))
    mS = torch.empty(1, dtype=torch.float64, device=dev)
    vS = torch.empty(1, dtype=torch.float64, device=dev)
    mI = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.check(ctx.lib.gpslc_predict_dev(ctx.h, 1, *[ptr(t) for t in packs], 1, ptr(ddo), 1e-10, 0, 0, None,
                                        ptr(mS), ptr(vS), ptr(mI), None))
    torch.cuda.synchronize()
    ctx.close()
    rm, rv, m_ref = float(gold["meanSATE"]), float(gold["varSATE"]), gold["meanITE"]
    gm, gv, mi = float(mS.item()), float(vS.item()), mI.cpu().numpy()
    tol = 1e-6
    em, ev = abs(gm - rm) / abs(rm), abs(gv - rv) / abs(rv)
    ei = float(np.max(np.abs(mi - m_ref)) / np.max(np.abs(m_ref)))
    return {"mean_rel_err": em, "var_rel_err": ev, "mean_ite_rel_err": ei, "tolerance": tol,
            "mode": "fp32 kernel build + fp64 Cholesky" if fp32 else "fp64",
            "ok": bool(abs(gm - rm) <= tol * abs(rm) + 1e-12 and abs(gv - rv) <= tol * abs(rv) + 1e-9 * float(post["yScale"][0])
                       and ei <= tol),
            "reference": "LITERAL CPU restatement at full size (oracle.ite_distributions + conditional_sate: 5 kernel builds, 3 "
                         "symmetric-indefinite solves, 4 GEMMs; src/likelihood.jl:24-49, src/estimation.jl:46-47, 82, 116-121), "
                         f"committed as {os.path.relpath(path, ROOT)} ({float(gold['seconds']):.0f} s of host CPU when it was "
                         "generated); unit = (S = 1 posterior of seed 1234, doT = 1) on this configuration's data set"}


def write_bench_pack(path, X, T, Y, post, binary_t):
    """The data set and the posterior samples of ALL ranks as one posterior pack (gpslc_pack_save): generated once per
    node by rank 0, every rank then loads only its own block of samples (gpslc_pack_load(path, s0, s1))."""
    import numpy as np
    from causalgpslc_jl_amd import _lib
    from causalgpslc_jl_amd.api import _p
    n = len(T)
    nX = 0 if X is None else X.shape[1]
    nU = 0 if post["U"] is None else post["U"].shape[1]
    h = _lib.PackHeader(n, nX, nU, len(post["tyLS"]), 1 if binary_t else 0, 0)
    for i, v in enumerate((nU if nU else -1.0, 0, 0, 0, 0, 1, 1e-10)):
        h.hyper[i] = float(v)
    arrs = [None if x is None else np.asfortranarray(x, dtype=np.float64)
            for x in (X, T, Y, post["U"], post["uyLS"], post["xyLS"], post["tyLS"], post["yNoise"], post["yScale"])]
    st = _lib.load().gpslc_pack_save(os.fsencode(path), C.byref(h), *[_p(x) for x in arrs])
    if st != 0:
        raise RuntimeError(f"gpslc_pack_save({path}): status {st}")


def read_bench_pack(path, s0, s1):
    """(X, T, Y, post) with the posterior samples [s0, s1) of the pack rank 0 wrote."""
    from causalgpslc_jl_amd import pack
    g = pack.loadGPSLCObject(path, samples=(s0, s1))
    post = {"U": g.U, "uyLS": g.uyLS, "xyLS": g.xyLS, "tyLS": g.tyLS, "yNoise": g.yNoise, "yScale": g.yScale}
    return g.X, g.T, g.Y, post


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and rank == 0 and not a.no_cpu_baseline:
        # torch.distributed.run exports OMP_NUM_THREADS=1 to its workers; rank 0 also runs the CPU leg, whose BLAS pool should be
        # the one an N = 1 run gets.  Set before NumPy / SciPy load their OpenBLAS (raising the pool afterwards through
        # threadpoolctl crashed the bundled OpenBLAS on the GPU box).
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ[k] = str(min(64, os.cpu_count() or 1))

    import numpy as np
    import torch
    import torch.distributed as dist

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
    if a.lib:
        gp._lib.LIB_PATH = os.path.abspath(a.lib)

    n, D, K, L = a.n, a.d, a.nu, a.levels
    Sr = a.samples_per_step
    if world == 1:
        X, T, Y, obj = synth.make_dataset(n, D, binary_t=a.binary_t)
        post = synth.make_posterior(n, D, K, Sr, obj, seed=1234)
    else:
        # one data set and one posterior of Sr * world samples per NODE: rank 0 generates and writes a posterior pack, every
        # rank loads its own contiguous block of samples — the partition of sharded.py (src/prediction.jl:30-33 over
        # src/estimation.jl:78-84: the loop over posterior samples is what shards)
        pack_path = os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                 f"gpslc_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}.pack")
        obj = None
        if rank == 0:
            X0, T0, Y0, obj = synth.make_dataset(n, D, binary_t=a.binary_t)
            write_bench_pack(pack_path, X0, T0, Y0, synth.make_posterior(n, D, K, Sr * world, obj, seed=1234), a.binary_t)
        dist.barrier()
        X, T, Y, post = read_bench_pack(pack_path, rank * Sr, (rank + 1) * Sr)
        dist.barrier()
        if rank == 0:
            os.remove(pack_path)
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
    ctx.set_task_schedule(a.task_min_tiles, a.task_tiles, a.task_min_matrices, a.task_group)

    gathered_m = [torch.empty_like(mS) for _ in range(world)] if use_dist else None
    gathered_v = [torch.empty_like(vS) for _ in range(world)] if use_dist else None

    def step():
        st = ctx.lib.gpslc_predict_dev(ctx.h, Sr, ptr(dU), ptr(duy), ptr(dxy), ptr(dty), ptr(dys), ptr(dyn), L,
                                       ptr(ddo), 1e-10, 0, 0, None, ptr(mS), ptr(vS), ptr(mI), None)
        if not a.timing_only:
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
    launches, kms, kflop = ctx.profile_get(0)      # tile_gemm_nt_kernel<1, 0>: the dominant kernel
    launches1, kms1, kflop1 = ctx.profile_get(1)   # tile_fused_strip_kernel: in-panel column update fused with the panel solve
    ctx_prof4 = ctx.profile_get(4)                 # potrf_tasks_kernel: the persistent factorisation launch (N <= 4096)
    # the spread of the measurement: the same K steps timed again (--repeats - 1 more regions, bracketed like the first).  `value`
    # stays the FIRST region (exactly K steps after W warm-up steps, as the contract says); value_runs lists all of them.
    region_s = [dt]
    for _ in range(max(0, a.repeats - 1)):
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        d1 = time.perf_counter() - t1
        if use_dist:
            tt = torch.tensor([d1], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d1 = float(tt.item())
        region_s.append(d1)
    kname = "tile_gemm_nt_kernel<1, 0> (f64 MFMA tile update: trailing updates of the blocked Cholesky)"
    kname1 = "tile_fused_strip_kernel<8> (f64 MFMA tile update: in-panel column update fused with the panel solve)"
    # Round 6: at the default sizes the whole factorisation is ONE persistent launch of tile tasks.  The schedule of rounds 1-5
    # (left-looking panels of 8 tile columns, one launch per column, one trailing update per panel) is timed beside it in a short
    # untimed-for-`value` leg, so that the trailing-update kernel's roofline figure stays comparable across rounds.
    panel_leg = None
    if ctx_prof4[1] > 0 and a.task_tiles < 0 and not a.timing_only and not a.no_profile and not a.no_panel_leg:
        ctx.set_task_schedule(0, 0, 0, 0)
        step()
        ctx.profile_reset()
        fence()
        tp = time.perf_counter()
        psteps = min(a.steps, 3)
        for _ in range(psteps):
            step()
        fence()
        dtp = time.perf_counter() - tp
        if use_dist:
            tt = torch.tensor([dtp], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtp = float(tt.item())
        pl0, pms0, pfl0 = ctx.profile_get(0)
        pl1, pms1, pfl1 = ctx.profile_get(1)
        panel_leg = {"what": "the same step with one launch per tile column (gpslc_set_task_schedule(max_tiles = 0): panels of 8 tile "
                             "columns + one trailing update per panel, the schedule of rounds 1-5), timed after the regions above",
                     "value": Sr * world * psteps / dtp, "unit": "posterior samples/s", "steps": psteps,
                     "ms_per_step": 1e3 * dtp / psteps}
        if pl0 > 0 and pms0 > 0:
            panel_leg["trailing_update_kernel"] = {"kernel": kname, "achieved": pfl0 / (pms0 * 1e-3) / 1e12,
                                                   "frac": pfl0 / (pms0 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "launches": int(pl0),
                                                   "avg_launch_ms": pms0 / pl0, "share_of_step_time": pms0 * 1e-3 / dtp}
        if pl1 > 0 and pms1 > 0:
            panel_leg["fused_in_panel_kernel"] = {"kernel": kname1, "achieved": pfl1 / (pms1 * 1e-3) / 1e12,
                                                  "frac": pfl1 / (pms1 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "launches": int(pl1),
                                                  "avg_launch_ms": pms1 / pl1, "share_of_step_time": pms1 * 1e-3 / dtp}
        if "trailing_update_kernel" in panel_leg:
            ptr_, pnote = pmc_traffic((n, D, K, L, Sr) == (4096, 8, 2, 1, 1024) and a.max_batch == 0 and a.panel == 0, "pmc_tile_gemm")
            panel_leg["trailing_update_kernel"]["traffic"] = ptr_
            panel_leg["trailing_update_kernel"]["traffic_note"] = pnote
        ctx.set_task_schedule(a.task_min_tiles, 32, a.task_min_matrices, a.task_group)

    # ---- BASELINE configs[3]: the same posterior samples x 64 intervention levels (the sweep of
    # src/prediction.jl:30-33 over src/estimation.jl:78-84), sharded like the L = 1 region: second timed region
    c4 = None
    if not a.no_config4 and not a.timing_only:
        L4 = a.config4_levels
        doT4 = synth.levels(T, L4)
        ddo4 = to_dev(doT4)
        mS4 = torch.empty(Sr * L4, dtype=torch.float64, device=dev)
        vS4 = torch.empty(Sr * L4, dtype=torch.float64, device=dev)
        mI4 = None if a.no_mean_ite else torch.empty(n * Sr * L4, dtype=torch.float64, device=dev)
        g4m = [torch.empty_like(mS4) for _ in range(world)] if use_dist and not rehearsal else None
        g4v = [torch.empty_like(vS4) for _ in range(world)] if use_dist and not rehearsal else None

        def step4():
            ctx.check(ctx.lib.gpslc_predict_dev(ctx.h, Sr, ptr(dU), ptr(duy), ptr(dxy), ptr(dty), ptr(dys), ptr(dyn), L4,
                                                ptr(ddo4), 1e-10, 0, 0, None, ptr(mS4), ptr(vS4), ptr(mI4), None))
            if use_dist:
                if rehearsal:
                    gm = [torch.empty(Sr * L4, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(gm, mS4.cpu())
                    dist.all_gather(gm, vS4.cpu())
                else:
                    dist.all_gather(g4m, mS4)
                    dist.all_gather(g4v, vS4)
        step4()
        fence()
        t4 = time.perf_counter()
        for _ in range(a.config4_steps):
            step4()
        fence()
        dt4 = time.perf_counter() - t4
        if use_dist:
            tt = torch.tensor([dt4], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt4 = float(tt.item())
        c4 = {"workload": f"BASELINE configs[3] shape: N={n} D={D} nU={K}, {Sr} posterior samples per GPU per step x {L4} "
                          f"intervention levels (unit A with MeanITE for every level; the levels of a sample share its "
                          f"factor of A), sharded over {world} rank(s), all_gather of the (S x L) SATE arrays at step end",
              "value": Sr * world * a.config4_steps / dt4, "unit": "posterior samples/s",
              "sample_level_units_per_s": Sr * world * L4 * a.config4_steps / dt4, "levels": L4,
              "steps": a.config4_steps, "warmup": 1, "ms_per_step": 1e3 * dt4 / a.config4_steps, "scaling": "weak",
              "mean_ite": not a.no_mean_ite}
        if rank == 0:
            # parity of rank 0's first unit: SATE mean / variance of all levels and MeanITE of the two end levels against
            # the structured CPU restatement
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import gpslc_oracle as orc
            p0 = orc.PosteriorSample(post["uyLS"][:, 0] if K else None, post["xyLS"][:, 0] if D else None,
                                     float(post["tyLS"][0]), float(post["yNoise"][0]), float(post["yScale"][0]),
                                     post["U"][:, :, 0] if K else None)
            rm, rv = orc.structured_sate(p0, X, T, Y, doT4)[:2]
            gm4 = mS4.cpu().numpy().reshape(Sr, L4, order="F")[0]
            gv4 = vS4.cpu().numpy().reshape(Sr, L4, order="F")[0]
            tol = 1e-4 if a.fp32_kernel else 1e-6
            okm = bool(np.all(np.abs(gm4 - rm) <= tol * np.abs(rm) + 1e-12))
            okv = bool(np.all(np.abs(gv4 - rv) <= tol * np.abs(rv) + 1e-9 * float(post["yScale"][0])))
            par = {"mean_rel_err": float(np.max(np.abs(gm4 - rm) / np.abs(rm))),
                   "var_rel_err": float(np.max(np.abs(gv4 - rv) / np.abs(rv))), "levels_checked": L4, "tolerance": tol,
                   "reference": "structured CPU restatement (oracle.structured_sate / structured_ite), rank 0's first unit"}
            if mI4 is not None:
                mi4 = mI4.cpu().numpy().reshape(n, Sr, L4, order="F")[:, 0, :]
                e = 0.0
                for l in (0, L4 - 1):
                    m_ref = orc.structured_ite(p0, X, T, Y, float(doT4[l]))[0]
                    e = max(e, float(np.max(np.abs(mi4[:, l] - m_ref)) / np.max(np.abs(m_ref))))
                par["mean_ite_rel_err"] = e
                okm = okm and e <= tol
            if not a.no_cpu_baseline:
                # one (sample, level) unit of this region against the LITERAL restatement as well (5 kernel builds, 3 symmetric-
                # indefinite solves, 4 GEMMs: src/likelihood.jl:24-49, src/estimation.jl:46-47, 82, 116-121) — the structured
                # restatement above shares the GPU path's algebra, this one does not
                ll = L4 // 2
                tl0 = time.perf_counter()
                Ml, Cl = orc.ite_distributions([p0], X, T, Y, float(doT4[ll]))
                lm, lv = orc.conditional_sate(Ml[0], Cl[0])
                lit = {"level": ll, "doT": float(doT4[ll]), "mean_rel_err": abs(float(gm4[ll]) - lm) / abs(lm),
                       "var_rel_err": abs(float(gv4[ll]) - lv) / abs(lv), "seconds": time.perf_counter() - tl0,
                       "reference": "literal CPU restatement (oracle.ite_distributions + conditional_sate), fp64"}
                okl = (abs(float(gm4[ll]) - lm) <= tol * abs(lm) + 1e-12 and
                       abs(float(gv4[ll]) - lv) <= tol * abs(lv) + 1e-9 * float(post["yScale"][0]))
                if mI4 is not None:
                    mil = mI4.cpu().numpy().reshape(n, Sr, L4, order="F")[:, 0, ll]
                    lit["mean_ite_rel_err"] = float(np.max(np.abs(mil - Ml[0])) / np.max(np.abs(Ml[0])))
                    okl = okl and lit["mean_ite_rel_err"] <= tol
                lit["ok"] = bool(okl)
                par["literal_unit"] = lit
                okm = okm and okl
                del Ml, Cl
            par["ok"] = okm and okv
            c4["parity"] = par
        del mS4, vS4, mI4

    if kms1 > kms:
        # small N (at most one panel of tile columns): the in-panel instantiation is the dominant kernel
        launches, kms, kflop, launches1, kms1, kflop1 = launches1, kms1, kflop1, launches, kms, kflop
        kname, kname1 = kname1, kname
    launches4, kms4, kflop4 = ctx_prof4
    if kms4 > kms:
        # N <= 4096 with the default schedule: the whole factorisation is ONE persistent launch of tile tasks
        launches1, kms1, kflop1 = launches, kms, kflop
        kname1 = kname
        launches, kms, kflop = launches4, kms4, kflop4
        kname = ("potrf_tasks_kernel (f64 MFMA tile tasks: the whole left-looking factorisation of a chunk in one persistent "
                 "launch; flop = n^3/3 + right-hand sides x n^2 per matrix)")

    rc = 0
    if rank == 0:
        total_samples = Sr * world * a.steps
        val = total_samples / dt
        # sanity: results are finite and the two SATE paths agree (mean of MeanITE == MeanSATE)
        ms_h = mS.cpu().numpy()
        vs_h = vS.cpu().numpy()
        assert a.timing_only or np.all(np.isfinite(ms_h)), "non-finite SATE in the benchmark output"
        if mI is not None and not a.timing_only:
            mi_h = mI.cpu().numpy().reshape(n, Sr, L, order="F")
            chk = np.max(np.abs(mi_h.mean(axis=0)[:, 0] - ms_h.reshape(Sr, L, order="F")[:, 0]))
            assert chk <= (1e-5 if a.fp32_kernel else 1e-8) * max(1.0, np.max(np.abs(ms_h))), chk
        out = {
            "metric": "posterior samples/sec (kernel+chol+predict) at N=%d; SATE rel-err vs CPU" % n,
            "value": val, "unit": "posterior samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_runs": [total_samples / d for d in region_s],
            "dtype": "f64" if not a.fp32_kernel else "f64 factorisation, f32 kernel build",
            "data": "synthetic" + (" (MEASUREMENT BUILD libgpslc_hip_diag.so — not a result)" if a.diag_lib else "") + (f" (MEASUREMENT BUILD {os.path.basename(a.lib)} — not a result)" if a.lib else "") + (" (TIMING ONLY: results unchecked)" if a.timing_only else "") + (" (REHEARSAL: all ranks on one GPU, gloo — not a result)" if rehearsal else ""),
            "config": {"workload": f"Synthetic N={n} D={D} nU={K}, unit A (Gram build + potrf + alpha + MeanITE + "
                                   f"SATE mean/var), L={L} level(s), {Sr} posterior samples per GPU per step"
                                   + (", binary treatment" if a.binary_t else ""),
                       "samples_per_gpu_per_step": Sr, "levels": L, "mean_ite": not a.no_mean_ite,
                       "sharding": f"posterior samples over {world} rank(s), all_gather of SATE at step end"},
        }
        if launches > 0 and kms > 0:
            ach = kflop / (kms * 1e-3) / 1e12
            # HBM bytes per launch of this kernel from the committed PMC passes of this same command (separate
            # rocprofv3 --pmc runs, FETCH_SIZE doubled per the gfx950 correction; tools/profile_r04.sh).  The
            # summary records the git blob hash of the kernel source it was taken from: a different source today
            # means the number no longer describes this kernel, and it is withheld.
            traffic, tnote = pmc_traffic((n, D, K, L, Sr) == (4096, 8, 2, 1, 1024) and a.max_batch == 0 and a.panel == 0,
                                         "pmc_potrf_tasks" if kname.startswith("potrf_tasks") else "pmc_tile_gemm")
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
        if panel_leg is not None and "roofline" in out:
            out["roofline"]["panel_schedule"] = panel_leg
        if c4 is not None:
            out["config4"] = c4
            if not c4.get("parity", {}).get("ok", True):
                rc = 4
        unit0, unit0_name = None, ""
        if not a.no_cpu_baseline and not a.timing_only:
            # N > 1: rank 0 times the CPU leg on ITS block of samples (fewer units, so the other ranks, parked at the
            # barrier below, wait about 30 s); the line then carries cpu_baseline and sate_rel_err for every N
            cpu_units = a.cpu_units if world == 1 else min(a.cpu_units, 2)
            rec, ref, unit0 = cpu_baseline(n, D, K, cpu_units, X, T, Y, post, float(doT[0]))
            unit0_name = "literal CPU restatement (oracle.ite_distributions: src/estimation.jl:36-50, 82), fp64"
            out["cpu_baseline"] = rec
            # the metric's second half: this run's GPU results for the very units the CPU leg computed
            ms2, vs2 = ms_h.reshape(Sr, L, order="F"), vs_h.reshape(Sr, L, order="F")
            em = max(abs(ms2[s, 0] - ref[s][0]) / abs(ref[s][0]) for s in range(cpu_units))
            ev = max(abs(vs2[s, 0] - ref[s][1]) / abs(ref[s][1]) for s in range(cpu_units))
            tol = 1e-4 if a.fp32_kernel else 1e-6
            ok = all(abs(ms2[s, 0] - ref[s][0]) <= tol * abs(ref[s][0]) + 1e-12 and
                     abs(vs2[s, 0] - ref[s][1]) <= tol * abs(ref[s][1]) + 1e-9 * float(post["yScale"][s])
                     for s in range(cpu_units))
            out["sate_rel_err"] = {"mean": em, "var": ev, "units": cpu_units, "tolerance": tol, "ok": ok,
                                   "reference": "literal CPU restatement (oracle.ite_distributions + conditional_sate: "
                                                "src/estimation.jl:36-50, 82, 116-121), fp64, same inputs",
                                   "rule": "|dMean| <= tol |ref| + 1e-12 and |dVar| <= tol |ref| + 1e-9 yScale (SURVEY §8d)"}
            if not ok:
                rc = 4
        if world == 1 and not a.no_units and not a.timing_only:
            ctx.close()      # the timed context's workspace (82 GB at the default batch) is not needed any more
            if unit0 is None:
                unit0 = structured_unit0(n, D, K, X, T, Y, post, float(doT[0]))
                unit0_name = "structured CPU restatement (oracle.structured_ite) + 1e-10 I, fp64"
            out["units"] = measure_units(gp, synth, np, torch, a, dev, local_rank, X, T, Y, obj, dX, dT, dY,
                                         post0=post, doT0=float(doT[0]), unit0=unit0, unit0_name=unit0_name)
            if not out["units"]["B"].get("parity", {}).get("ok", True):
                rc = 4
            flop_a = float(n) ** 3 / 3.0 + float(n) ** 2 * (3 * (D + K + 1) + 4 + 5 * L)
            ua = {"what": "the headline value: one posterior sample (Gram + potrf + alpha + MeanITE + SATE)",
                  "value": val, "unit": "posterior samples/s", "algorithmic_flop_per_unit": flop_a,
                  "ceiling_units_per_s": FP64_PEAK_TFLOPS * 1e12 / flop_a,
                  "frac_of_ceiling": val * flop_a / (FP64_PEAK_TFLOPS * 1e12)}
            if (n, D, K, L) == (4096, 8, 2, 1) and not a.no_mean_ite and not a.fp32_kernel:
                # fp64 VALU and fp64 MFMA share one datapath on gfx950 (profiles/r02_coexec_f64_microbench.txt): the
                # irreducible fp64 VALU issue time of the Gram build and the MeanITE pass adds to the MFMA time
                gus, gnote = measured_constant("gram_valu_us_n4096")
                ius, inote = measured_constant("ite_mean_valu_us_n4096")
                if gus is None or ius is None:
                    ua["ceiling_shared_datapath_units_per_s"] = None
                    ua["ceiling_shared_datapath_note"] = gnote if gus is None else inote
                else:
                    ua["ceiling_shared_datapath_units_per_s"] = 1.0 / (flop_a / (FP64_PEAK_TFLOPS * 1e12) + (gus + ius) * 1e-6)
                    ua["ceiling_shared_datapath_note"] = (
                        f"1 / (flop / peak + {gus:.1f} us Gram + {ius:.1f} us MeanITE of pure fp64 VALU issue time per sample at "
                        "2.4 GHz: SQ_ACTIVE_INST_VALU x 4 clocks / 1024 SIMDs per 1,024-sample launch, " + gnote + "): the "
                        "MFMA-only ceiling ignores that both instruction classes use the same fp64 units")
            out["units"]["A"] = ua
        if world > 1:
            out["units"] = "N=1 only (units B / C and the unit-A ceilings are single-GPU measurements: run without --gpus)"
            out["configs"] = "N=1 only (BASELINE configs[1] and configs[4] are timed on one GPU: run without --gpus)"
        if world == 1 and not a.no_configs and not a.timing_only:
            ctx.close()
            torch.cuda.empty_cache()
            out["configs"] = {
                "c2": run_config(gp, synth, np, torch, dev, local_rank, 1024, 4, 1, 8192, 1, False, False, 3, 1,
                                 "BASELINE configs[1]: Synthetic N=1024 D=4 nU=1 continuous treatment, fp64, unit A with "
                                 "MeanITE, L=1, 8192 posterior samples per step"),
                "c2_literal": run_config(gp, synth, np, torch, dev, local_rank, 1024, 4, 1, 1000, 1, False, False, 5, 1,
                                         "BASELINE configs[1] AS STATED: Synthetic N=1024 D=4 nU=1 continuous treatment, "
                                         "1k posterior samples = ONE gpslc_predict_dev call per step (S = 1000: one chunk — "
                                         "Gram build, ONE persistent factorisation launch, MeanITE pass, epilogue — with nothing "
                                         "else in flight), fp64, unit A with MeanITE, L=1; kernel launches per call: "
                                         "profiles/r06_c2_literal_kernel_stats.md"),
                "c3_literal": run_config(gp, synth, np, torch, dev, local_rank, 4096, 8, 2, 5000, 1, False, False, 1, 1,
                                         "BASELINE configs[2] AS STATED: Synthetic N=4096 D=8 nU=2, 5k posterior samples = ONE "
                                         "gpslc_predict_dev call per step (S = 5000: five internal chunks of <= 1,024 matrices), "
                                         "fp64, unit A with MeanITE, L=1"),
                "c5": run_config(gp, synth, np, torch, dev, local_rank, 16384, 16, 4, 64, 1, True, True, 1, 1,
                                 "BASELINE configs[4] shape on ONE GPU: Synthetic N=16384 D=16 nU=4 binary treatment, "
                                 "fp32 kernel build + fp64 Cholesky, unit A with MeanITE, doT=1, 64 posterior samples per step",
                                 literal_golden=os.path.join(ROOT, "tests", "golden", "config5_literal.npz")),
            }
            if not a.no_multi_abi:
                # the multi-GPU entry a Julia caller uses (gpslc_predict_multi), everything delivered to host arrays: seconds
                # of compute against seconds of delivery at BASELINE configs[3]'s per-GPU share (tools/bench_multi.py)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_multi
                out["multi_abi"] = {
                    "what": "gpslc_predict / gpslc_predict_multi (devices [0] and [0, 0]: the pool's boxes have ONE GPU, so the "
                            "second row pair prices the entry point, not a speed-up) with every output delivered to HOST arrays, "
                            "best of 2 calls after a warm-up; delivery_s = call - compute-only gpslc_predict_dev; "
                            "round5_staging_restated = the per-shard std::vector staging round 5 used, restated with the same calls",
                    "config4_share": bench_multi.run(4096, 8, 2, 1024, 64, 0, False, True),
                    "draw_tensor": bench_multi.run(4096, 8, 2, 64, 1, 10, True, True)}
            for cfg in out["configs"].values():
                if not cfg["parity"]["ok"]:
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
