#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_r02.sh run from gpurun_out/<tag>/ into profiles/ (tracked), stamping the PMC
summary of the dominant kernel with the git blob hash of the kernel source it was taken from (bench.py withholds the
`traffic` figure when that hash no longer matches)."""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"          # file-name prefix under profiles/
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


sha = blob_sha(os.path.join(ROOT, "causalgpslc.jl_amd", "csrc", "k_tilegemm.hip"))
names = {"pmc_potrf_tasks.md": f"{rnd}_pmc_potrf_tasks.md", "kernel_stats.md": f"{rnd}_bench_kernel_stats.md", "kernel_stats_panel_schedule.md": f"{rnd}_bench_kernel_stats_panel_schedule.md", "kernel_stats_unit_b.md": f"{rnd}_unit_b_kernel_stats.md",
         "pmc_tile_gemm.md": f"{rnd}_pmc_tile_gemm.md", "pmc_fused.md": f"{rnd}_pmc_fused_in_panel.md",
         "pmc_draws.md": f"{rnd}_pmc_draws.md", "kernel_stats_c2.md": f"{rnd}_n1024_kernel_stats.md",
         "pmc_gram.md": f"{rnd}_pmc_gram.md", "pmc_ite_mean.md": f"{rnd}_pmc_ite_mean.md",
         "kernel_stats_c2_literal.md": f"{rnd}_c2_literal_kernel_stats.md", "pmc_c2_per_kernel.md": f"{rnd}_pmc_n1024_per_kernel.md"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for a, b in {"pmc_potrf_tasks.json": f"{rnd}_pmc_potrf_tasks.json", "pmc_tile_gemm.json": f"{rnd}_pmc_tile_gemm.json", "pmc_fused.json": f"{rnd}_pmc_fused_in_panel.json",
             "pmc_draws.json": f"{rnd}_pmc_draws.json"}.items():
    if not os.path.exists(os.path.join(src, a)):
        continue
    d = json.load(open(os.path.join(src, a)))
    d["kernel_src_sha"] = sha
    d["kernel_src"] = "causalgpslc.jl_amd/csrc/k_tilegemm.hip (git blob hash)"
    d["note"] = ("FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes, "
                 f"tools/profile_{rnd}.sh; per-launch means")
    json.dump(d, open(os.path.join(dst, b), "w"), indent=1)
# ---- measured constants bench.py quotes (round 5): each stamped with the git blob hashes of the sources it was measured on;
# bench.measured_constant withholds a figure whose sources have changed since
CS = os.path.join("causalgpslc.jl_amd", "csrc")


def shas(*files):
    return {os.path.join(CS, f): blob_sha(os.path.join(ROOT, CS, f)) for f in files}


consts_path = os.path.join(dst, f"{rnd}_bench_constants.json")
consts = json.load(open(consts_path)) if os.path.exists(consts_path) else {}
for key, fn, files in (("gram_valu_us_n4096", "pmc_gram.json", ("k_gram.hip", "gp_math.h")),
                       ("ite_mean_valu_us_n4096", "pmc_ite_mean.json", ("k_solve.hip", "gp_math.h"))):
    f = os.path.join(src, fn)
    if os.path.exists(f):
        d = json.load(open(f)).get("SQ_ACTIVE_INST_VALU")
        if d:
            # wave-instruction issue cycles summed over the chip: x 4 clocks per fp64 wave instruction / 1,024 SIMDs / 1,024
            # samples per launch / 2.4 GHz
            us = d["sum"] / d["launches"] * 4.0 / 1024.0 / 1024.0 / 2.4e9 * 1e6
            consts[key] = {"value": us, "source_shas": shas(*files), "from": f"profiles/{rnd}_{fn.replace('.json', '.md')}",
                           "what": "fp64 VALU issue time per posterior sample at N=4096 D=8 nU=2 (SQ_ACTIVE_INST_VALU x 4 / 1024 "
                                   "SIMDs / 1024 samples / 2.4 GHz)"}
c2dir = os.path.join(src, "c2json")
if os.path.isdir(c2dir):
    tot, per = 0.0, {}
    for fn in sorted(os.listdir(c2dir)):
        d = json.load(open(os.path.join(c2dir, fn)))
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            b = (2.0 * d["FETCH_SIZE"]["sum"] + d["WRITE_SIZE"]["sum"]) * 1024.0      # KB -> B, FETCH_SIZE doubled (gfx950)
            per[d["kernel"]] = b
            tot += b
    samples = 4 * 8192          # tools/profile script: 1 warm-up + 3 timed steps of 8,192 posterior samples
    if tot > 0:
        consts["c2_hbm_bytes_per_sample"] = {
            "value": tot / samples, "samples_per_step": 8192,
            "source_shas": shas("k_tilegemm.hip", "k_gram.hip", "k_solve.hip", "k_diag.hip", "diag_block.h", "api.hip"),
            "from": f"profiles/{rnd}_pmc_n1024_per_kernel.md",
            "per_kernel_bytes_per_sample": {k: v / samples for k, v in per.items()},
            "what": "HBM bytes per posterior sample at BASELINE config 2 (N=1024 D=4 nU=1, 8,192 samples per step): FETCH_SIZE x 2 "
                    "+ WRITE_SIZE summed over every kernel of the step"}
if consts:
    json.dump(consts, open(consts_path, "w"), indent=1)
for l in open(os.path.join(src, "trace.log")) if os.path.exists(os.path.join(src, "trace.log")) else []:
    if l.startswith('{"metric"'):
        open(os.path.join(dst, f"{rnd}_bench_under_rocprof.json"), "w").write(l)
print("profiles/ updated from", tag, "kernel sha", sha)
