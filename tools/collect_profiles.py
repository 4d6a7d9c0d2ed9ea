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
names = {"kernel_stats.md": f"{rnd}_bench_kernel_stats.md", "kernel_stats_unit_b.md": f"{rnd}_unit_b_kernel_stats.md",
         "pmc_tile_gemm.md": f"{rnd}_pmc_tile_gemm.md", "pmc_fused.md": f"{rnd}_pmc_fused_in_panel.md",
         "pmc_draws.md": f"{rnd}_pmc_draws.md", "kernel_stats_c2.md": f"{rnd}_n1024_kernel_stats.md",
         "pmc_gram.md": f"{rnd}_pmc_gram.md", "pmc_ite_mean.md": f"{rnd}_pmc_ite_mean.md",
         "kernel_stats_c2_literal.md": f"{rnd}_c2_literal_kernel_stats.md", "pmc_c2_per_kernel.md": f"{rnd}_pmc_n1024_per_kernel.md"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for a, b in {"pmc_tile_gemm.json": f"{rnd}_pmc_tile_gemm.json", "pmc_fused.json": f"{rnd}_pmc_fused_in_panel.json",
             "pmc_draws.json": f"{rnd}_pmc_draws.json"}.items():
    if not os.path.exists(os.path.join(src, a)):
        continue
    d = json.load(open(os.path.join(src, a)))
    d["kernel_src_sha"] = sha
    d["kernel_src"] = "causalgpslc.jl_amd/csrc/k_tilegemm.hip (git blob hash)"
    d["note"] = ("FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes, "
                 f"tools/profile_{rnd}.sh; per-launch means")
    json.dump(d, open(os.path.join(dst, b), "w"), indent=1)
for l in open(os.path.join(src, "trace.log")) if os.path.exists(os.path.join(src, "trace.log")) else []:
    if l.startswith('{"metric"'):
        open(os.path.join(dst, f"{rnd}_bench_under_rocprof.json"), "w").write(l)
print("profiles/ updated from", tag, "kernel sha", sha)
