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
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


sha = blob_sha(os.path.join(ROOT, "causalgpslc.jl_amd", "csrc", "k_tilegemm.hip"))
names = {"kernel_stats.md": "r02_bench_kernel_stats.md", "kernel_stats_unit_b.md": "r02_unit_b_kernel_stats.md",
         "pmc_tile_gemm.md": "r02_pmc_tile_gemm.md", "pmc_fused.md": "r02_pmc_fused_in_panel.md",
         "pmc_draws.md": "r02_pmc_draws.md"}
for a, b in names.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for a, b in {"pmc_tile_gemm.json": "r02_pmc_tile_gemm.json", "pmc_fused.json": "r02_pmc_fused_in_panel.json",
             "pmc_draws.json": "r02_pmc_draws.json"}.items():
    d = json.load(open(os.path.join(src, a)))
    d["kernel_src_sha"] = sha
    d["kernel_src"] = "causalgpslc.jl_amd/csrc/k_tilegemm.hip (git blob hash)"
    d["note"] = ("FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes, tools/profile_r02.sh; "
                 "per-launch means")
    json.dump(d, open(os.path.join(dst, b), "w"), indent=1)
for l in open(os.path.join(src, "trace.log")):
    if l.startswith('{"metric"'):
        open(os.path.join(dst, "r02_bench_under_rocprof.json"), "w").write(l)
print("profiles/ updated from", tag, "kernel sha", sha)
