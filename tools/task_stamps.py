#!/usr/bin/env python3
"""Summary of the per-task stamps of one persistent factorisation launch (measurement build, GPSLC_TASK_DBG=<launch #>;
potrf_tasks_kernel writes 8 words per task: fetch start, body start, body end, published, descriptor, workgroup, producers
seen, XCC_ID).  Shader clocks (s_memtime).  usage: task_stamps.py gpurun_out/task_dbg.bin"""
import sys

import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 3] > 0]
desc = d[:, 4]
kind4 = (desc >> np.uint64(30)).astype(int)          # 0 strip, 1 diag, 2 strip + augmented tile, 3 back-substitution
kind = (kind4 == 1).astype(int)
isback = kind4 == 3
k = ((desc >> np.uint64(19)) & np.uint64(31)).astype(int)
i = ((desc >> np.uint64(24)) & np.uint64(63)).astype(int)
nt = int(i[(kind4 == 0) | (kind4 == 2)].max()) if ((kind4 == 0) | (kind4 == 2)).any() else 0
kind = np.where(isback, 7, kind)
t = d[:, :4].astype(np.int64)
ready = d[:, 6].astype(np.int64)
span = t[:, 3].max() - t[:, 0].min()
print(f"{len(d)} tasks, launch span {span / 1e3:.0f} k clocks, {len(np.unique(d[:, 5]))} workgroups, XCDs {sorted(set(int(x) & 15 for x in d[:, 7]))}")
busy = (t[:, 3] - t[:, 0]).sum() / (len(np.unique(d[:, 5])) * span)
print(f"workgroup busy fraction (fetch -> published) {busy:.3f}")
print("| task | count | fetch+wait | of it waiting for producers | acquire+barrier | body | publish |")
print("|---|---:|---:|---:|---:|---:|---:|")
for name, sel in [("back-substitution", isback), ("diag(0)", (kind == 1) & (k == 0)), ("diag(k>0)", (kind == 1) & (k > 0)),
                  ("strip full", (kind == 0) & (i < nt)), ("strip aug", (kind == 0) & (i == nt))] + \
                 [(f"strip k={kk}", (kind == 0) & (i < nt) & (k == kk)) for kk in sorted(set(k[kind == 0]))] + \
                 [(f"diag k={kk}", (kind == 1) & (k == kk)) for kk in sorted(set(k[kind == 1]))]:
    if not sel.any():
        continue
    tt, rr = t[sel], ready[sel]
    wait = np.where(rr > 0, rr - tt[:, 0], 0)
    acq = np.where(rr > 0, tt[:, 1] - rr, tt[:, 1] - tt[:, 0])
    print(f"| {name} | {sel.sum()} | {wait.mean() / 1e3:.1f} k | max {wait.max() / 1e3:.0f} k | {acq.mean() / 1e3:.1f} k | "
          f"{(tt[:, 2] - tt[:, 1]).mean() / 1e3:.1f} k | {(tt[:, 3] - tt[:, 2]).mean() / 1e3:.1f} k |")
tot = (t[:, 3] - t[:, 0]).sum()
print(f"shares of workgroup time: fetch+wait {(np.where(ready > 0, ready - t[:, 0], 0)).sum() / tot:.3f}, "
      f"acquire+barrier {(np.where(ready > 0, t[:, 1] - ready, t[:, 1] - t[:, 0])).sum() / tot:.3f}, "
      f"body {(t[:, 2] - t[:, 1]).sum() / tot:.3f}, publish {(t[:, 3] - t[:, 2]).sum() / tot:.3f}")
