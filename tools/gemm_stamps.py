#!/usr/bin/env python3
"""Summarise gpurun_out/gemm_dbg.bin (in-kernel stamps of one tile-kernel launch, measurement build:
GPSLC_GEMM_DBG=<m> for a trailing update with m tile rows, GPSLC_GEMM_DBG_FUSEK=<K> for a fused in-panel launch with a
K-tile loop): per item [entry, after prologue, after K loop, after the stores drained] shader clocks, HW_ID, XCC_ID,
realtime (100 MHz), blockIdx."""
import sys

import numpy as np

d = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_dbg.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] != 0]
st = d[:, :4].astype(np.int64)
full = st[:, 1] != 0          # items that ran a K loop (augmented-row tiles of a fused launch only take the second phase)
pro, loop, tail = (st[:, 1] - st[:, 0])[full], (st[:, 2] - st[:, 1])[full], (st[:, 3] - st[:, 2])[full]
print(f"{len(d)} items ({int(full.sum())} with a K loop): prologue {pro.mean():.0f}  K loop {loop.mean():.0f}  "
      f"second phase / epilogue + stores drained {tail.mean():.0f}  item {(st[:, 3] - st[:, 0])[full].mean():.0f} shader clocks (means)")
# gaps between consecutive items of one workgroup
blk = d[:, 7].astype(np.int64)
gaps, busy, span = [], 0, 0
for b in np.unique(blk):
    s = st[blk == b]
    s = s[np.argsort(s[:, 0])]
    if len(s) > 1:
        gaps += list(s[1:, 0] - s[:-1, 3])
    busy += int((s[:, 3] - s[:, 0]).sum())
    span += int(s[-1, 3] - s[0, 0])
gaps = np.array(gaps)
print(f"gap between consecutive items of a workgroup: mean {gaps.mean():.0f}, median {np.median(gaps):.0f} clocks; "
      f"workgroups busy {busy / span:.3f} of their span; items per workgroup {len(d) / len(np.unique(blk)):.1f}")
rt = d[:, 6].astype(np.int64)
print(f"launch span {(rt.max() - rt.min()) / 100.0:.0f} us (realtime counter)")
# shader clock: s_memtime ticks between the ends of consecutive items of a workgroup against the 100 MHz realtime counter
ghz = []
for b in np.unique(blk):
    m = blk == b
    o = np.argsort(st[m][:, 3])
    e, r = st[m][o, 3], rt[m][o]
    if len(e) > 1 and r[-1] > r[0]:
        ghz.append((e[-1] - e[0]) / ((r[-1] - r[0]) * 10.0))
print(f"s_memtime runs at {np.median(ghz):.3f} GHz against s_memrealtime during this launch (median over workgroups)")
