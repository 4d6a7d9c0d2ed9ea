#!/usr/bin/env python3
"""Summarise gpurun_out/gram_dbg.bin (GPSLC_GRAM_DBG=1, measurement build): per workgroup of the first Gram launch
[entry, features staged, columns done, tile stores drained] shader clocks (thread 0 of the workgroup)."""
import sys

import numpy as np

d = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gram_dbg.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 0] != 0]
st = d[:, :4].astype(np.int64)
rt = d[:, 4:6].astype(np.int64)
ghz = (st[:, 3] - st[:, 0]) / ((rt[:, 1] - rt[:, 0]) * 10.0)          # s_memtime ticks per ns of the 100 MHz realtime counter
print(f"s_memtime runs at {np.median(ghz):.3f} GHz against s_memrealtime (median over workgroups; 10 ns resolution)")
stage, comp, drain = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
span_us = (rt[:, 1].max() - rt[:, 0].min()) / 100.0
print(f"{len(d)} workgroups: staging {stage.mean():.0f} (median {np.median(stage):.0f}), columns {comp.mean():.0f} "
      f"(median {np.median(comp):.0f}), wave 0's stores drained after {drain.mean():.0f} (median {np.median(drain):.0f}) shader clocks")
life = (st[:, 3] - st[:, 0]).mean()
print(f"mean stamped lifetime {life:.0f} clocks = {(rt[:, 1] - rt[:, 0]).mean() / 100.0:.1f} us; launch span {span_us:.0f} us; "
      f"resident workgroups implied = {(rt[:, 1] - rt[:, 0]).sum() / 100.0 / span_us / 256:.2f} per CU")
