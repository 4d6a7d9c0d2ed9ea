#!/usr/bin/env python3
"""Summarise gpurun_out/gram_dbg.bin (GPSLC_GRAM_DBG=1, measurement build): per workgroup of the first Gram launch
[entry, features staged, columns done, tile stores drained] shader clocks (thread 0 of the workgroup)."""
import sys

import numpy as np

d = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gram_dbg.bin", dtype=np.uint64).reshape(-1, 4)
d = d[d[:, 0] != 0]
st = d.astype(np.int64)
stage, comp, drain = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
span = st[:, 3].max() - st[:, 0].min()
print(f"{len(d)} workgroups: staging {stage.mean():.0f} (median {np.median(stage):.0f}), columns {comp.mean():.0f} "
      f"(median {np.median(comp):.0f}), wave 0's stores drained after {drain.mean():.0f} (median {np.median(drain):.0f}) shader clocks")
life = (st[:, 3] - st[:, 0]).mean()
print(f"mean stamped lifetime {life:.0f} clocks; launch span {span} clocks; resident workgroups implied = {len(d) * life / span / 256:.2f} per CU")
