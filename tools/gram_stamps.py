#!/usr/bin/env python3
"""Summarise gpurun_out/gram_dbg.bin (GPSLC_GRAM_DBG=1, measurement build): per workgroup of the first Gram launch
[entry, features staged, columns done] shader clocks + the realtime counter (100 MHz) at the end."""
import sys

import numpy as np

d = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gram_dbg.bin", dtype=np.uint64).reshape(-1, 4)
d = d[d[:, 0] != 0]
st = d[:, :3].astype(np.int64)
stage, comp = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1]
rt = d[:, 3].astype(np.int64)
span_us = (rt.max() - rt.min()) / 100.0
print(f"{len(d)} workgroups: staging {stage.mean():.0f} (median {np.median(stage):.0f}), columns {comp.mean():.0f} "
      f"(median {np.median(comp):.0f}) shader clocks; launch span {span_us:.0f} us")
life = (st[:, 2] - st[:, 0]).mean()
print(f"mean stamped lifetime {life:.0f} clocks; resident workgroups implied = {len(d) * life / 2.4e3 / span_us / 256:.2f} per CU (at 2.4 GHz)")
