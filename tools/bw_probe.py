#!/usr/bin/env python3
"""HBM write / read / copy rates of this GPU with plain torch kernels (fill_, sum, copy_) on 8 GiB: the write roof the Gram build
(a write-only stream of the lower tiles) and the read roof the draw kernel sit under.  Usage: python3 tools/bw_probe.py"""
import time
import torch

n = 1 << 30                     # doubles: 8 GiB
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: a.fill_(1.0)); print(f"fill_  (write only) {8 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: a.zero_()); print(f"zero_  (write only) {8 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: a.sum()); print(f"sum    (read only)  {8 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: b.copy_(a)); print(f"copy_  (read+write) {16 * n / t / 1e12:.2f} TB/s of traffic")
t = timed(lambda: torch.add(a, 1.0, out=b)); print(f"add    (read+write) {16 * n / t / 1e12:.2f} TB/s of traffic")
