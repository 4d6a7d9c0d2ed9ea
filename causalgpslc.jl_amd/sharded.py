"""Multi-GPU ensemble: posterior samples shard over ranks, one collective at the end.

The (posterior sample x intervention level) ensemble is embarrassingly parallel given the replicated
data (X, T, Y: N*(D+2) doubles), so the partition is a static contiguous block of the sample index per
rank — each Cholesky of A is computed exactly once and reused for all levels of that sample — and the
only communication is ONE all_gather of the per-rank (S_r x L) SATE arrays (RCCL over xGMI when the
process group is "nccl"; the reference has no distributed code at all).

One process per GPU (torch.distributed); this module only slices, calls the single-GPU entry point and
gathers.  `compute` is injectable so the N > 1 path can be exercised on CPU (gloo) in the tests with a
stand-in for the HIP entry point; the default is the HIP path and nothing else.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def shard_range(S: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [s0, s1) of the posterior-sample index owned by `rank` (sizes differ by <= 1)."""
    q, r = divmod(S, world)
    s0 = rank * q + min(rank, r)
    return s0, s0 + q + (1 if rank < r else 0)


def slice_object(g, s0: int, s1: int):
    """The GPSLCObject restricted to posterior samples [s0, s1) (data replicated)."""
    from .api import GPSLCObject
    return GPSLCObject(g.X, g.T, g.Y,
                       None if g.U is None else g.U[:, :, s0:s1],
                       None if g.uyLS is None else g.uyLS[:, s0:s1],
                       None if g.xyLS is None else g.xyLS[:, s0:s1],
                       g.tyLS[s0:s1], g.yNoise[s0:s1], g.yScale[s0:s1],
                       hyperparams=g.hyperparams, device=g.device)


def _hip_compute(g_local, doTs):
    from .api import predict
    ms, vs, _ = predict(g_local, doTs)
    return ms, vs


def predict_sharded(g, doTs: Sequence[float], group=None, device=None,
                    compute: Optional[Callable] = None):
    """SATE mean / variance (S x L) for all posterior samples of `g`, computed on `world` ranks.

    Every rank passes the same `g` (same data, same posterior pack) and receives the full result.
    """
    import torch
    import torch.distributed as dist

    compute = compute or _hip_compute
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    S = g.getNumPosteriorSamples()
    L = len(np.atleast_1d(doTs))
    s0, s1 = shard_range(S, world, rank)
    if s1 > s0:
        ms, vs = compute(slice_object(g, s0, s1), doTs)
    else:
        ms = np.zeros((0, L))
        vs = np.zeros((0, L))
    if world == 1:
        return np.asarray(ms), np.asarray(vs)
    # one collective: equal-sized (padded) blocks, [mean | var] packed in a single tensor
    blk = (S + world - 1) // world
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    send = torch.zeros((blk, 2 * L), dtype=torch.float64, device=dev)
    if s1 > s0:
        send[: s1 - s0, :L] = torch.from_numpy(np.ascontiguousarray(ms)).to(dev)
        send[: s1 - s0, L:] = torch.from_numpy(np.ascontiguousarray(vs)).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out_m = np.zeros((S, L))
    out_v = np.zeros((S, L))
    for r in range(world):
        a, b = shard_range(S, world, r)
        blk_r = recv[r].cpu().numpy()
        out_m[a:b] = blk_r[: b - a, :L]
        out_v[a:b] = blk_r[: b - a, L:]
    return out_m, out_v
