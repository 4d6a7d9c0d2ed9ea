"""Multi-GPU ensemble: posterior samples shard over ranks, one collective at the end.

The (posterior sample x intervention level) ensemble is embarrassingly parallel given the replicated
data (X, T, Y: N*(D+2) doubles), so the partition is a static contiguous block of the sample index per
rank — each Cholesky of A is computed exactly once and reused for all levels of that sample — and the
only communication is ONE all_gather of the per-rank (S_r x L) SATE arrays (RCCL over xGMI when the
process group is "nccl"; the reference has no distributed code at all), plus, on request,
one more of the per-rank MeanITE block (n x S_r x L).

One process per GPU (torch.distributed).  Each rank factorises on ITS OWN device (``device`` argument, else
LOCAL_RANK, else torch's current device), keeps its block in HBM (``gpslc_predict_dev`` writes straight into
the send buffer) and gathers device tensors; with a gloo group the same buffers go through host memory.
A rank needs only its own block of the posterior pack: ``predict_sharded_pack`` loads exactly that block
(``gpslc_pack_load(path, s0, s1)``).  ``compute`` is injectable so the partition / padding / gather logic
can be exercised on CPU (gloo) in the tests with a stand-in for the HIP entry point; the default is the
HIP path and nothing else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def shard_range(S: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [s0, s1) of the posterior-sample index owned by `rank` (sizes differ by <= 1)."""
    q, r = divmod(S, world)
    s0 = rank * q + min(rank, r)
    return s0, s0 + q + (1 if rank < r else 0)


def slice_object(g, s0: int, s1: int, device: Optional[int] = None):
    """The GPSLCObject restricted to posterior samples [s0, s1) (data replicated), bound to `device`."""
    from .api import GPSLCObject
    return GPSLCObject(g.X, g.T, g.Y,
                       None if g.U is None else g.U[:, :, s0:s1],
                       None if g.uyLS is None else g.uyLS[:, s0:s1],
                       None if g.xyLS is None else g.xyLS[:, s0:s1],
                       g.tyLS[s0:s1], g.yNoise[s0:s1], g.yScale[s0:s1],
                       hyperparams=g.hyperparams, device=g.device if device is None else int(device),
                       fp32_kernel=g.fp32_kernel)


def rank_device(device=None) -> int:
    """The GPU this rank computes on: explicit argument, else LOCAL_RANK (torch.distributed.run), else torch's
    current device."""
    if device is not None:
        return int(getattr(device, "index", device) or 0)
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    import torch
    return int(torch.cuda.current_device())


def _hip_compute_into(g_local, doTs, dev_index, send, L, want_mean_ite):
    """Run the HIP path for this rank's block with every output left in HBM: MeanSATE / VarSATE go straight
    into the first two L-column groups of the send buffer (S_r x L each, sample fastest = the library's
    layout); returns the MeanITE device tensor (n, S_r, L) or None."""
    import torch
    dev = torch.device("cuda", dev_index)
    n, Sl = g_local.getN(), g_local.getNumPosteriorSamples()

    def to_dev(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1, order="F"))).to(dev)

    def ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    ctx = g_local.ctx()     # created on g_local.device == dev_index
    packs = [to_dev(a) for a in (g_local.U, g_local.uyLS, g_local.xyLS, g_local.tyLS, g_local.yScale, g_local.yNoise)]
    ddo = to_dev(np.asarray(doTs, dtype=np.float64))
    ms = torch.empty(Sl * L, dtype=torch.float64, device=dev)
    vs = torch.empty(Sl * L, dtype=torch.float64, device=dev)
    mi = torch.empty(n * Sl * L, dtype=torch.float64, device=dev) if want_mean_ite else None
    torch.cuda.synchronize(dev)     # stream contract of the _dev entry points: inputs complete before the call
    st = ctx.lib.gpslc_predict_dev(ctx.h, Sl, *[ptr(t) for t in packs], L, ptr(ddo),
                                   float(g_local.hyperparams.predictionCovarianceNoise), 0, 0, None,
                                   ptr(ms), ptr(vs), ptr(mi), None)
    ctx.check(st)
    send[:Sl, :L] = ms.view(L, Sl).t()
    send[:Sl, L:2 * L] = vs.view(L, Sl).t()
    return None if mi is None else mi.view(L, Sl, n).permute(2, 1, 0)     # (n, S_r, L) view of the column-major block


def predict_sharded(g, doTs: Sequence[float], group=None, device=None, compute: Optional[Callable] = None,
                    gather_mean_ite: bool = False, samples: Optional[Tuple[int, int, int]] = None):
    """SATE mean / variance (S x L) for all posterior samples, computed on `world` ranks; every rank receives
    the full result (and MeanITE (n, S, L) when ``gather_mean_ite``).

    ``g`` holds either all S posterior samples (every rank passes the same object; the rank's block is sliced
    out) or — with ``samples = (S, s0, s1)`` — only this rank's block [s0, s1) of a pack of S samples.
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    doTs = np.atleast_1d(np.asarray(doTs, dtype=np.float64))
    L = len(doTs)
    if samples is None:
        S = g.getNumPosteriorSamples()
        s0, s1 = shard_range(S, world, rank)
        local = None
    else:
        S, s0, s1 = (int(v) for v in samples)
        if (s0, s1) != shard_range(S, world, rank) or g.getNumPosteriorSamples() != s1 - s0:
            raise ValueError("samples=(S, s0, s1) must be this rank's shard_range block and match g")
        local = g
    n = g.getN()
    blk = (S + world - 1) // world
    use_hip = compute is None
    backend = dist.get_backend(group) if distributed else None
    dev_index = rank_device(device) if use_hip else None
    buf_dev = torch.device("cuda", dev_index) if (use_hip and backend != "gloo") else torch.device("cpu")
    if use_hip:
        torch.cuda.set_device(dev_index)

    # one send block per rank: [mean | var] (blk x 2L), zero-padded to the common block size
    send = torch.zeros((blk, 2 * L), dtype=torch.float64, device=torch.device("cuda", dev_index) if use_hip else "cpu")
    mi_local = None
    if s1 > s0:
        g_local = local if local is not None else slice_object(g, s0, s1, device=dev_index)
        if use_hip:
            if local is not None and local.device != dev_index:
                g_local = slice_object(local, 0, s1 - s0, device=dev_index)
            mi_local = _hip_compute_into(g_local, doTs, dev_index, send, L, gather_mean_ite)
        else:
            res = compute(g_local, doTs)
            send[: s1 - s0, :L] = torch.from_numpy(np.ascontiguousarray(res[0]))
            send[: s1 - s0, L:] = torch.from_numpy(np.ascontiguousarray(res[1]))
            if gather_mean_ite:
                mi_local = torch.from_numpy(np.ascontiguousarray(res[2]))
    if world == 1:
        out = send[:S].cpu().numpy()
        res = (out[:, :L].copy(), out[:, L:].copy())
        if gather_mean_ite:
            res += (np.zeros((n, 0, L)) if mi_local is None else mi_local.cpu().numpy().copy(),)
        return res

    send = send.to(buf_dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)          # THE collective of the path
    allb = torch.stack(recv).cpu().numpy()            # (world, blk, 2L)
    out_m = np.zeros((S, L))
    out_v = np.zeros((S, L))
    for r in range(world):
        a, b = shard_range(S, world, r)
        out_m[a:b] = allb[r, : b - a, :L]
        out_v[a:b] = allb[r, : b - a, L:]
    if not gather_mean_ite:
        return out_m, out_v
    # optional second collective: the MeanITE blocks (n x blk x L per rank, 2.1 GB per rank at config 4)
    smi = torch.zeros((n, blk, L), dtype=torch.float64, device=buf_dev)
    if mi_local is not None:
        smi[:, : s1 - s0, :] = mi_local.to(buf_dev)
    rmi = [torch.empty_like(smi) for _ in range(world)]
    dist.all_gather(rmi, smi, group=group)
    out_i = np.zeros((n, S, L))
    for r in range(world):
        a, b = shard_range(S, world, r)
        out_i[:, a:b, :] = rmi[r][:, : b - a, :].cpu().numpy()
    return out_m, out_v, out_i


def predict_sharded_pack(path: str, doTs: Sequence[float], group=None, device=None, gather_mean_ite: bool = False,
                         fp32_kernel: bool = False):
    """As predict_sharded, from a posterior pack file: every rank reads the header and loads ONLY its own block
    of posterior samples (gpslc_pack_load(path, s0, s1)) — no rank ever holds the whole pack."""
    import torch.distributed as dist
    from .pack import loadGPSLCObject, readPackHeader

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    S = int(readPackHeader(path)["S"])
    s0, s1 = shard_range(S, world, rank)
    g_local = loadGPSLCObject(path, device=rank_device(device), samples=(s0, s1), fp32_kernel=fp32_kernel)
    return predict_sharded(g_local, doTs, group=group, device=device, gather_mean_ite=gather_mean_ite,
                           samples=(S, s0, s1))
