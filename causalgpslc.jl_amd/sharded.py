"""Multi-GPU ensemble: posterior samples shard over ranks, one collective at the end.

The (posterior sample x intervention level) ensemble is embarrassingly parallel given the replicated
data (X, T, Y: N*(D+2) doubles), so the partition is a static contiguous block of the sample index per
rank — each Cholesky of A is computed exactly once and reused for all levels of that sample — and the
only communication is ONE all_gather of the per-rank (S_r x L) SATE arrays (RCCL over xGMI when the
process group is "nccl"; the reference has no distributed code at all).

The LARGE outputs — MeanITE (n x S x L: 17 GB at BASELINE config 4) and the predictive draws
(L x n x S*spp) — are never delivered to every rank by default.  Per output the caller chooses

    "none"   not computed (default)
    "local"  every rank keeps its own block in its HBM (a device tensor); no communication
    "root"   ONE gather to rank `root` ("gather to rank 0 ... optionally meanITE and draws"):
             7 x 2.1 GB over 7 distinct xGMI links at config 4; the other ranks receive nothing
    "all"    all_gather (every rank ends with the whole tensor: world x the traffic and the memory)

and receives device tensors (host NumPy only on request, `to_host=True`).  The SATE summaries are a few MB and
keep the all_gather ("all"), with "root" / "none" available.

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
from dataclasses import dataclass
from typing import Any, Callable, Optional, Sequence, Tuple

import numpy as np

MODES = ("none", "local", "root", "all")


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


@dataclass
class ShardedResult:
    """What one rank holds after `predict_sharded_full`.

    meanSATE, varSATE   (S, L) NumPy — on every rank ("all"), on `root` only ("root"), this rank's (S_r, L) block ("none")
    meanITE             tensor (n, S_r, L) ["local"] / (n, S, L) on root, None elsewhere ["root"] / (n, S, L) ["all"]
    draws               tensor (L, n, S_r*spp) ["local"] / (L, n, S*spp) on root ["root"] / everywhere ["all"] — the
                        reference's level-fastest tensor of predictCounterfactualEffects (src/prediction.jl:30), columns
                        ordered posterior sample outer, draw inner (src/estimation.jl:100-107)
    s0, s1              this rank's block of the posterior-sample index
    """
    meanSATE: Optional[np.ndarray]
    varSATE: Optional[np.ndarray]
    meanITE: Any
    draws: Any
    s0: int
    s1: int
    rank: int
    world: int


def _hip_compute(g_local, doTs, dev_index, L, want_mean_ite, spp, seed, z_local, placement=None):
    """Run the HIP path for this rank's block with every output left in HBM.  Returns device tensors
    (ms (S_r, L), vs (S_r, L), mi (n, S_r, L) | None, draws (L, n, S_r*spp) | None) — views of the library's
    column-major buffers, nothing copied."""
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
    mi = torch.empty(n * Sl * L, dtype=torch.float64, device=dev) if (want_mean_ite or spp > 0) else None
    dr = torch.empty(L * n * Sl * spp, dtype=torch.float64, device=dev) if spp > 0 else None
    dz = to_dev(z_local) if (spp > 0 and z_local is not None) else None
    torch.cuda.synchronize(dev)     # stream contract of the _dev entry points: inputs complete before the call
    if placement is not None:       # (s0, S): this block's place in the whole ensemble, for the library's normals
        ctx.set_ensemble(*placement)
    st = ctx.lib.gpslc_predict_dev(ctx.h, Sl, *[ptr(t) for t in packs], L, ptr(ddo),
                                   float(g_local.hyperparams.predictionCovarianceNoise), int(spp),
                                   int(seed) if dz is None else 0, ptr(dz),
                                   ptr(ms), ptr(vs), ptr(mi), ptr(dr))
    if placement is not None:
        ctx.set_ensemble(0, 0)
    ctx.check(st)
    return (ms.view(L, Sl).t(), vs.view(L, Sl).t(),
            None if not want_mean_ite else mi.view(L, Sl, n).permute(2, 1, 0),
            None if dr is None else dr.view(Sl * spp, n, L).permute(2, 1, 0))


def _global_rank(group, group_rank):
    """`root` is a rank OF THE GROUP everywhere in this module (it is compared with dist.get_rank(group)); torch's
    gather wants the global rank as `dst`."""
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _collect(x_local, full_shape_of, pad_shape, axis, mode, world, rank, root, group, buf_dev, ranges, to_host):
    """One large output: x_local = this rank's block (tensor) with the sharded index along `axis`; "root" = one gather to
    `root`, "all" = all_gather.  Blocks are zero-padded to the common size along `axis` for the collective and the
    receiver keeps each rank's valid part."""
    import torch
    import torch.distributed as dist
    if mode == "none":
        return None
    if mode == "local" or world == 1:
        out = x_local
    else:
        send = torch.zeros(pad_shape, dtype=torch.float64, device=buf_dev)
        if x_local is not None and x_local.shape[axis] > 0:
            send.narrow(axis, 0, x_local.shape[axis]).copy_(x_local)
        if mode == "root":
            recv = [torch.empty_like(send) for _ in range(world)] if rank == root else None
            dist.gather(send, recv, dst=_global_rank(group, root), group=group)
        else:
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send, group=group)
        if recv is None:
            return None
        out = torch.empty(full_shape_of, dtype=torch.float64, device=buf_dev)
        for r, (a, b) in enumerate(ranges):
            if b > a:
                out.narrow(axis, a, b - a).copy_(recv[r].narrow(axis, 0, b - a))
            recv[r] = None          # the padded block is not needed any more: the peak stays near one full tensor + blocks in flight
    if out is not None and to_host:
        out = out.cpu().numpy()
    return out


def predict_sharded_full(g, doTs: Sequence[float], group=None, device=None, compute: Optional[Callable] = None,
                         sate: str = "all", mean_ite: str = "none", draws: str = "none", spp: int = 10,
                         seed: int = 1, z=None, root: int = 0, to_host: bool = False,
                         samples: Optional[Tuple[int, int, int]] = None) -> ShardedResult:
    """The sharded prediction with a gather policy per output (module docstring).

    ``g`` holds either all S posterior samples (every rank passes the same object; the rank's block is sliced
    out) or — with ``samples = (S, s0, s1)`` — only this rank's block [s0, s1) of a pack of S samples.
    Draws: ``z`` = the caller's standard normals (n, spp, S, L) for the WHOLE ensemble (each rank takes its block), or
    None for the library's Philox normals with ``seed``: every rank tells the library where its block sits in the
    ensemble (``gpslc_set_ensemble(s0, S)``), so sample s draws from stream s + S*l whatever the number of ranks — the
    draws of (seed, posterior pack) do not depend on the sharding (round 3 seeded ``seed + rank``: they did).
    ``compute(g_local, doTs)`` -> (ms, vs[, mi]) or, when draws are requested,
    ``compute(g_local, doTs, spp, z_local)`` -> (ms, vs, mi, draws (L, n, S_r*spp)): the CPU stand-in of the tests.
    """
    import torch
    import torch.distributed as dist

    for m in (sate, mean_ite, draws):
        if m not in MODES:
            raise ValueError(f"gather mode must be one of {MODES}, got {m!r}")
    if sate == "local":
        sate = "none"
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
    Sl = s1 - s0
    blk = (S + world - 1) // world
    want_draws = draws != "none"
    want_mi = mean_ite != "none"
    use_hip = compute is None
    backend = dist.get_backend(group) if distributed else None
    dev_index = rank_device(device) if use_hip else None
    comp_dev = torch.device("cuda", dev_index) if use_hip else torch.device("cpu")
    buf_dev = comp_dev if (use_hip and backend != "gloo") else torch.device("cpu")
    if use_hip:
        torch.cuda.set_device(dev_index)
    z_local = None
    if want_draws and z is not None:
        z = np.asarray(z, dtype=np.float64).reshape(n, spp, S, L, order="F")
        z_local = np.asfortranarray(z[:, :, s0:s1, :])

    ms = vs = mi_local = dr_local = None
    if Sl > 0:
        g_local = local if local is not None else slice_object(g, s0, s1, device=dev_index)
        if use_hip:
            if local is not None and local.device != dev_index:
                g_local = slice_object(local, 0, Sl, device=dev_index)
            ms, vs, mi_local, dr_local = _hip_compute(g_local, doTs, dev_index, L, want_mi, spp if want_draws else 0,
                                                      seed, z_local, placement=(s0, S))
        else:
            res = compute(g_local, doTs, spp, z_local) if want_draws else compute(g_local, doTs)
            ms = torch.from_numpy(np.ascontiguousarray(res[0]))
            vs = torch.from_numpy(np.ascontiguousarray(res[1]))
            if want_mi:
                mi_local = torch.from_numpy(np.ascontiguousarray(res[2]))
            if want_draws:
                dr_local = torch.from_numpy(np.ascontiguousarray(res[3]))
    ranges = [shard_range(S, world, r) for r in range(world)]

    # ---- THE collective of the path: [mean | var] blocks (blk x 2L, zero-padded to the common block size)
    send = torch.zeros((blk, 2 * L), dtype=torch.float64, device=comp_dev)
    if Sl > 0:
        send[:Sl, :L] = ms
        send[:Sl, L:] = vs
    out_m = out_v = None
    if sate == "none" or world == 1:
        h = send[:Sl].cpu().numpy()
        out_m, out_v = h[:, :L].copy(), h[:, L:].copy()
    else:
        send = send.to(buf_dev)
        if sate == "all":
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send, group=group)
        else:
            recv = [torch.empty_like(send) for _ in range(world)] if rank == root else None
            dist.gather(send, recv, dst=_global_rank(group, root), group=group)
        if recv is not None:
            allb = torch.stack(recv).cpu().numpy()            # (world, blk, 2L)
            out_m = np.zeros((S, L))
            out_v = np.zeros((S, L))
            for r, (a, b) in enumerate(ranges):
                out_m[a:b] = allb[r, : b - a, :L]
                out_v[a:b] = allb[r, : b - a, L:]

    # ---- the large outputs, each by its own policy (never to every rank unless asked)
    if mi_local is None and want_mi:
        mi_local = torch.zeros((n, 0, L), dtype=torch.float64, device=comp_dev)
    if dr_local is None and want_draws:
        dr_local = torch.zeros((L, n, 0), dtype=torch.float64, device=comp_dev)
    if mi_local is not None and world > 1 and mean_ite in ("root", "all"):
        mi_local = mi_local.to(buf_dev)
    if dr_local is not None and world > 1 and draws in ("root", "all"):
        dr_local = dr_local.to(buf_dev)
    out_i = _collect(mi_local, (n, S, L), (n, blk, L), 1, mean_ite, world, rank, root, group, buf_dev, ranges, to_host)
    dranges = [(a * spp, b * spp) for a, b in ranges]
    out_d = _collect(dr_local, (L, n, S * spp), (L, n, blk * spp), 2, draws, world, rank, root, group, buf_dev, dranges,
                     to_host)
    return ShardedResult(out_m, out_v, out_i, out_d, s0, s1, rank, world)


def predict_sharded(g, doTs: Sequence[float], group=None, device=None, compute: Optional[Callable] = None,
                    gather_mean_ite: bool = False, samples: Optional[Tuple[int, int, int]] = None):
    """SATE mean / variance (S x L) for all posterior samples, computed on `world` ranks; every rank receives
    the (small) full result.  ``gather_mean_ite=True`` additionally all_gathers MeanITE (n, S, L) to EVERY rank as
    host NumPy — world x the traffic of a root gather; kept for callers that want exactly that.  Anything large
    should go through ``predict_sharded_full`` (mean_ite / draws = "local" | "root" | "all", device tensors)."""
    r = predict_sharded_full(g, doTs, group=group, device=device, compute=compute, sate="all",
                             mean_ite="all" if gather_mean_ite else "none", to_host=True, samples=samples)
    if gather_mean_ite:
        return r.meanSATE, r.varSATE, r.meanITE
    return r.meanSATE, r.varSATE


def predict_sharded_pack(path: str, doTs: Sequence[float], group=None, device=None, gather_mean_ite: bool = False,
                         fp32_kernel: bool = False, **full_kwargs):
    """As predict_sharded, from a posterior pack file: every rank reads the header and loads ONLY its own block
    of posterior samples (gpslc_pack_load(path, s0, s1)) — no rank ever holds the whole pack.  With any of
    ``sate= / mean_ite= / draws= / ...`` it returns predict_sharded_full's ShardedResult instead of the tuple."""
    import torch.distributed as dist
    from .pack import loadGPSLCObject, readPackHeader

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    S = int(readPackHeader(path)["S"])
    s0, s1 = shard_range(S, world, rank)
    g_local = loadGPSLCObject(path, device=rank_device(device), samples=(s0, s1), fp32_kernel=fp32_kernel)
    if full_kwargs:
        return predict_sharded_full(g_local, doTs, group=group, device=device, samples=(S, s0, s1), **full_kwargs)
    return predict_sharded(g_local, doTs, group=group, device=device, gather_mean_ite=gather_mean_ite,
                           samples=(S, s0, s1))
