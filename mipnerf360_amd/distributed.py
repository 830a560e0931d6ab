"""Multi-GPU frame rendering: chunk-granular ray sharding + one RCCL all-gather of pixels.

The reference has no distributed code; this is the sharding spec of SURVEY.md
§8e.  Rays are independent EXCEPT for the chunk-global contraction norm
(intern/parameterization.py:23-29 at :75), so the unit of sharding is the reference's own
chunk (model.py:262-264): rank r renders a contiguous block of whole chunks with replicated
weights and no exchange during compute; afterwards the [rays_local, 5] fp32 pixel blocks
(rgb, distance, acc) are collected with ONE all-gather (RCCL over xGMI; 20 B per ray).
The assembled frame is bit-identical to a single-GPU render with the same `chunks`.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def chunk_partition(num_rays: int, chunks: int, world_size: int) -> List[Tuple[int, int]]:
    """[(ray_begin, ray_end)] per rank: contiguous blocks of ceil(n_chunks / world) whole chunks."""
    if num_rays < 0 or chunks < 1 or world_size < 1:
        raise ValueError("chunk_partition: num_rays >= 0, chunks >= 1, world_size >= 1 required")
    n_chunks = (num_rays + chunks - 1) // chunks
    per_rank = (n_chunks + world_size - 1) // world_size
    out = []
    for r in range(world_size):
        b = min(r * per_rank * chunks, num_rays)
        e = min((r + 1) * per_rank * chunks, num_rays)
        out.append((b, e))
    return out


def _host_staged(group) -> bool:
    """gloo cannot all-gather device tensors: stage them through the host (used by the tests that run several ranks of
    the HIP renderer on ONE GPU; with RCCL - the production backend - device tensors go over xGMI directly)."""
    return dist.get_backend(group) == "gloo"


def _all_gather_into(recv: torch.Tensor, send: torch.Tensor, group=None) -> None:
    if send.is_cuda and _host_staged(group):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def _all_reduce_sum(t: torch.Tensor, group=None) -> None:
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def gather_pixels(local: torch.Tensor, spans: List[Tuple[int, int]], group=None) -> torch.Tensor:
    """All-gather the per-rank pixel blocks local[n_r, C] (n_r = spans[rank] length) into [sum n_r, C]
    on every rank.  Blocks are padded to the longest span so a single fixed-size all-gather suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [e - b for b, e in spans]
    if len(spans) != world or local.shape[0] != counts[rank]:
        raise ValueError("gather_pixels: spans do not match the process group / local block")
    width = local.shape[1]
    longest = max(max(counts), 1)
    send = torch.zeros(longest, width, dtype=local.dtype, device=local.device)
    send[:counts[rank]] = local
    recv = torch.empty(world * longest, width, dtype=local.dtype, device=local.device)
    _all_gather_into(recv, send, group)
    recv = recv.view(world, longest, width)
    return torch.cat([recv[r, :counts[r]] for r in range(world)], 0)


def render_rays_sharded(model, rays, chunks: int = 4096, group=None):
    """Each rank renders its block of chunks with `model.render_rays`; returns the full
    (rgb[n,3], distance[n], acc[n]) float tensors on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = rays[0].shape[0]
    spans = chunk_partition(n, chunks, world)
    b, e = spans[rank]
    local_rays = type(rays)(*[f[b:e] for f in rays])
    rgb, dist_, acc = model.render_rays(local_rays, chunks)
    local = torch.cat([rgb, dist_[:, None], acc[:, None]], 1)
    full = gather_pixels(local, spans, group)
    return full[:, :3].contiguous(), full[:, 3].contiguous(), full[:, 4].contiguous()


def forward_sharded(model, rays, group=None):
    """ONE logical batch split over the ranks (BASELINE configs[4]: 8192 rays over 8 GPUs): every rank
    passes ITS slice of the rays and gets its slice of (rgb, distance, acc).  The contraction norm of
    intern/parameterization.py:25 spans the whole batch, so each stage exchanges one fp64 sum of squares
    (all-reduce, 8 bytes) - the path's only data-path collective; results equal the single-device forward of the
    concatenated batch up to the summation order of that one scalar.

    `model` provides the four rank-local pieces: sharded_sample(rays) -> t, sharded_sumsq(rays, t) -> float64[1],
    sharded_prop(rays, t, norm) -> (w, t_new), sharded_nerf(rays, t_new, norm) -> (rgb, distance, acc)
    (mipNeRF360 implements them on the HIP path)."""
    t_hat = model.sharded_sample(rays)
    ss = model.sharded_sumsq(rays, t_hat)
    _all_reduce_sum(ss, group)
    _, t_new = model.sharded_prop(rays, t_hat, ss.sqrt().float())
    ss = model.sharded_sumsq(rays, t_new)
    _all_reduce_sum(ss, group)
    return model.sharded_nerf(rays, t_new, ss.sqrt().float())


def render_image_sharded(model, rays, height: int, width: int, chunks: int = 4096, group=None):
    """Multi-GPU counterpart of mipNeRF360.render_image (model.py:254-274): same outputs."""
    from . import ops
    rgb, dist_, acc = render_rays_sharded(model, rays, chunks, group)
    rgb8 = ops.to8b(rgb).reshape(height, width, 3).cpu().numpy()
    return rgb8, dist_.reshape(height, width).cpu().numpy(), acc.reshape(height, width).cpu().numpy()
