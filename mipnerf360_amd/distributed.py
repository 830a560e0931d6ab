"""Multi-GPU frame rendering: chunk-granular ray sharding + one RCCL all-gather of pixels.

The reference has no distributed code; this is the sharding spec of SURVEY.md
§8e.  Rays are independent EXCEPT for the chunk-global contraction norm
(intern/parameterization.py:23-29 at :75), so the unit of sharding is the reference's own
chunk (model.py:262-264): rank r renders a contiguous block of whole chunks with replicated
weights and no exchange during compute; afterwards the per-rank pixel blocks (rgb, distance,
acc: 20 B per ray) are collected with ONE all-gather (RCCL over xGMI).
The assembled frame is bit-identical to a single-GPU render with the same `chunks`.

Buffers: `PixelGather` owns the send / receive blocks of one (num_rays, chunks, world) shape.  The renderer writes
its pixels straight into views of the send block (no concatenation, no padding copy per call), the all-gather lands in
a preallocated receive block, and - because rank r's block of chunks starts at ray r * longest - the assembled outputs
are three strided copies into [n, .] tensors (fresh ones by default, the caller's own with `out=`).  With `slots=2` consecutive frames alternate between two
send / receive pairs so that the all-gather of frame i (on a side stream) overlaps the compute of frame i + 1.
"""
from __future__ import annotations

import inspect
import weakref
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

PIXEL_FLOATS = 5  # rgb (3) + distance + acc


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


def partition_efficiency(num_rays: int, chunks: int, world_size: int) -> float:
    """Upper bound of the strong-scaling efficiency the whole-chunk partition allows: n_chunks / (world * ceil(n_chunks /
    world)) with the last, partial chunk counted by its rays (the busiest rank sets the frame time)."""
    spans = chunk_partition(num_rays, chunks, world_size)
    longest = max(e - b for b, e in spans)
    return 1.0 if longest == 0 else num_rays / (world_size * longest)


def _initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def _world_rank(group=None) -> Tuple[int, int]:
    """(world size, rank); (1, 0) in a process without a process group (single-GPU use of the same code)."""
    if not _initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _host_staged(group) -> bool:
    """gloo cannot all-gather device tensors: stage them through the host (used by the tests that run several ranks of
    the HIP renderer on ONE GPU; with RCCL - the production backend - device tensors go over xGMI directly)."""
    return dist.get_backend(group) == "gloo"


def _all_gather_into(recv: torch.Tensor, send: torch.Tensor, group=None) -> None:
    if not _initialized():  # one process, no group: the "gather" is the local block
        recv.view(-1)[:send.numel()].copy_(send.view(-1))
    elif send.is_cuda and _host_staged(group):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def _all_reduce(t: torch.Tensor, op, group=None) -> None:
    if not _initialized():
        return
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_reduce_sum(t: torch.Tensor, group=None) -> None:
    _all_reduce(t, dist.ReduceOp.SUM, group)


class PixelGather:
    """Preallocated send / receive blocks for the pixel all-gather of one frame shape.

    Layout of a rank's block (`5 * longest` floats, `longest` = rays of the largest span): rgb[longest,3] | distance[longest]
    | acc[longest] - structure of arrays, because that is what the finisher kernel writes (three dense outputs).  The
    receive block is [world, 5 * longest]."""

    def __init__(self, num_rays: int, chunks: int, device, group=None, slots: int = 1, dtype=torch.float32):
        self.group = group
        self.world, self.rank = _world_rank(group)
        self.num_rays, self.chunks = int(num_rays), int(chunks)
        self.spans = chunk_partition(num_rays, chunks, self.world)
        self.counts = [e - b for b, e in self.spans]
        self.longest = max(max(self.counts), 1)
        self.device = torch.device(device)
        L = self.longest
        self.send = [torch.zeros(PIXEL_FLOATS * L, dtype=dtype, device=self.device) for _ in range(slots)]
        self.recv = [torch.empty(self.world, PIXEL_FLOATS * L, dtype=dtype, device=self.device) for _ in range(slots)]
        self.dtype = dtype

    @property
    def span(self) -> Tuple[int, int]:
        return self.spans[self.rank]

    def local_outputs(self, slot: int = 0):
        """(rgb[n_local,3], distance[n_local], acc[n_local]): views of the send block the renderer writes in place."""
        L, c, s = self.longest, self.counts[self.rank], self.send[slot]
        return s[:3 * c].view(c, 3), s[3 * L:3 * L + c], s[4 * L:4 * L + c]

    def gather(self, slot: int = 0) -> None:
        """The path's one exchange step: all-gather of the send block (on the current stream)."""
        _all_gather_into(self.recv[slot].view(-1), self.send[slot], self.group)

    def assemble(self, slot: int = 0, out=None):
        """(rgb[n,3], distance[n], acc[n]) of the whole frame from the received blocks.  Rank r's rays are rows
        [r * longest, r * longest + counts[r]) of the frame (chunk_partition hands out equal blocks of whole chunks).
        `out`: three tensors to fill (a frame loop that recycles its frame buffers); default: fresh tensors, like every
        function of the reference's API returns."""
        L, n, r = self.longest, self.num_rays, self.recv[slot]
        if out is None:
            out = (torch.empty(n, 3, dtype=self.dtype, device=self.device), torch.empty(n, dtype=self.dtype, device=self.device),
                   torch.empty(n, dtype=self.dtype, device=self.device))
        rgb, dist_, acc = out
        if n == 0:
            return rgb, dist_, acc
        nr = (n + L - 1) // L  # ranks that hold rays
        full = n // L          # ranks with a full block
        if full:
            rgb[:full * L].view(full, L, 3).copy_(r[:full, :3 * L].view(full, L, 3))
            dist_[:full * L].view(full, L).copy_(r[:full, 3 * L:4 * L])
            acc[:full * L].view(full, L).copy_(r[:full, 4 * L:5 * L])
        if nr > full:
            c = n - full * L
            rgb[full * L:].copy_(r[full, :3 * c].view(c, 3))
            dist_[full * L:].copy_(r[full, 3 * L:3 * L + c])
            acc[full * L:].copy_(r[full, 4 * L:4 * L + c])
        return rgb, dist_, acc


_GATHERS: "OrderedDict[tuple, PixelGather]" = OrderedDict()
_MAX_GATHERS = 4


def pixel_gather_for(num_rays: int, chunks: int, device, group=None, slots: int = 1) -> PixelGather:
    """Cached PixelGather per frame shape (at most 4 shapes are kept: a renderer alternates between very few)."""
    # the group object itself is part of the key (kept alive by the cache: no id() reuse by a later group)
    key = (int(num_rays), int(chunks), str(torch.device(device)), group, _world_rank(group), int(slots))
    pg = _GATHERS.get(key)
    if pg is None:
        pg = _GATHERS[key] = PixelGather(num_rays, chunks, device, group, slots)
        while len(_GATHERS) > _MAX_GATHERS:
            _GATHERS.popitem(last=False)
    else:
        _GATHERS.move_to_end(key)
    return pg


def gather_pixels(local: torch.Tensor, spans: List[Tuple[int, int]], group=None) -> torch.Tensor:
    """All-gather per-rank blocks local[n_r, C] (n_r = spans[rank] length) of ARBITRARY spans into [sum n_r, C] on every
    rank (general-purpose helper; the frame renderer uses PixelGather, whose buffers are preallocated)."""
    world, rank = _world_rank(group)
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


# ------------------------------------------------------------------------------------------- replica agreement
# model -> (parameter versions, data pointers, fingerprint): a LOCAL cache of a local computation, keyed by the object itself
# (weakly: no id() reuse after garbage collection).  Whether the collective runs never depends on it.
_FP_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_FP_MOD = 65521


def replica_fingerprint(model) -> Optional[torch.Tensor]:
    """int64[2] = (sample counts / MLP precision, hash of all parameter bit patterns) of a torch model, on its device; None
    for renderers without parameters (test doubles).  The hash weights every element with its position inside its tensor
    (1 + index mod 65521) and every tensor with its position in the model, in wrapping int64 arithmetic: permuted or
    compensating values inside one tensor change it, as do swapped layers."""
    if not hasattr(model, "parameters"):
        return None
    params = list(model.parameters())
    if not params:
        return None
    key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params), getattr(model, "mlp_dtype", None),
           getattr(model, "num_samples", None), getattr(model, "num_samples_fine", None))
    try:
        hit = _FP_CACHE.get(model)
    except TypeError:  # an unhashable / non-weakref-able renderer: recompute every time
        hit = None
    if hit is not None and hit[0] == key:
        return hit[1].clone()
    dev = params[0].device
    acc = torch.zeros((), dtype=torch.int64, device=dev)
    for i, p in enumerate(params):
        bits = p.detach().contiguous().view(torch.int32).reshape(-1).to(torch.int64)
        pos = torch.arange(bits.numel(), dtype=torch.int64, device=dev).remainder_(_FP_MOD).add_(1)
        acc = acc + (bits * pos).sum() * (2 * i + 1)
    ns = int(getattr(model, "num_samples", 0)) * 100003 + int(getattr(model, "num_samples_fine", 0) or 0)
    ns = ns * 7 + {"fp32": 0, "bf16": 1, "bf16x3": 2}.get(getattr(model, "mlp_dtype", "fp32"), 3)  # the MLP precision too
    fp = torch.stack([torch.tensor(ns, dtype=torch.int64, device=dev), acc])
    try:
        _FP_CACHE[model] = (key, fp.clone())
    except TypeError:
        pass
    return fp


def check_replicas(model, group=None) -> None:
    """Every rank must render with the same weights and sample counts or the frame is silently inconsistent.  EVERY call on a
    process group of more than one rank issues the same collective on every rank - one all-reduce (MAX) of 4 int64 =
    (f, -f): all ranks agree iff max(f) == -max(-f) - so a rank whose weights changed can never be the only one inside a
    collective (a per-rank cache deciding that would pair its all-reduce with the other ranks' all-gather: a hang or
    corrupted data instead of this error).  Only the local fingerprint computation is cached (per model object and parameter
    versions).  A renderer without parameters contributes zeros."""
    if _world_rank(group)[0] == 1:
        return
    fp = replica_fingerprint(model)
    if fp is None:
        dev = getattr(model, "device", None)
        dev = torch.device(dev) if dev is not None and dist.get_backend(group) == "nccl" else torch.device("cpu")
        fp = torch.zeros(2, dtype=torch.int64, device=dev)
    both = torch.cat([fp, -fp])
    _all_reduce(both, dist.ReduceOp.MAX, group)
    both = both.cpu()
    if not (both[0] == -both[2] and both[1] == -both[3]):
        raise RuntimeError("mipnerf360_amd.distributed: the ranks of this process group hold different weights or sample "
                           "counts (replica fingerprint mismatch) - load the same checkpoint on every rank")


# ------------------------------------------------------------------------------------------- sharded rendering
def render_local_block(model, rays, chunks: int, pg: PixelGather, slot: int = 0, local_rays=None) -> None:
    """This rank's block of chunks rendered into the send block of `pg`.  `local_rays`: the rank's own rays when the
    caller generated only its span (m360_generate_rays_span); otherwise they are sliced from the full `rays`."""
    b, e = pg.span
    if local_rays is None:
        local_rays = type(rays)(*[f[b:e] for f in rays])
    outs = pg.local_outputs(slot)
    if e == b:
        return
    if "out" in inspect.signature(model.render_rays).parameters:
        model.render_rays(local_rays, chunks, out=outs)  # the HIP renderer writes its outputs in place
    else:  # a renderer without `out=` (test doubles): copy
        for dst, src in zip(outs, model.render_rays(local_rays, chunks)):
            dst.copy_(src)


def render_rays_sharded(model, rays, chunks: int = 4096, group=None):
    """Each rank renders its block of chunks with `model.render_rays`; returns the full
    (rgb[n,3], distance[n], acc[n]) float tensors on every rank."""
    check_replicas(model, group)
    n = rays[0].shape[0]
    device = getattr(model, "device", None)
    if device is None or not torch.device(device).type == "cuda":
        device = rays[0].device
    pg = pixel_gather_for(n, chunks, device, group)
    render_local_block(model, rays, chunks, pg)
    pg.gather()
    return pg.assemble()


def forward_sharded(model, rays, group=None):
    """ONE logical batch split over the ranks (BASELINE configs[4]: 8192 rays over 8 GPUs): every rank
    passes ITS slice of the rays and gets its slice of (rgb, distance, acc).  The contraction norm of
    intern/parameterization.py:25 spans the whole batch, so each stage exchanges one fp64 sum of squares
    (all-reduce, 8 bytes) - the path's only data-path collective; results equal the single-device forward of the
    concatenated batch up to the summation order of that one scalar.

    `model` provides the four rank-local pieces: sharded_sample(rays) -> t, sharded_sumsq(rays, t) -> float64[1],
    sharded_prop(rays, t, norm) -> (w, t_new), sharded_nerf(rays, t_new, norm) -> (rgb, distance, acc)
    (mipNeRF360 implements them on the HIP path)."""
    check_replicas(model, group)
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


def render_view_sharded(model, cam_to_world, height: int, width: int, focal: float, near: float, far: float,
                        ndc: bool = False, chunks: int = 4096, group=None):
    """Multi-GPU counterpart of mipNeRF360.render_view: only the pose is shared; every rank generates the rays of ITS
    block of chunks on its own device (m360_generate_rays_span), renders them, and the pixels are all-gathered.
    -> (rgb[n,3], distance[n], acc[n]) device tensors on every rank, n = height * width."""
    from .intern.ray import generate_rays
    check_replicas(model, group)
    dev = torch.device(model.device)
    n = int(height) * int(width)
    pg = pixel_gather_for(n, chunks, dev, group)
    pose = torch.as_tensor(cam_to_world).to(device=dev, dtype=torch.float32)
    local = generate_rays(pose, height, width, focal, near, far, ndc, span=pg.span)
    render_local_block(model, None, chunks, pg, local_rays=local)
    pg.gather()
    return pg.assemble()
