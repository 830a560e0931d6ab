"""Drop-in mirror of the reference's model.py: `prop_net`, `nerf_net`, `mipNeRF360`.

Same constructor arguments, attributes, `state_dict` layout (30 tensors) and
return values as the reference (model.py:14-283) — but `forward` / `render_image` run the
hand-written HIP pipeline of libm360 (sampling -> contraction -> IPE -> fp32-MFMA MLPs ->
resampling -> alpha composite) instead of eager PyTorch ops.

Intentional differences:
  * differentiable with respect to the network PARAMETERS (what train.py:62,80 needs): when autograd is enabled
    and a parameter requires grad, `prop_net.forward` / `nerf_net.forward` run the tape-keeping forward and their
    backward is libm360's hand-written backward (m360_prop_backward / m360_nerf_backward).  Sample positions and
    `coarse_weights` carry no gradient - as in the reference, which resamples under no_grad (intern/ray.py:136);
  * caller tensors are never mutated (the reference's `g()` bumps near/far/t_vals in place);
  * `render_image` is silent unless `self.verbose` and keeps every chunk on the device.
There is no CPU path: tensors must be on a HIP device and libm360.so must be built.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .intern.encoding import PositionalEncoding, ViewdirectionEncoding
from .intern.ray import Rays, namedtuple_map

_WORKSPACES: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
_MAX_WORKSPACES_PER_DEVICE = 3  # streams per device whose scratch is kept (one entry is 4.5 GB at 4096 x 128, full width)

# Default of the `mutate_like_reference` attribute of newly built models (install_dropin(..., mutate_like_reference=True)
# sets it): reproduce the reference's in-place g() side effect on rays.near / rays.far (intern/parameterization.py:15-21).
MUTATE_LIKE_REFERENCE = False
EPS_G = 1e-6


def _kaiming_init(model):
    """model.py:8-12."""
    for module in model.modules():
        if isinstance(module, nn.Linear):
            nn.init.kaiming_uniform_(module.weight)


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): work queued on different streams (a second model rendering on a
    side stream, eval overlapped with training) never shares scratch, and everything launched on one stream is
    ordered by that stream.  A buffer that is outgrown goes back to torch's caching allocator, which is stream-aware
    for the stream it was allocated on - the one it was used on.  At most 3 streams per device keep their buffer (least
    recently used first out): short-lived side streams do not pile up multi-GB entries, and a recycled stream handle
    finds at worst a buffer that was used on a stream with the same handle, i.e. itself."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        _WORKSPACES.pop(key, None)
        ws = None
        with torch.cuda.device(device):
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            ops.workspace_init(ws)  # the status block at its start (counters of the bf16 mode's layer chain)
        _WORKSPACES[key] = ws
        mine = [k for k in _WORKSPACES if k[0] == device]
        for k in mine[:max(0, len(mine) - _MAX_WORKSPACES_PER_DEVICE)]:
            # safe while kernels queued on ITS stream still use it: the caching allocator keeps a freed block in the pool of
            # the stream it was allocated on (= the only stream it was used on) and reuses it in that stream's order
            del _WORKSPACES[k]
    else:
        _WORKSPACES.move_to_end(key)
    return ws


def release_workspaces() -> None:
    """Drop every cached scratch buffer (after all queued work is done: synchronise first)."""
    _WORKSPACES.clear()


class _PackedMLP:
    """Zero-padded, k-contiguous copies of a sub-network's Linear layers for m360_linear,
    rebuilt whenever a parameter tensor is replaced or modified in place."""

    def __init__(self):
        self.key = None
        self.w, self.b = [], []
        self.head_w = self.head_b = None
        self.in_pad = self.h_pad = 0

    @staticmethod
    def _key(params):
        return tuple((p.data_ptr(), p._version, p.device) for p in params)

    def invalidate(self) -> None:
        self.key = None

    def check_nan_pending(self) -> None:
        """Training mode: look at the NaN-parameter flag the last re-pack sent to pinned memory (bf16 modes).  Called by the next re-pack,
        by the backward of the forward that packed (before its gradients are handed out: no update is ever made from a NaN-parameter
        forward) and by `flush_nan_check()`.  The copy was queued before the forward's kernels, so waiting for it stalls nothing."""
        pending = getattr(self, "_nan_pending", None)
        if pending is None:
            return
        self._nan_pending = None
        pending[1].synchronize()
        if int(pending[0][0]) != 0:
            self.key = None
            raise RuntimeError(pending[2])

    def refresh(self, hidden_layers, heads, bf16: int = 0, always: bool = False, defer_nan: bool = False) -> "_PackedMLP":
        """`always`: re-pack even when the (data_ptr, version) key is unchanged.  Writes through `.data`
        (`p.data.mul_()`, EMA swaps, weight clamping) do not bump a tensor's version counter, so in training mode
        - where parameters are expected to change between forwards - the packing is rebuilt on every forward
        (13 small kernels, ~30 MB); in eval mode the key decides, and `invalidate_packed()` forces a rebuild."""
        params = [p for lin in list(hidden_layers) + list(heads) for p in (lin.weight, lin.bias)]
        bf16 = int(bf16)  # 0 = fp32, 1 = bf16, 2 = bf16x3 (two bf16 terms per value, three MFMA passes)
        key = self._key(params) + (bf16,)
        if key == self.key and not always:
            if torch.cuda.current_stream(self.head_w.device) != self.pack_stream:
                torch.cuda.current_stream(self.head_w.device).wait_event(self.ready)  # packed on another stream
            return self
        first = hidden_layers[0]
        ops._require_device(first.weight, "model parameters")
        pad = 64 if bf16 else 32  # the bf16 MFMA K-step is 64 elements
        self.bf16 = bf16
        self.in_pad = ops.round_up(first.in_features, pad)
        self.h_pad = ops.round_up(first.out_features, pad)
        # ONE launch for every packing of the sub-network and (bf16 modes) the NaN scan of the tensors it reads anyway (m360_pack_many; until
        # round 6 a kernel per layer behind a 56 us scan: 0.1 - 0.15 ms of small launches in front of every forward of a training step)
        hidden_fmt = {0: _lib.PACK_F32, 1: _lib.PACK_BF16, 2: _lib.PACK_BF16X3}[bf16]
        # first layers: bf16x3: all 24 bits of features and weights ("x6"); bf16: 16 bits of each (two bf16 terms, three products), one term out
        first_fmt = {0: _lib.PACK_F32, 1: _lib.PACK_BF16X3, 2: _lib.PACK_BF16X6}[bf16]
        items = [(lin.weight, lin.bias, self.h_pad, self.in_pad if i == 0 else self.h_pad, first_fmt if i == 0 else hidden_fmt)
                 for i, lin in enumerate(hidden_layers)]
        n_heads = sum(h.weight.shape[0] for h in heads)
        dev0 = first.weight.device
        head_w = torch.empty(n_heads, self.h_pad, device=dev0)
        head_b = torch.empty(n_heads, device=dev0)
        row = 0
        for h in heads:  # the heads' rows side by side: [n_heads, h_pad] and [n_heads], fp32
            r = h.weight.shape[0]
            items.append((h.weight, h.bias, r, self.h_pad, _lib.PACK_F32, head_w[row:row + r], head_b[row:row + r]))
            row += r
        if bf16:
            self.key = None  # a refused packing must not look current to the next call
        outs, flag = ops.pack_many(items, nan_flag=bool(bf16))
        if bf16:
            # The bf16 matrix pipe answers a NaN operand with the default NaN 0xFFC00000 (sign bit set), which its packed integer-max
            # ReLU reads as a negative number: no NaN survives a ReLU layer.  NaN FEATURES are carried around the MLP by per-sample
            # flags (the finishers poison those samples, as nn.ReLU would have); a NaN PARAMETER cannot be - refuse it loudly.
            # Rendering (the key decides: a re-pack is rare; any forward without a tape counts) waits for the flag here.  A training step
            # (`defer_nan`: the tape-keeping forward in training mode) re-packs on every forward: the flag travels to pinned memory behind
            # an event and is looked at by the NEXT re-pack and by this forward's backward - by then it has long arrived, so a training
            # step never stalls on it and a NaN parameter still never reaches the optimizer.
            msg = ("mlp_dtype='bf16' / 'bf16x3': the parameters hold NaN values; the bf16 matrix pipe cannot propagate them "
                   "the way nn.ReLU does (the reference renders NaN) - use mlp_dtype='fp32' for this checkpoint")
            if always and defer_nan:  # the tape-keeping forward of a training step: it must not stall on the flag
                # (A no-grad forward in training mode - rendering with a model that was never put into eval(), or the other net's forward
                # inside a training step - reads its flag at once below: it has no backward that could still refuse, and a silent render
                # from NaN parameters is worse than one host sync.)
                self.check_nan_pending()  # the previous re-pack's flag: arrived long ago
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev0))
                self._nan_pending = (host, ev, msg)
            elif int(flag.item()) != 0:
                raise RuntimeError(msg)
        self.w = [o[0] for o in outs[:len(hidden_layers)]]
        self.b = [o[1] for o in outs[:len(hidden_layers)]]
        self.head_w = head_w
        self.head_b = head_b
        hw = head_w
        self.key = key
        # a forward on a different stream must not read the packing before the kernels that write it have run
        self.pack_stream = torch.cuda.current_stream(hw.device)
        self.ready = torch.cuda.Event()
        self.ready.record(self.pack_stream)
        return self


def _rays_struct(rays):
    keep = [ops.dev(getattr(rays, f), f"rays.{f}") for f in Rays._fields]
    B = keep[0].shape[0]
    for f, t in zip(Rays._fields, keep):
        if t.shape[0] != B:
            raise RuntimeError(f"rays.{f} has {t.shape[0]} rows, expected {B}")
        if t.device != keep[0].device:
            raise RuntimeError(f"rays.{f} is on {t.device}, rays.{Rays._fields[0]} on {keep[0].device}")
    return _lib.RaysStruct(*[t.data_ptr() for t in keep]), keep, B


def _model_struct(in_ch, prop: Optional[_PackedMLP], nerf: Optional[_PackedMLP], device=None):
    m = _lib.ModelStruct()
    ref = prop or nerf
    if device is not None and ref.head_w.device != device:
        raise RuntimeError(f"model parameters are on {ref.head_w.device}, rays on {device}")
    m.in_ch, m.in_pad = in_ch, ref.in_pad
    m.hp_pad = (prop or nerf).h_pad
    m.hn_pad = (nerf or prop).h_pad
    m.mlp_bf16 = int(getattr(ref, "bf16", 0))
    m.packed_layout = _lib.PACKED_LAYOUT
    if prop is not None and nerf is not None and getattr(prop, "bf16", False) != getattr(nerf, "bf16", False):
        raise RuntimeError("proposal and NeRF networks must use the same MLP precision")
    if prop is not None:
        for i in range(4):
            m.prop_w[i], m.prop_b[i] = prop.w[i].data_ptr(), prop.b[i].data_ptr()
        m.prop_head_w, m.prop_head_b = prop.head_w.data_ptr(), prop.head_b.data_ptr()
    if nerf is not None:
        for i in range(8):
            m.nerf_w[i], m.nerf_b[i] = nerf.w[i].data_ptr(), nerf.b[i].data_ptr()
        m.nerf_head_w, m.nerf_head_b = nerf.head_w.data_ptr(), nerf.head_b.data_ptr()
    return m


def _hyper_struct(num_samples, min_deg, max_deg, white_bkgd=False, density_bias=-1.0, rgb_padding=0.001,
                  resample_padding=0.01, prof=None):
    h = _lib.HyperStruct(int(num_samples), int(min_deg), int(max_deg), int(bool(white_bkgd)), float(density_bias),
                         float(rgb_padding), float(resample_padding))
    h.prof = prof.handle if prof is not None else None  # optional _lib.Prof event recorder (bench.py, tools/)
    ops.apply_tuning(h)  # this thread's per-call A/B switches (libm360 holds none of its own)
    return h


def _set_randomized(hyper, device, jitter: bool, cdf: bool, state=None):
    """randomized=True: the kernels draw their uniforms themselves (m360_hyper_t.randomized / rng_seed / rng_offset) from torch's
    device generator state - no [B, N+1] torch.rand tensors (the reference's intern/ray.py:31,104).  `state`: a (seed, offset) already
    drawn for this forward (the outer forward hands ONE to both stages, so that the staged and the fused path draw the same samples).
    Returns (seed, offset) or None."""
    bits = (1 if jitter else 0) | (2 if cdf else 0)
    hyper.randomized = bits
    if not bits:
        return None
    hyper.rng_seed, hyper.rng_offset = state if state is not None else ops.philox_state(device)
    return hyper.rng_seed, hyper.rng_offset


def _replay_uniforms(module, shape, device):
    """`module.replay_uniforms` (default None): unit uniforms in [0, 1) a randomized stage uses INSTEAD of drawing its own - the
    jitter's `torch.rand` of intern/ray.py:104 ([B, N + 1]) for prop_net, the draw behind `uniform_(to=s - eps)` of intern/ray.py:33 for
    nerf_net.  This is how recorded draws of the reference are replayed through the kernels (fixture G22); ignored unless the module is
    randomized, like the reference ignores its generator then."""
    u = getattr(module, "replay_uniforms", None)
    if u is None or not module.randomized:
        return None
    u = ops.dev(u, "replay_uniforms")
    if tuple(u.shape) != tuple(shape):
        raise RuntimeError(f"replay_uniforms has shape {tuple(u.shape)}, this forward draws {tuple(shape)}")
    return u


def _mutable_field(rays, name):
    """The caller's own tensor of a ray field, for the in-place bumps of mutate_like_reference mode."""
    t = getattr(rays, name)
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"mutate_like_reference: rays.{name} must be a contiguous float32 device tensor (it is mutated "
                           f"in place, like the reference's g() does)")
    return t


def _wants_grad(module: nn.Module) -> bool:
    """Tape-keeping differentiable forward?  The fp32 and (round 5) the bf16 MLP are trainable - in bf16 the tape holds bf16 layer
    outputs, dz travels in bf16, products accumulate in fp32 and the parameter gradients come out in fp32 (fp32 master weights, the
    optimizer is untouched); a model built with mlp_dtype='bf16x3' is forward-only (its outputs never carry a graph)."""
    if getattr(module, "mlp_bf16", 0) == 2 or getattr(module, "mlp_dtype", "fp32") == "bf16x3":
        return False
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


class _TrainCtx:
    """What one tape-keeping stage forward leaves behind for its backward."""

    def __init__(self, module, stage, rays_keep, rstruct, B, N, hyper, packed, mstruct):
        self.module, self.stage, self.rays_keep, self.rstruct, self.B, self.N = module, stage, rays_keep, rstruct, B, N
        self.hyper, self.packed, self.mstruct = hyper, packed, mstruct
        # mstruct holds raw pointers into THIS packing: keep its tensors alive until the backward even if a later
        # forward of the same module re-packs (training mode re-packs on every forward)
        self.packed_keep = (list(packed.w), list(packed.b), packed.head_w, packed.head_b)
        self.h_pad = packed.h_pad
        dev = rays_keep[0].device
        lib = _lib.lib()
        self.tape = torch.empty(max(int(lib.m360_train_tape_bytes(B, N, C.byref(mstruct), stage)), 256), dtype=torch.uint8,
                                device=dev)
        layers, _ = module._layers()
        # transposed packings for the input-gradient GEMMs (layer 0 needs none), made from the same parameter
        # versions as the forward packing (bf16 mode: bf16 transposes for m360_linear_dgrad_bf16)
        self.bf16 = int(getattr(packed, "bf16", 0))
        fmt_t = _lib.PACK_BF16_T if self.bf16 else _lib.PACK_F32_T
        outs_t, _ = ops.pack_many([(lin.weight, None, packed.h_pad, packed.h_pad, fmt_t) for lin in layers[1:]])  # one launch
        self.w_t = [None] + [o[0] for o in outs_t]
        self.in_pad = packed.in_pad
        self.overlap = ops.backward_overlap_wanted()  # decided on the forward's thread, like hyper.tuning
        self.versions = [p._version for p in module.parameters()]

    def backward(self, grad_args):
        if self.tape is None:
            raise RuntimeError("trying to backward through a mipnerf360_amd stage a second time: its tape (saved activations) "
                               "was freed by the first backward")
        module, packed, dev = self.module, self.packed, self.tape.device
        # the event recorder of the FORWARD may have been detached / closed since: use what is attached now, or none
        prof = getattr(module, "prof", None)
        self.hyper.prof = prof.handle if (prof is not None and getattr(prof, "handle", None)) else None
        # (self.hyper carries the per-call switches of the thread that ran the FORWARD: autograd runs this on its own engine thread)
        # bf16 mode: a second stream of our own for the ReLU mask beside the weight gradient (m360_side_t; the library owns none)
        self.hyper.side = ops.side_handle(dev) if (self.bf16 and self.overlap) else None
        if [p._version for p in module.parameters()] != self.versions:
            raise RuntimeError("a parameter was modified in place between the forward and its backward")
        lib = _lib.lib()
        layers, heads = module._layers()
        L, H = len(layers), sum(h.out_features for h in heads)
        # fp32 gradients in the packed [n_pad, k_pad] layouts of the fp32 path, whatever the forward's precision (layer 0: [n_pad, in_pad])
        gw = [torch.empty(self.h_pad, self.in_pad if i == 0 else self.h_pad, device=dev) for i in range(L)]
        gb = [torch.empty(self.h_pad, device=dev) for _ in range(L)]
        ghw, ghb = torch.empty(H, self.h_pad, device=dev), torch.empty(H, device=dev)
        gstruct, tstruct = _lib.MlpGradsStruct(), _lib.MlpTransposedStruct()
        for i in range(L):
            gstruct.w[i], gstruct.b[i] = gw[i].data_ptr(), gb[i].data_ptr()
            tstruct.w_t[i] = ops.ptr(self.w_t[i])
        gstruct.head_w, gstruct.head_b = ghw.data_ptr(), ghb.data_ptr()
        ws = _workspace(max(int(lib.m360_backward_workspace_bytes(self.B, self.N, C.byref(self.mstruct), self.stage)), 256),
                        dev)
        grad_args = [None if g is None else ops.dev(g, "gradient") for g in grad_args]
        if self.stage == 0:
            (g_w_hat,) = grad_args
            if g_w_hat is None:
                g_w_hat = torch.zeros(self.B, self.N, device=dev)
            ops.call("m360_prop_backward", C.byref(self.rstruct), C.byref(self.mstruct), C.byref(tstruct),
                     C.byref(self.hyper), self.B, self.tape, self.tape.numel(), g_w_hat, C.byref(gstruct), ws,
                     ws.numel(), ops.STREAM, device=dev)
        else:
            g_rgb, g_dist, g_acc, g_w = grad_args
            ops.call("m360_nerf_backward", C.byref(self.rstruct), C.byref(self.mstruct), C.byref(tstruct),
                     C.byref(self.hyper), self.B, self.tape, self.tape.numel(), g_rgb, g_dist, g_acc, g_w,
                     C.byref(gstruct), ws, ws.numel(), ops.STREAM, device=dev)
        grads = []
        for i, lin in enumerate(layers):
            grads += [gw[i][:lin.out_features, :lin.in_features], gb[i][:lin.out_features]]
        row = 0
        for h in heads:
            grads += [ghw[row:row + h.out_features, :h.in_features], ghb[row:row + h.out_features]]
            row += h.out_features
        self.tape = self.packed_keep = None  # one backward per forward, like autograd's freed buffers
        packed.check_nan_pending()  # bf16 modes: this forward's parameters held a NaN -> raise instead of handing gradients to the optimizer
        return [g.contiguous() for g in grads]


class _PropTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, rays, *params):
        t_hat, w_hat, ctx.train = module._forward_impl(rays, train=True)
        ctx.mark_non_differentiable(t_hat)
        return t_hat, w_hat

    @staticmethod
    def backward(ctx, _g_t, g_w):
        return (None, None, *ctx.train.backward([g_w]))


class _NerfTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, rays, t_vals, coarse_weights, *params):
        outs, ctx.train = module._forward_impl(rays, t_vals, coarse_weights, train=True)
        ctx.mark_non_differentiable(outs[3], outs[5])
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_dist, g_acc, _g_t, g_w, _g_s):
        return (None, None, None, None, *ctx.train.backward([g_rgb, g_dist, g_acc, g_w]))


def _ws_for(B, N, mstruct, device):
    nbytes = _lib.lib().m360_forward_workspace_bytes(B, N, C.byref(mstruct))
    return _workspace(max(int(nbytes), 256), device)


class prop_net(nn.Module):
    """Proposal network, model.py:14-94 of the reference (4 x hidden_proposal, density only)."""

    def __init__(self, randomized=False, num_samples=128, hidden_proposal=256, density_bias=-1, viewdir_min_deg=0,
                 viewdir_max_deg=4, device=torch.device("cuda")):
        super().__init__()
        self.randomized = randomized
        self.num_samples = num_samples
        self.hidden_proposal = hidden_proposal
        self.density_bias = density_bias
        self.viewdir_min_deg = viewdir_min_deg
        self.viewdir_max_deg = viewdir_max_deg
        self.device = device
        self.positional_encoding = PositionalEncoding()
        self.viewdirs_encoding = ViewdirectionEncoding(self.viewdir_min_deg, self.viewdir_max_deg)
        self.input_size = 21 * 2 + (self.viewdir_max_deg - self.viewdir_min_deg) * 2 * 2
        self.density_activation = nn.Softplus()
        h = self.hidden_proposal
        self.model = nn.Sequential(
            nn.Linear(self.input_size, h), nn.ReLU(True),
            nn.Linear(h, h), nn.ReLU(True),
            nn.Linear(h, h), nn.ReLU(True),
            nn.Linear(h, h), nn.Sigmoid(),
            nn.Linear(h, 1))
        _kaiming_init(self)
        self.to(device)
        self._packed = _PackedMLP()
        self.mutate_like_reference = MUTATE_LIKE_REFERENCE
        self.replay_uniforms = None  # unit uniforms to use instead of drawing (see _replay_uniforms)

    def _pack(self, defer_nan: bool = False) -> _PackedMLP:
        return self._packed.refresh([self.model[i] for i in (0, 2, 4, 6)], [self.model[8]],
                                    getattr(self, "mlp_bf16", False), always=self.training, defer_nan=defer_nan)

    def invalidate_packed(self) -> None:
        """Forget the packed copy of the weights: call after changing parameters through `.data` in eval mode."""
        self._packed.invalidate()

    def density_to_weight(self, t_vals, density, dirs):
        """model.py:59-78."""
        return ops.density_to_weight(t_vals, density, dirs)

    def _layers(self):
        return [self.model[i] for i in (0, 2, 4, 6)], [self.model[8]]

    def forward(self, rays):
        """model.py:80-94 -> (t_vals[B,N+1], weights[B,N]).  With autograd enabled and trainable parameters the
        weights carry the graph to the parameters (train.py:55-62)."""
        if self.mutate_like_reference:
            near, far = _mutable_field(rays, "near"), _mutable_field(rays, "far")
        out = _PropTrainFn.apply(self, rays, *self.parameters()) if _wants_grad(self) else self._forward_impl(rays)
        if self.mutate_like_reference:  # sample_along_rays: g(far), g(near) (intern/ray.py:100) bump the caller's tensors
            with torch.no_grad():
                far.add_(EPS_G)
                near.add_(EPS_G)
        return out

    def _forward_impl(self, rays, train=False):
        rstruct, keep, B = _rays_struct(rays)
        dev = keep[0].device
        N = self.num_samples
        packed = self._pack(defer_nan=train)
        mstruct = _model_struct(self.input_size, packed, None, dev)
        hyper = _hyper_struct(N, self.viewdir_min_deg, self.viewdir_max_deg, density_bias=self.density_bias,
                              prof=getattr(self, "prof", None))
        t_hat = torch.empty(B, N + 1, device=dev)
        w_hat = torch.empty(B, N, device=dev)
        # randomized: drawn inside the kernels (stage_prologue_kernel / sample_t_kernel) unless uniforms are handed in for replay
        t_rand = _replay_uniforms(self, (B, N + 1), dev)
        self.last_rng = _set_randomized(hyper, dev, bool(self.randomized) and t_rand is None, False, getattr(self, "_rng_state", None))
        ws = _ws_for(B, N, mstruct, dev)
        if train:
            tc = _TrainCtx(self, 0, keep, rstruct, B, N, hyper, packed, mstruct)
            ops.call("m360_prop_forward_train", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_rand, t_hat,
                     w_hat, tc.tape, tc.tape.numel(), ws, ws.numel(), ops.STREAM, device=dev)
            return t_hat, w_hat, tc
        ops.call("m360_prop_forward", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_rand, t_hat, w_hat, ws,
                 ws.numel(), ops.STREAM, device=dev)
        return t_hat, w_hat


class nerf_net(nn.Module):
    """NeRF network, model.py:96-200 of the reference (8 x hidden_nerf, density + colour heads)."""

    def __init__(self, randomized=False, num_samples=128, hidden_nerf=1024, density_bias=-1, rgb_padding=0.001,
                 resample_padding=0.01, white_bkgd=False, viewdir_min_deg=0, viewdir_max_deg=4,
                 device=torch.device("cuda")):
        super().__init__()
        self.randomized = randomized
        self.num_samples = num_samples
        self.hidden_nerf = hidden_nerf
        self.density_bias = density_bias
        self.rgb_padding = rgb_padding
        self.resample_padding = resample_padding
        self.white_bkgd = white_bkgd
        self.viewdir_min_deg = viewdir_min_deg
        self.viewdir_max_deg = viewdir_max_deg
        self.device = device
        self.positional_encoding = PositionalEncoding()
        self.viewdirs_encoding = ViewdirectionEncoding(self.viewdir_min_deg, self.viewdir_max_deg)
        self.input_size = 21 * 2 + (self.viewdir_max_deg - self.viewdir_min_deg) * 2 * 2
        self.density_activation = nn.Softplus()
        h = self.hidden_nerf
        layers = [nn.Linear(self.input_size, h), nn.ReLU(True)]
        for _ in range(6):
            layers += [nn.Linear(h, h), nn.ReLU(True)]
        layers += [nn.Linear(h, h), nn.Sigmoid()]
        self.model = nn.Sequential(*layers)
        self.final_density = nn.Sequential(nn.Linear(h, 1), nn.Sigmoid())
        self.final_color = nn.Sequential(nn.Linear(h, 3), nn.Sigmoid())
        _kaiming_init(self)
        self.to(device)
        self._packed = _PackedMLP()
        self.mutate_like_reference = MUTATE_LIKE_REFERENCE
        self.replay_uniforms = None  # unit uniforms to use instead of drawing (see _replay_uniforms)

    def _pack(self, defer_nan: bool = False) -> _PackedMLP:
        return self._packed.refresh([self.model[i] for i in range(0, 16, 2)],
                                    [self.final_density[0], self.final_color[0]], getattr(self, "mlp_bf16", False),
                                    always=self.training, defer_nan=defer_nan)

    def invalidate_packed(self) -> None:
        """Forget the packed copy of the weights: call after changing parameters through `.data` in eval mode."""
        self._packed.invalidate()

    def _hyper(self, N, n_fine=0):
        h = _hyper_struct(N, self.viewdir_min_deg, self.viewdir_max_deg, self.white_bkgd, self.density_bias,
                          self.rgb_padding, self.resample_padding, prof=getattr(self, "prof", None))
        h.num_samples_fine = int(n_fine or 0)
        h.rays_mutated = int(bool(self.mutate_like_reference))
        return h

    def _stash(self, outs):
        # model.py:192-196: kept on the module for the distillation / regularisation losses
        self.fine_weights, self.t_vals, self.s_vals = outs["fine_w"], outs["t_vals"], outs["s_vals"]

    def _layers(self):
        return [self.model[i] for i in range(0, 16, 2)], [self.final_density[0], self.final_color[0]]

    def forward(self, rays, t_vals, coarse_weights):
        """model.py:163-200 -> (rgb[B,3], distance[B], acc[B], t_vals[B,N+1], fine_weights[B,N], s_vals[B,N+1]).
        The number of fine samples is t_vals.shape[-1]-1, as in the reference (intern/ray.py:147).  With autograd
        enabled and trainable parameters rgb / distance / acc / fine_weights carry the graph to the parameters
        (train.py:72-80); t_vals and coarse_weights are constants, as under the reference's no_grad resampling."""
        if self.mutate_like_reference:
            near, far = _mutable_field(rays, "near"), _mutable_field(rays, "far")
        if _wants_grad(self):
            outs = _NerfTrainFn.apply(self, rays, t_vals.detach(), coarse_weights.detach(), *self.parameters())
            self.fine_weights, self.t_vals, self.s_vals = outs[4], outs[3], outs[5]
        else:
            outs = self._forward_impl(rays, t_vals, coarse_weights)
        if self.mutate_like_reference:  # t_to_s (model.py:196): g(near), g(far), g(near) on the caller's tensors
            with torch.no_grad():
                near.add_(EPS_G)
                far.add_(EPS_G)
                near.add_(EPS_G)
        return outs

    def _forward_impl(self, rays, t_vals, coarse_weights, train=False):
        rstruct, keep, B = _rays_struct(rays)
        dev = keep[0].device
        t_vals, coarse_weights = ops.dev(t_vals, "t_vals"), ops.dev(coarse_weights, "coarse_weights")
        N = t_vals.shape[-1] - 1
        Nf = getattr(self, "num_samples_fine", None) or N  # extension; None = the reference's behaviour
        mstruct = _model_struct(self.input_size, None, self._pack(defer_nan=train), dev)
        hyper = self._hyper(N, Nf)
        outs = _alloc_outputs(B, Nf, dev, with_prop=False)
        ostruct = _outputs_struct(outs)
        # randomized: drawn inside the resample kernel unless uniforms are handed in for replay
        u_rand = _replay_uniforms(self, (B, Nf + 1), dev)
        self.last_rng = _set_randomized(hyper, dev, False, bool(self.randomized) and u_rand is None, getattr(self, "_rng_state", None))
        ws = _ws_for(B, max(N, Nf), mstruct, dev)
        if train:
            tc = _TrainCtx(self, 1, keep, rstruct, B, Nf, hyper, self._packed, mstruct)
            ops.call("m360_nerf_forward_train", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_vals,
                     coarse_weights, u_rand, C.byref(ostruct), tc.tape, tc.tape.numel(), ws, ws.numel(), ops.STREAM,
                     device=dev)
            return (outs["rgb"], outs["distance"], outs["acc"], outs["t_vals"], outs["fine_w"], outs["s_vals"]), tc
        ops.call("m360_nerf_forward", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_vals, coarse_weights,
                 u_rand, C.byref(ostruct), ws, ws.numel(), ops.STREAM, device=dev)
        self._stash(outs)
        return outs["rgb"], outs["distance"], outs["acc"], self.t_vals, self.fine_weights, self.s_vals


def _alloc_outputs(B, N, dev, with_prop=True, rgb=None, distance=None, acc=None):
    o = dict(rgb=rgb if rgb is not None else torch.empty(B, 3, device=dev),
             distance=distance if distance is not None else torch.empty(B, device=dev),
             acc=acc if acc is not None else torch.empty(B, device=dev),
             t_vals=torch.empty(B, N + 1, device=dev), fine_w=torch.empty(B, N, device=dev),
             s_vals=torch.empty(B, N + 1, device=dev))
    if with_prop:
        o["t_hat"] = torch.empty(B, N + 1, device=dev)
        o["w_hat"] = torch.empty(B, N, device=dev)
    return o


def _outputs_struct(o):
    return _lib.OutputsStruct(*[ops.ptr(o.get(k)) for k in ("rgb", "distance", "acc", "t_hat", "w_hat", "t_vals",
                                                            "fine_w", "s_vals")])


class mipNeRF360(nn.Module):
    """model.py:202-283 of the reference."""

    def __init__(self, randomized=False, num_samples=128, hidden_proposal=256, hidden_nerf=1024, density_bias=-1,
                 rgb_padding=0.001, resample_padding=0.01, white_bkgd=False, viewdir_min_deg=0, viewdir_max_deg=4,
                 device=torch.device("cuda"), num_samples_fine=None, mlp_dtype="fp32"):
        """Same arguments as the reference (model.py:203-215), plus two keyword extensions at the end:
        `num_samples_fine`: number of NeRF-stage samples per ray when it should differ from the proposal count
        ("64+128" rendering, BASELINE configs[2]); None keeps the reference's behaviour (equal counts);
        `mlp_dtype`: "fp32" (default: exact-fp32 MFMA, the parity path), "bf16" (BASELINE configs[4]: bf16
        weights / hidden activations, fp32 accumulation, heads and ray math in fp32) or "bf16x3" (every
        value carried as two bf16 terms, products formed as xh wh + xl wh + xh wl on the bf16 MFMA with fp32
        accumulation: within the fp32 render tolerance of 1e-4 at about 3x the fp32 rays/s; forward only).  In both
        reduced-precision modes the FIRST layers (model.py:44,132) see more bits of the encoded features and of their weights
        than the hidden layers - bf16: 16 (two bf16 terms, three products), bf16x3: all 24 (three terms, six products:
        include/m360.h "x6") - because positions inside a contracted chunk differ in their low-order bits only."""
        super().__init__()
        self.randomized = randomized
        self.num_samples = num_samples
        self.num_samples_fine = num_samples_fine
        self.hidden_proposal = hidden_proposal
        self.hidden_nerf = hidden_nerf
        self.density_bias = density_bias
        self.rgb_padding = rgb_padding
        self.resample_padding = resample_padding
        self.white_bkgd = white_bkgd
        self.viewdir_min_deg = viewdir_min_deg
        self.viewdir_max_deg = viewdir_max_deg
        self.device = device
        self.init_randomized = randomized
        self.verbose = False
        self.prof = None  # optional _lib.Prof event recorder: set through `set_prof()`
        self.mutate_like_reference = MUTATE_LIKE_REFERENCE
        self.super_batch_rays = 4096  # render_rays launches this many rays at once when `chunks` is smaller
        self.prop_net = prop_net(randomized=self.randomized, num_samples=self.num_samples,
                                 hidden_proposal=self.hidden_proposal, density_bias=self.density_bias,
                                 viewdir_min_deg=self.viewdir_min_deg, viewdir_max_deg=self.viewdir_max_deg,
                                 device=self.device)
        self.nerf_net = nerf_net(randomized=self.randomized, num_samples=self.num_samples, hidden_nerf=self.hidden_nerf,
                                 density_bias=self.density_bias, rgb_padding=self.rgb_padding,
                                 resample_padding=self.resample_padding, white_bkgd=self.white_bkgd,
                                 viewdir_min_deg=self.viewdir_min_deg, viewdir_max_deg=self.viewdir_max_deg,
                                 device=self.device)
        self.nerf_net.num_samples_fine = num_samples_fine
        if mlp_dtype not in ("fp32", "bf16", "bf16x3", torch.float32, torch.bfloat16):
            raise ValueError(f"mlp_dtype must be 'fp32', 'bf16' or 'bf16x3', got {mlp_dtype!r}")
        self.mlp_dtype = "bf16" if mlp_dtype in ("bf16", torch.bfloat16) else ("bf16x3" if mlp_dtype == "bf16x3" else "fp32")
        self.prop_net.mlp_bf16 = self.nerf_net.mlp_bf16 = {"fp32": 0, "bf16": 1, "bf16x3": 2}[self.mlp_dtype]
        self.to(device)

    def set_prof(self, prof) -> None:
        """Attach (or with None detach) a `_lib.Prof` event recorder: the stage drivers then record one HIP-event pair
        per kernel on the launch stream (measurement only; bench.py's roofline)."""
        self.prof = self.prop_net.prof = self.nerf_net.prof = prof

    def set_mutate_like_reference(self, on: bool = True) -> None:
        """Opt in to the reference's side effect: every prop_net.forward adds 1e-6 to the caller's rays.near / rays.far in
        place and every nerf_net.forward 2e-6 / 1e-6 more (g() of intern/parameterization.py:15-21), so the three forward
        pairs of one train.py iteration (train.py:51-71) see the same drifting near / far as in the reference."""
        self.mutate_like_reference = self.prop_net.mutate_like_reference = self.nerf_net.mutate_like_reference = bool(on)

    def invalidate_packed(self) -> None:
        """Forget the packed copies of the weights.  Needed only after changing parameters through `.data` (no version
        bump) while in eval mode; in training mode every forward re-packs."""
        self.prop_net.invalidate_packed()
        self.nerf_net.invalidate_packed()

    def flush_nan_check(self) -> None:
        """bf16 modes, training: raise now if the parameters of the LAST forward of either sub-net held a NaN (the flag travels to pinned
        memory beside each training-mode forward and is otherwise looked at by the next forward or by the backward).  For the end of a
        training loop, or before a checkpoint is written."""
        self.prop_net._packed.check_nan_pending()
        self.nerf_net._packed.check_nan_pending()

    # ------------------------------------------------------------------ fused two-stage forward
    def _forward_fused(self, rays, rgb=None, distance=None, acc=None, stash=True, norm_group_rays=0):
        rstruct, keep, B = _rays_struct(rays)
        dev = keep[0].device
        N = self.prop_net.num_samples
        Nf = self.nerf_net.num_samples_fine or N
        mstruct = _model_struct(self.prop_net.input_size, self.prop_net._pack(), self.nerf_net._pack(), dev)
        hyper = self.nerf_net._hyper(N, Nf)
        hyper.norm_group_rays = int(norm_group_rays)
        self.last_rng = _set_randomized(hyper, dev, bool(self.prop_net.randomized), bool(self.nerf_net.randomized))
        if stash:
            outs = _alloc_outputs(B, Nf, dev, with_prop=False, rgb=rgb, distance=distance, acc=acc)
        else:
            outs = dict(rgb=rgb if rgb is not None else torch.empty(B, 3, device=dev),
                        distance=distance if distance is not None else torch.empty(B, device=dev),
                        acc=acc if acc is not None else torch.empty(B, device=dev))
        ostruct = _outputs_struct(outs)
        ws = _ws_for(B, max(N, Nf), mstruct, dev)
        ops.call("m360_forward", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, C.byref(ostruct), ws,
                 ws.numel(), ops.STREAM, device=dev)
        if stash:
            self.nerf_net._stash(outs)
        return outs["rgb"], outs["distance"], outs["acc"]

    def chain_status(self) -> dict:
        """Counters of the bf16 mode's hidden-layer chain on this model's device and the CURRENT stream's scratch buffer since it was
        allocated (ops.workspace_status; waits for the stream): `launches`, `recoveries` (launches whose self-check failed and that the
        gated layer-by-layer re-run repaired on the device), `timeouts`, `xcc_mismatch`, `last_error`.  Purely informational - outputs are
        right either way; `recoveries > 0` means the chain's launches were not alone on the GPU (or the device places workgroups
        differently) and cost time.  All zero when no forward has run the chain here."""
        dev = torch.device(self.device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        ws = _WORKSPACES.get((dev, torch.cuda.current_stream(dev).cuda_stream))
        if ws is None:
            return dict.fromkeys(ops.CHAIN_STATUS_FIELDS, 0)
        return ops.workspace_status(ws)

    def chain_error(self) -> bool:
        """True when a chain launch of the last forward on this stream reported an error (and was repaired by the gated re-run)."""
        return self.chain_status()["last_error"] != 0

    def forward(self, rays):
        """model.py:247-252 -> (rgb[B,3], distance[B], acc[B])."""
        staged = self.prop_net.mutate_like_reference or self.nerf_net.mutate_like_reference
        # recorded draws to replay (fixture G22): the staged entry points take them as t_rand / u_rand, m360_forward has no such argument
        staged = staged or getattr(self.prop_net, "replay_uniforms", None) is not None or getattr(self.nerf_net, "replay_uniforms", None) is not None
        if not _wants_grad(self) and not staged:  # randomized or not: the kernels draw their own uniforms (round 5)
            return self._forward_fused(rays)
        # ONE generator state for both stages, like the fused path (streams 0 / 1 of the same counter block): the same seed gives the same
        # samples whichever path a forward takes
        draws = (bool(self.prop_net.randomized) and self.prop_net.replay_uniforms is None) or (bool(self.nerf_net.randomized) and self.nerf_net.replay_uniforms is None)
        state = ops.philox_state(_rays_struct(rays)[1][0].device) if draws else None
        self.prop_net._rng_state = self.nerf_net._rng_state = state
        try:
            t_hat, w_hat = self.prop_net.forward(rays)
            rgb, dist, acc, _, _, _ = self.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        finally:
            self.prop_net._rng_state = self.nerf_net._rng_state = None
        self.last_rng = state
        return rgb, dist, acc

    # ------------------------------------------------------------------ one batch sharded over ranks
    def _sharded_common(self, rays):
        rstruct, keep, B = _rays_struct(rays)
        N = self.prop_net.num_samples
        Nf = self.nerf_net.num_samples_fine or N
        mstruct = _model_struct(self.prop_net.input_size, self.prop_net._pack(), self.nerf_net._pack(), keep[0].device)
        hyper = self.nerf_net._hyper(N, Nf)
        return rstruct, keep, B, N, Nf, mstruct, hyper, _ws_for(B, max(N, Nf), mstruct, keep[0].device)

    def sharded_sample(self, rays):
        """this rank's proposal sample positions t_hat[B,N+1] (deterministic sampling; intern/ray.py:99-110)"""
        return ops.sample_t(rays.near, rays.far, self.prop_net.num_samples)

    def sharded_sumsq(self, rays, t_vals):
        return ops.mean_sumsq(t_vals, rays.directions, rays.radii)

    def sharded_prop(self, rays, t_hat, norm):
        """proposal stage with the batch-global contraction norm `norm` (device float[1]) -> (w_hat, t_new)"""
        rstruct, keep, B, N, Nf, mstruct, hyper, ws = self._sharded_common(rays)
        t_hat, norm = ops.dev(t_hat, "t_hat"), ops.dev(norm, "norm")
        dev = keep[0].device
        w_hat, t_new = torch.empty(B, N, device=dev), torch.empty(B, Nf + 1, device=dev)
        with torch.no_grad():
            ops.call("m360_prop_forward_from_t", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_hat, norm,
                     w_hat, t_new, ws, ws.numel(), ops.STREAM, device=dev)
        return w_hat, t_new

    def sharded_nerf(self, rays, t_new, norm):
        """NeRF stage with the batch-global contraction norm -> (rgb[B,3], distance[B], acc[B])"""
        rstruct, keep, B, N, Nf, mstruct, hyper, ws = self._sharded_common(rays)
        t_new, norm = ops.dev(t_new, "t_new"), ops.dev(norm, "norm")
        dev = keep[0].device
        outs = dict(rgb=torch.empty(B, 3, device=dev), distance=torch.empty(B, device=dev), acc=torch.empty(B, device=dev))
        ostruct = _outputs_struct(outs)
        with torch.no_grad():
            ops.call("m360_nerf_forward_from_t", C.byref(rstruct), C.byref(mstruct), C.byref(hyper), B, t_new, norm,
                     C.byref(ostruct), ws, ws.numel(), ops.STREAM, device=dev)
        return outs["rgb"], outs["distance"], outs["acc"]

    # ------------------------------------------------------------------ chunked frame rendering
    def render_rays(self, rays, chunks=4096, out=None):
        """Chunk loop of model.py:261-269 with everything resident on the device: one H2D copy of the
        ray batch, outputs written in place, no per-chunk sync.  The chunk PARTITION is the reference's
        (consecutive blocks of `chunks` rays) because the global contraction norm makes results
        chunk-dependent.  Returns float device tensors (rgb[n,3], distance[n], acc[n]); `out` = three such
        contiguous tensors to write into (the send block of a multi-GPU pixel gather, distributed.PixelGather)."""
        dev = torch.device(self.device)
        rays = namedtuple_map(lambda r: torch.as_tensor(r).to(device=dev, dtype=torch.float32).contiguous(), rays)
        length = rays[0].shape[0]
        if out is not None:
            rgb, dist, acc = out
            for t, shape in ((rgb, (length, 3)), (dist, (length,)), (acc, (length,))):
                if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != rays[0].device:
                    raise RuntimeError(f"render_rays: `out` tensors must be contiguous float32 {shape} on {rays[0].device}")
        else:
            rgb = torch.empty(length, 3, device=dev)
            dist = torch.empty(length, device=dev)
            acc = torch.empty(length, device=dev)
        fused = not self.mutate_like_reference
        # Small chunks (the reference's default is 128 rays, config.py:49) are launched many at a time: the chunk
        # partition only matters through the per-chunk contraction norm, which the kernels keep per group of `chunks`
        # rays (m360_hyper_t.norm_group_rays) - bit-identical to one launch per chunk, at large-batch efficiency.
        n_max = max(self.prop_net.num_samples, self.nerf_net.num_samples_fine or self.prop_net.num_samples)
        group = chunks if (fused and chunks < self.super_batch_rays and chunks * n_max <= 131072) else 0
        # at most 1024 contraction-norm groups per launch (kMaxNormGroups of the encode kernel)
        step = min(self.super_batch_rays // chunks, 1024) * chunks if group else chunks
        with torch.no_grad():
            for i in range(0, length, step):
                chunk = namedtuple_map(lambda r: r[i:i + step], rays)
                if fused:
                    self._forward_fused(chunk, rgb[i:i + step], dist[i:i + step], acc[i:i + step], stash=False,
                                        norm_group_rays=group)
                else:
                    r, d, a = self(chunk)
                    rgb[i:i + chunks], dist[i:i + chunks], acc[i:i + chunks] = r, d, a
                if self.verbose:
                    print("rendering,schedule:%s / %s" % (i // chunks, length // chunks))
        return rgb, dist, acc

    def render_image(self, rays, height, width, chunks=4096):
        """model.py:254-274 -> (uint8[H,W,3], float32[H,W], float32[H,W]) NumPy arrays."""
        rgb, dist, acc = self.render_rays(rays, chunks)
        rgbs = ops.to8b(rgb).reshape(height, width, 3).cpu().numpy()
        dists = dist.reshape(height, width).cpu().numpy()
        accs = acc.reshape(height, width).cpu().numpy()
        return rgbs, dists, accs

    def render_view(self, cam_to_world, height, width, focal, near, far, ndc=False, chunks=4096):
        """Extension: render one camera pose without ever materialising rays on the host.
        Rays are generated on the device exactly like NeRFDataset.generate_rays / LLFF.generate_rays
        (dataset.py:109-145, :364-387; `ndc=True` for the nerf_360 / llff configuration) and fed to the same
        chunk loop as `render_image`; only the 3x4 pose goes up and the finished frame comes down."""
        from .intern.ray import generate_rays
        dev = torch.device(self.device)
        pose = torch.as_tensor(cam_to_world).to(device=dev, dtype=torch.float32)
        rays = generate_rays(pose, height, width, focal, near, far, ndc)
        return self.render_image(rays, height, width, chunks)

    def train(self, mode=True):
        """model.py:276-279 (note: sub-nets keep the `randomized` they were built with)."""
        self.randomized = self.init_randomized
        super().train(mode)
        return self

    def eval(self):
        """model.py:281-283."""
        self.randomized = False
        return super().eval()
