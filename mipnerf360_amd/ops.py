"""Tensor-level wrappers over the libm360 C-ABI.

PyTorch is plumbing here: it owns device memory and the HIP stream; every
function below validates its tensors, allocates outputs with torch and calls one
`extern "C"` entry point with raw device pointers on the current stream.
CPU tensors are rejected — there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib

IPE_CH = 42


def _require_device(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: mipnerf360_amd executes on HIP (MI355X) devices only; there is no CPU path")


def dev(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """fp32, contiguous, on a HIP device (dtype other than fp32 is an error, as in the reference
    whose Jacobian buffer is fp32-only, intern/parameterization.py:75)."""
    _require_device(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _Stream:
    """Placeholder for the `m360_stream_t` argument: `call` replaces it by the current HIP stream of the device the
    tensor arguments live on."""

    def __repr__(self):
        return "STREAM"


STREAM = _Stream()


def stream(device=None) -> int:
    """Raw handle of torch's current HIP stream on `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def round_up(v: int, m: int = 32) -> int:
    return (v + m - 1) // m * m


def call(name: str, *args, device=None):
    """One C-ABI call.  Tensor arguments are passed as raw device pointers; all of them (and `device`, when given:
    descriptor structs carry pointers ctypes cannot see) must live on ONE HIP device.  The call runs with that device
    current (libm360 launches on the current device: hipGetDevice / the stream's device) and `STREAM` becomes torch's
    current stream of THAT device - so `mipNeRF360(device='cuda:1')` works without a prior torch.cuda.set_device(1),
    like the reference with `device=config.device` alone."""
    device = None if device is None else torch.device(device)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                raise RuntimeError(f"{name}: tensor argument on {a.device}; mipnerf360_amd has no CPU path")
            if device is None:
                device = a.device
            elif a.device != device:
                raise RuntimeError(f"{name}: tensor arguments on different devices ({device} and {a.device})")
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    fn = getattr(_lib.lib(), name)
    if device is None:  # no tensor at all (size queries)
        _lib.check(fn(*[None if a is STREAM else a for a in conv]), name)
        return
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(device):
        s = torch.cuda.current_stream(device).cuda_stream
        _lib.check(fn(*[s if a is STREAM else a for a in conv]), name)


_call = call


# ----------------------------------------------------------------------------- per-call tuning (include/m360.h: m360_hyper_t.tuning, .side, .chain_debug_*)
# libm360 holds NO switch of its own (0.2.0): what used to be process-wide m360_set_* calls are fields of every call's m360_hyper_t.  The host
# mirror keeps the CURRENT THREAD's choice here - thread-local, so two threads driving two models never see each other's setting - and
# writes it into each hyper struct it builds (apply_tuning).  Defaults: chain on, paired rows, one-wave weight gradient, overlapped backward.
import threading  # noqa: E402

_TUNE_DEFAULTS = dict(hidden_chain=True, paired_rows=True, wgrad_form=1, backward_overlap=True, chain_cooperative=False, chain_ungated=False,
                      chain_wait_ticks=0, chain_fault=0)


class _Tuning(threading.local):
    def __init__(self):
        self.__dict__.update(_TUNE_DEFAULTS)


_tuning = _Tuning()


def _swap(name, value):
    old = getattr(_tuning, name)
    setattr(_tuning, name, value)
    return old


def set_paired_rows(on: bool) -> bool:
    """Paired rows between the layers of the bf16 / bf16x3 MLPs on / off for this thread's calls (M360_TUNE_PLAIN_ROWS; same bits either
    way); returns the old setting"""
    return _swap("paired_rows", bool(on))


def set_hidden_chain(on: bool) -> bool:
    """bf16 mode: the six hidden NeRF layers as one launch (default) or six (M360_TUNE_NO_HIDDEN_CHAIN; same bits); returns the old setting"""
    return _swap("hidden_chain", bool(on))


def set_wgrad_bf16_form(form: int) -> int:
    """m360_linear_wgrad_bf16: 1 = one wave per SIMD (default), 0 = the 8-wave kernel (M360_TUNE_WGRAD_FORM0); returns the old setting"""
    return _swap("wgrad_form", 1 if form else 0)


def set_backward_overlap(on: bool) -> bool:
    """bf16 backward: hand the calls a second stream (m360_side_t) for the ReLU mask beside the weight gradient (default) or none"""
    return _swap("backward_overlap", bool(on))


def set_chain_cooperative(on: bool) -> bool:
    return _swap("chain_cooperative", bool(on))


def set_chain_debug(wait_ticks: int = 0, fault: int = 0) -> None:
    """Test hooks of the layer chain, honoured by the DIAGNOSTICS build only (`with _lib.use_library(_lib.DIAG_LIB_PATH)`; the product
    library refuses a call that carries them): bound of one wait in 100 MHz ticks (0 = default 0.1 s), fault to inject (0 none, 1 = a
    workgroup reports a foreign XCD, 2 = every wave gives up at its first wait, -1 = no gated re-run behind the chain: A/B of its cost)."""
    _tuning.chain_wait_ticks = max(int(wait_ticks), 0)
    _tuning.chain_fault = int(fault) if int(fault) > 0 else 0
    _tuning.chain_ungated = int(fault) == -1


def tuning_bits() -> int:
    t = _tuning
    return ((0 if t.hidden_chain else _lib.TUNE_NO_HIDDEN_CHAIN) | (0 if t.paired_rows else _lib.TUNE_PLAIN_ROWS) |
            (0 if t.wgrad_form else _lib.TUNE_WGRAD_FORM0) | (_lib.TUNE_CHAIN_COOPERATIVE if t.chain_cooperative else 0) |
            (_lib.TUNE_CHAIN_UNGATED if t.chain_ungated else 0))


def apply_tuning(hyper) -> None:
    """Write this thread's choices into a _lib.HyperStruct (every call of the stage drivers carries its own)."""
    hyper.tuning = tuning_bits()
    hyper.chain_debug_wait_ticks = _tuning.chain_wait_ticks
    hyper.chain_debug_fault = _tuning.chain_fault


def _opts_struct():
    h = _lib.HyperStruct()
    apply_tuning(h)
    return h


class _Side:
    """A caller-owned second stream of libm360 (m360_side_t) for one (library build, device, stream)."""

    def __init__(self, device):
        import ctypes as C
        self.lib = _lib.lib()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.m360_side_create(C.byref(h)), "m360_side_create")
        self.handle = h.value

    def __del__(self):
        try:
            if self.handle:
                self.lib.m360_side_destroy(self.handle)
        except Exception:
            pass


_SIDES = {}
_SIDES_LOCK = threading.Lock()


def backward_overlap_wanted() -> bool:
    return bool(_tuning.backward_overlap)


def side_handle(device):
    """The m360_side_t to put into m360_hyper_t.side of a bf16 backward on torch's current stream of `device`.  One per (build of the
    library, device, stream): calls on one stream are issued one after the other."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (getattr(_lib.lib(), "m360_path", ""), device.index, torch.cuda.current_stream(device).cuda_stream)
    with _SIDES_LOCK:
        side = _SIDES.get(key)
        if side is None:
            side = _SIDES[key] = _Side(device)
    return side.handle


def _ws(device) -> torch.Tensor:
    n = _lib.lib().m360_contract_workspace_bytes()
    return torch.empty(n, dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- random numbers
def philox_state(device) -> Tuple[int, int]:
    """(seed, offset) for one randomized libm360 call, taken from torch's device generator - and the generator advanced by 4, so
    that no later torch kernel (nor a later call of ours) reuses the counters (include/m360.h: m360_hyper_t.rng_offset).  Same
    `torch.manual_seed` -> same numbers, like torch.rand.  The kernels draw their uniforms themselves (Philox4x32-10): no
    [B, N + 1] tensor of torch.rand is materialised (the reference's intern/ray.py:31,104).
    Which counters a call uses depends on how many randomized calls came before it: `mipNeRF360.forward` draws ONE state for both of its
    stages (fused or staged: the same samples for the same seed), `prop_net.forward` and `nerf_net.forward` called on their own (train.py)
    draw one each.  Always torch's DEFAULT generator of the device (what `torch.manual_seed` / `torch.random.fork_rng(devices=[i])`
    govern); a user-made torch.Generator cannot be handed in - replay recorded uniforms through `module.replay_uniforms` instead.
    Not under stream capture: (seed, offset) are host values a graph would bake in, every replay would draw the same numbers."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("randomized=True inside a HIP-graph capture: the Philox (seed, offset) of the call would be baked into the graph and every "
                           "replay would draw the same uniforms - capture a deterministic model, or hand the draws in as tensors "
                           "(prop_net.replay_uniforms / nerf_net.replay_uniforms, refilled between replays)")
    gen = torch.cuda.default_generators[idx]
    seed, off = int(gen.initial_seed()), int(gen.get_offset())
    gen.set_offset(off + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, off // 4


def philox_uniform(seed: int, offset: int, stream_id: int, n: int, device) -> torch.Tensor:
    """m360_philox_uniform: the uniforms a kernel draws for elements 0..n-1 of stream `stream_id` (0 = t_rand, 1 = u_rand)."""
    out = torch.empty(n, device=device)
    _call("m360_philox_uniform", C.c_ulonglong(seed), C.c_ulonglong(offset), int(stream_id), int(n), out, STREAM)
    return out


# ----------------------------------------------------------------------------- sampling
def sample_t(near, far, num_samples: int, t_rand=None, philox: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """t_rand: uniforms [B, N+1] for the stratified jitter (a tensor wins); philox = (seed, offset): drawn in the kernel."""
    near, far = dev(near, "near"), dev(far, "far")
    B = near.shape[0]
    t = torch.empty(B, num_samples + 1, device=near.device)
    if t_rand is not None:
        t_rand = dev(t_rand, "t_rand")
    if t_rand is None and philox is not None:
        _call("m360_sample_t_philox", near, far, B, num_samples, C.c_ulonglong(philox[0]), C.c_ulonglong(philox[1]), t, STREAM)
    else:
        _call("m360_sample_t", near, far, t_rand, B, num_samples, t, STREAM)
    return t


def g(x) -> torch.Tensor:
    x = dev(x, "x")
    y = torch.empty_like(x)
    _call("m360_g", x, x.numel(), y, STREAM)
    return y


def s_to_t(s_vals, near, far) -> torch.Tensor:
    near, far = dev(near, "near"), dev(far, "far")
    B = near.shape[0]
    s_vals = dev(s_vals, "s_vals")
    s2 = s_vals.expand(B, s_vals.shape[-1]).contiguous() if s_vals.dim() == 1 or s_vals.shape[0] != B else s_vals
    t = torch.empty_like(s2)
    _call("m360_s_to_t", s2, near, far, B, s2.shape[-1], t, STREAM)
    return t


def contract(x) -> torch.Tensor:
    x = dev(x, "x")
    y = torch.empty_like(x)
    ws = _ws(x.device)
    _call("m360_contract", x, x.numel(), y, ws, ws.numel(), STREAM)
    return y


def t_to_s(t_vals, near, far, near_calls: int = 1, far_calls: int = 1) -> torch.Tensor:
    t_vals, near, far = dev(t_vals, "t_vals"), dev(near, "near"), dev(far, "far")
    B, M = t_vals.shape
    s = torch.empty_like(t_vals)
    _call("m360_t_to_s", t_vals, near, far, B, M, near_calls, far_calls, s, STREAM)
    return s


# ----------------------------------------------------------------------------- gaussians
def frustum_moments(t0, t1, radii, stable: bool = True):
    t0, t1, radii = dev(t0, "t0"), dev(t1, "t1"), dev(radii, "radii")
    B, N = t0.shape
    outs = [torch.empty_like(t0) for _ in range(3)]
    _call("m360_frustum_moments" if stable else "m360_frustum_moments_unstable", t0, t1, radii, B, N, *outs, STREAM)
    return tuple(outs)


def gaussian_to_xyz_diag(d, t_mean, t_var, r_var):
    d, t_mean, t_var, r_var = dev(d, "d"), dev(t_mean, "t_mean"), dev(t_var, "t_var"), dev(r_var, "r_var")
    B, N = t_mean.shape
    mean = torch.empty(B, N, 3, device=d.device)
    cov = torch.empty(B, N, 3, device=d.device)
    _call("m360_gaussian_to_xyz_diag", d, t_mean, t_var, r_var, B, N, mean, cov, STREAM)
    return mean, cov


def gaussian_to_xyz(d, t_mean, t_var, r_var):
    d, t_mean, t_var, r_var = dev(d, "d"), dev(t_mean, "t_mean"), dev(t_var, "t_var"), dev(r_var, "r_var")
    B, N = t_mean.shape
    mean = torch.empty(B, N, 3, device=d.device)
    cov = torch.empty(B, N, 3, 3, device=d.device)
    _call("m360_gaussian_to_xyz", d, t_mean, t_var, r_var, B, N, mean, cov, STREAM)
    return mean, cov


def gaussian_contract(mean, cov):
    mean, cov = dev(mean, "mean"), dev(cov, "cov")
    S = mean.numel() // 3
    mo, co = torch.empty_like(mean), torch.empty_like(cov)
    ws = _ws(mean.device)
    _call("m360_gaussian_contract", mean, cov, S, mo, co, ws, ws.numel(), STREAM)
    return mo, co


def para_rays(t_vals, origins, directions, radii):
    t_vals, origins = dev(t_vals, "t_vals"), dev(origins, "origins")
    directions, radii = dev(directions, "directions"), dev(radii, "radii")
    B, M = t_vals.shape
    N = M - 1
    means = torch.empty(B, N, 3, device=t_vals.device)
    covs = torch.empty(B, N, 3, 3, device=t_vals.device)
    ws = _ws(t_vals.device)
    _call("m360_para_rays", t_vals, origins, directions, radii, B, N, means, covs,
          ws, ws.numel(), STREAM)
    return means, covs


# ----------------------------------------------------------------------------- encodings
def ipe(mean, cov=None) -> torch.Tensor:
    mean = dev(mean, "mean")
    cov = None if cov is None else dev(cov, "cov")
    S = mean.numel() // 3
    enc = torch.empty(mean.shape[:-1] + (IPE_CH,), device=mean.device)
    _call("m360_ipe", mean, cov, S, enc, STREAM)
    return enc


def viewdir_enc(viewdirs, min_deg: int, max_deg: int) -> torch.Tensor:
    viewdirs = dev(viewdirs, "viewdirs")
    B = viewdirs.numel() // 3
    enc = torch.empty(viewdirs.shape[:-1] + (4 * (max_deg - min_deg),), device=viewdirs.device)
    _call("m360_viewdir_enc", viewdirs, B, min_deg, max_deg, enc, STREAM)
    return enc


def encode_features(t_vals, origins, directions, radii, vdenc, ld_feat: Optional[int] = None, row_format: int = 0) -> torch.Tensor:
    """MLP input rows of one chunk (model.py:82-88).  row_format 0: fp32 [S, ld]; 1: bf16 [S, ld]; 2: bf16 [hi | lo] pairs
    [S, 2 ld]; 3: bf16 "x6" rows [S, 6 ld] = [lo | mid | hi | mid | hi | hi] (what the bf16 / bf16x3 MLPs read)."""
    t_vals, origins = dev(t_vals, "t_vals"), dev(origins, "origins")
    directions, radii, vdenc = dev(directions, "directions"), dev(radii, "radii"), dev(vdenc, "vdenc")
    B, M = t_vals.shape
    N = M - 1
    vd_ch = vdenc.shape[-1]
    ld = ld_feat or round_up(IPE_CH + vd_ch)
    ws = _ws(t_vals.device)
    if row_format == 0:
        feat = torch.empty(B * N, ld, device=t_vals.device)
        _call("m360_encode_features", t_vals, origins, directions, radii, vdenc, vd_ch, B, N,
              feat, ld, ws, ws.numel(), STREAM)
        return feat
    if row_format not in (1, 2, 3):
        raise ValueError(f"encode_features: row_format {row_format}")
    feat = torch.empty(B * N, {1: 1, 2: 2, 3: 6}[row_format] * ld, device=t_vals.device, dtype=torch.bfloat16)
    _call("m360_encode_features_grouped", t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat, ld, row_format, 0,
          ws, ws.numel(), STREAM)
    return feat


# ----------------------------------------------------------------------------- MLP
def pack_linear(weight, bias=None, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out), k_pad or round_up(k_in)
    wp = torch.empty(n_pad, k_pad, device=weight.device)
    bp = torch.empty(n_pad, device=weight.device)
    b = None if bias is None else dev(bias.detach(), "bias")
    _call("m360_pack_linear", weight, b, n_out, k_in, n_pad, k_pad, wp, bp, STREAM)
    return wp, bp


def linear(x, w_packed, b_packed, act: int = _lib.ACT_NONE, out: Optional[torch.Tensor] = None,
           balanced: bool = False) -> torch.Tensor:
    """One packed layer.  balanced=True hands the last tiles out through a (fresh, zeroed) queue word so that the 8 XCDs
    finish together (m360_linear_balanced); the result is bit-identical either way."""
    x, w_packed, b_packed = dev(x, "x"), dev(w_packed, "w_packed"), dev(b_packed, "b_packed")
    M, ldx = x.shape
    n_pad, k_pad = w_packed.shape
    if ldx != k_pad:
        raise RuntimeError(f"linear: x has {ldx} columns, packed weight expects {k_pad}")
    y = out if out is not None else torch.empty(M, n_pad, device=x.device)
    if balanced:
        queue = torch.zeros(1, dtype=torch.int32, device=x.device)
        _call("m360_linear_balanced", x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, y.shape[1], queue, STREAM)
    else:
        _call("m360_linear", x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, y.shape[1], STREAM)
    return y


def mean_sumsq(t_vals, directions, radii) -> torch.Tensor:
    """float64[1]: sum of squares of the un-contracted sample means of these rays (a shard of a larger batch)."""
    t_vals, directions, radii = dev(t_vals, "t_vals"), dev(directions, "directions"), dev(radii, "radii")
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    out = torch.empty(1, dtype=torch.float64, device=t_vals.device)
    ws = torch.empty(int(_lib.lib().m360_contract_workspace_bytes()), dtype=torch.uint8, device=t_vals.device)
    _call("m360_mean_sumsq", t_vals, directions, radii, B, N, out, ws, ws.numel(), STREAM)
    return out


def pack_linear_transposed(weight, n_pad: Optional[int] = None, k_pad: Optional[int] = None) -> torch.Tensor:
    """[n_out,k_in] weight -> zero-padded transpose [k_pad,n_pad] (the operand of linear_dgrad)."""
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out, 32), k_pad or round_up(k_in, 32)
    wt = torch.empty(k_pad, n_pad, device=weight.device)
    _call("m360_pack_linear_transposed", weight, n_out, k_in, n_pad, k_pad, wt, STREAM)
    return wt


def linear_dgrad(dz, wt_packed, relu_out=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = dz @ W (masked where relu_out <= 0), W given as its packed transpose [k_pad,n_pad]."""
    dz, wt_packed = dev(dz, "dz"), dev(wt_packed, "wt_packed")
    M, n_pad = dz.shape
    k_pad = wt_packed.shape[0]
    if wt_packed.shape[1] != n_pad:
        raise RuntimeError(f"linear_dgrad: dz has {n_pad} columns, packed transpose expects {wt_packed.shape[1]}")
    dx = out if out is not None else torch.empty(M, k_pad, device=dz.device)
    if relu_out is not None:
        relu_out = dev(relu_out, "relu_out")
        if tuple(relu_out.shape) != (M, dx.shape[1]):
            raise RuntimeError("linear_dgrad: relu_out must have the shape of dx")
    _call("m360_linear_dgrad", dz, M, n_pad, wt_packed, k_pad, n_pad, relu_out, dx, dx.shape[1], STREAM)
    return dx


def linear_wgrad(dz, x, want_bias: bool = True):
    """-> (grad_w[n_pad,k_pad] = dz^T @ x, grad_b[n_pad] = dz.sum(0) or None)"""
    dz, x = dev(dz, "dz"), dev(x, "x")
    M, n_pad = dz.shape
    if x.shape[0] != M:
        raise RuntimeError("linear_wgrad: dz and x need the same number of rows")
    k_pad = x.shape[1]
    gw = torch.empty(n_pad, k_pad, device=dz.device)
    gb = torch.empty(n_pad, device=dz.device) if want_bias else None
    ws = torch.empty(_lib.lib().m360_linear_wgrad_workspace_bytes(M, n_pad, k_pad), dtype=torch.uint8, device=dz.device)
    _call("m360_linear_wgrad", dz, n_pad, x, k_pad, M, n_pad, k_pad, gw, gb, ws, ws.numel(), STREAM)
    return gw, gb


def pack_linear_bf16_transposed(weight, n_pad: Optional[int] = None, k_pad: Optional[int] = None) -> torch.Tensor:
    """[n_out,k_in] fp32 weight -> zero-padded bf16 transpose [k_pad,n_pad] (the operand of linear_dgrad_bf16)."""
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out, 64), k_pad or round_up(k_in, 64)
    wt = torch.empty(k_pad, n_pad, device=weight.device, dtype=torch.bfloat16)
    _call("m360_pack_linear_bf16_transposed", weight, n_out, k_in, n_pad, k_pad, wt, STREAM)
    return wt


def linear_dgrad_bf16(dz, wt_packed, relu_out=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 dx = dz @ W (cleared where relu_out <= 0), W given as its packed bf16 transpose [k_pad,n_pad] (m360_linear_dgrad_bf16)."""
    dz, wt_packed = dev_bf16(dz, "dz"), dev_bf16(wt_packed, "wt_packed")
    M, n_pad = dz.shape
    k_pad = wt_packed.shape[0]
    if wt_packed.shape[1] != n_pad:
        raise RuntimeError(f"linear_dgrad_bf16: dz has {n_pad} columns, packed transpose expects {wt_packed.shape[1]}")
    dx = out if out is not None else torch.empty(M, k_pad, device=dz.device, dtype=torch.bfloat16)
    if relu_out is not None:
        relu_out = dev_bf16(relu_out, "relu_out")
        if tuple(relu_out.shape) != (M, dx.shape[1]):
            raise RuntimeError("linear_dgrad_bf16: relu_out must have the shape of dx")
    _call("m360_linear_dgrad_bf16", dz, M, n_pad, wt_packed, k_pad, n_pad, relu_out, dx, dx.shape[1], STREAM)
    return dx


def linear_wgrad_bf16(dz, x, want_bias: bool = True):
    """bf16 rows -> fp32 (grad_w[n_pad,k_pad] = dz^T @ x, grad_b[n_pad] = dz.sum(0) or None) (m360_linear_wgrad_bf16)"""
    dz, x = dev_bf16(dz, "dz"), dev_bf16(x, "x")
    M, n_pad = dz.shape
    if x.shape[0] != M:
        raise RuntimeError("linear_wgrad_bf16: dz and x need the same number of rows")
    k_pad = x.shape[1]
    gw = torch.empty(n_pad, k_pad, device=dz.device)
    gb = torch.empty(n_pad, device=dz.device) if want_bias else None
    ws = torch.empty(max(int(_lib.lib().m360_linear_wgrad_bf16_workspace_bytes(M, n_pad, k_pad)), 256), dtype=torch.uint8, device=dz.device)
    _call("m360_linear_wgrad_bf16", dz, n_pad, x, k_pad, M, n_pad, k_pad, gw, gb, ws, ws.numel(), tuning_bits(), STREAM)
    return gw, gb


def dev_bf16(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    _require_device(t, name)
    if t.dtype != torch.bfloat16:
        raise RuntimeError(f"{name} must be bfloat16, got {t.dtype}")
    return t.contiguous()


def params_nan_flag(tensors) -> torch.Tensor:
    """int32[1] on the device: 1 if any of the fp32 device tensors holds a NaN, else 0 - one launch for the whole parameter set
    (no host synchronisation: read the flag when it is needed)."""
    import ctypes
    ts = [dev(t.detach(), "parameter") for t in tensors]
    flag = torch.empty(1, dtype=torch.int32, device=ts[0].device)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    counts = (ctypes.c_long * len(ts))(*[t.numel() for t in ts])
    _call("m360_params_nan_flag", ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(counts, ctypes.c_void_p), len(ts), flag, STREAM)
    del ts  # (kept alive until the launch is queued)
    return flag


_PACK_COLS = {_lib.PACK_F32: 1, _lib.PACK_BF16: 1, _lib.PACK_BF16X3: 3, _lib.PACK_BF16X6: 6}


def pack_many(items, nan_flag: bool = False):
    """Every packing of a parameter set in ONE launch (m360_pack_many): `items` is a list of
    (weight, bias or None, n_pad, k_pad, format[, w_out, b_out]) with format one of _lib.PACK_*; the per-layer calls' pads, the same bits.
    Returns ([(w_packed, b_packed or None), ...], flag) - flag: int32[1] on the device, 1 when a source value is NaN (None unless asked for).
    w_out / b_out: write into these (views of a larger buffer: the output heads side by side) instead of fresh tensors."""
    import ctypes
    arr = (_lib.PackItem * max(len(items), 1))()
    outs, keep = [], []
    for i, item in enumerate(items):
        weight, bias, n_pad, k_pad, fmt = item[:5]
        w_out, b_out = (item[5], item[6]) if len(item) > 5 else (None, None)
        weight = dev(weight.detach(), "weight")
        n_out, k_in = weight.shape
        transposed = fmt in (_lib.PACK_F32_T, _lib.PACK_BF16_T)
        dt = torch.float32 if fmt in (_lib.PACK_F32, _lib.PACK_F32_T) else torch.bfloat16
        if w_out is None:
            shape = (k_pad, n_pad) if transposed else (n_pad, _PACK_COLS[fmt] * k_pad)
            w_out = torch.empty(*shape, device=weight.device, dtype=dt)
        elif w_out.dtype != dt or not w_out.is_contiguous() or w_out.device != weight.device:
            raise RuntimeError("pack_many: w_out must be a contiguous tensor of the packing's dtype on the weight's device")
        b = None if bias is None else dev(bias.detach(), "bias")
        if b_out is None and not transposed and (len(item) <= 5):
            b_out = torch.empty(n_pad, device=weight.device)
        if b_out is not None:
            b_out = dev(b_out, "b_out")
        arr[i] = _lib.PackItem(weight.data_ptr(), 0 if b is None else b.data_ptr(), w_out.data_ptr(), 0 if b_out is None else b_out.data_ptr(),
                               n_out, k_in, n_pad, k_pad, fmt, 0)
        outs.append((w_out, b_out))
        keep.append((weight, b))
    if not items:
        return [], None
    flag = torch.empty(1, dtype=torch.int32, device=keep[0][0].device) if nan_flag else None
    _call("m360_pack_many", ctypes.cast(arr, ctypes.c_void_p), len(items), flag, STREAM, device=keep[0][0].device)
    del keep  # (alive until the launch is queued)
    return outs, flag


def pack_linear_bf16(weight, bias=None, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
    """fp32 Linear -> zero-padded bf16 weight [n_pad,k_pad] (k_pad multiple of 64) + fp32 bias [n_pad]."""
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out, 64), k_pad or round_up(k_in, 64)
    wp = torch.empty(n_pad, k_pad, device=weight.device, dtype=torch.bfloat16)
    bp = torch.empty(n_pad, device=weight.device)
    b = None if bias is None else dev(bias.detach(), "bias")
    _call("m360_pack_linear_bf16", weight, b, n_out, k_in, n_pad, k_pad, wp, bp, STREAM)
    return wp, bp


def linear_bf16(x, w_packed, b_packed, act: int = _lib.ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 x [M,k_pad] * bf16 W^T + fp32 bias -> bf16 [M,n_pad] (fp32 accumulate on MFMA)."""
    x, w_packed, b_packed = dev_bf16(x, "x"), dev_bf16(w_packed, "w_packed"), dev(b_packed, "b_packed")
    M, ldx = x.shape
    n_pad, k_pad = w_packed.shape
    if ldx != k_pad:
        raise RuntimeError(f"linear_bf16: x has {ldx} columns, packed weight expects {k_pad}")
    y = out if out is not None else torch.empty(M, n_pad, device=x.device, dtype=torch.bfloat16)
    _call("m360_linear_bf16", x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, y.shape[1], STREAM)
    return y


def pack_linear_bf16x3(weight, bias=None, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
    """fp32 Linear -> bf16x3 packing [n_pad, 3 k_pad] = [Wh | Wh | Wl] (Wh = bf16(W), Wl = bf16(W - Wh)) + fp32 bias."""
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out, 64), k_pad or round_up(k_in, 64)
    wp = torch.empty(n_pad, 3 * k_pad, device=weight.device, dtype=torch.bfloat16)
    bp = torch.empty(n_pad, device=weight.device)
    b = None if bias is None else dev(bias.detach(), "bias")
    _call("m360_pack_linear_bf16x3", weight, b, n_out, k_in, n_pad, k_pad, wp, bp, STREAM)
    return wp, bp


def split_bf16x3(x: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K] -> the [hi | lo] bf16 pair rows [M, 2 K] that m360_linear_bf16x3 reads (what the encoder / a previous
    bf16x3 layer writes on the device; this host-side form is for tests and tools)."""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return torch.cat([hi, lo], 1).contiguous()


def join_bf16x3(y: torch.Tensor) -> torch.Tensor:
    """[M, 2 N] bf16 pair rows -> fp32 [M, N] = hi + lo"""
    n = y.shape[1] // 2
    return y[:, :n].float() + y[:, n:].float()


def linear_bf16x3(x, w_packed3, b_packed, act: int = _lib.ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M, 2 k_pad] bf16 (hi | lo) -> y [M, 2 n_pad] bf16 (hi | lo): y = act(x W^T + b) with every product formed as
    xh wh + xl wh + xh wl on the bf16 MFMA, fp32 accumulation (include/m360.h, m360_linear_bf16x3)."""
    x, w_packed3, b_packed = dev_bf16(x, "x"), dev_bf16(w_packed3, "w_packed3"), dev(b_packed, "b_packed")
    M, ldx = x.shape
    n_pad, k3 = w_packed3.shape
    k_pad = k3 // 3
    if ldx != 2 * k_pad:
        raise RuntimeError(f"linear_bf16x3: x has {ldx} columns, packed weight expects {2 * k_pad} (hi | lo)")
    y = out if out is not None else torch.empty(M, 2 * n_pad, device=x.device, dtype=torch.bfloat16)
    _call("m360_linear_bf16x3", x, M, ldx, w_packed3, b_packed, n_pad, k_pad, act, y, y.shape[1], STREAM)
    return y


def linear_bf16x3_bf16out(x, w_packed3, b_packed, act: int = _lib.ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M, 2 k_pad] bf16 (hi | lo), weights [Wh | Wh | Wl] -> y [M, n_pad] bf16: the three products of m360_linear_bf16x3 with ONE bf16
    term out (m360_linear_bf16x3_bf16out: the first layers of the bf16 mode)."""
    x, w_packed3, b_packed = dev_bf16(x, "x"), dev_bf16(w_packed3, "w_packed3"), dev(b_packed, "b_packed")
    M, ldx = x.shape
    n_pad, k3 = w_packed3.shape
    k_pad = k3 // 3
    if ldx != 2 * k_pad:
        raise RuntimeError(f"linear_bf16x3_bf16out: x has {ldx} columns, packed weight expects {2 * k_pad} (hi | lo)")
    y = out if out is not None else torch.empty(M, n_pad, device=x.device, dtype=torch.bfloat16)
    _call("m360_linear_bf16x3_bf16out", x, M, ldx, w_packed3, b_packed, n_pad, k_pad, act, y, y.shape[1], STREAM)
    return y


def pack_linear_bf16x6(weight, bias=None, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
    """fp32 Linear -> "x6" packing [n_pad, 6 k_pad] = [Wh | Wm | Wl | Wh | Wm | Wh] (three bf16 terms per weight: all 24 bits)
    + fp32 bias: the first layers of the bf16 / bf16x3 modes (include/m360.h, m360_pack_linear_bf16x6)."""
    weight = dev(weight.detach(), "weight")
    n_out, k_in = weight.shape
    n_pad, k_pad = n_pad or round_up(n_out, 64), k_pad or round_up(k_in, 64)
    wp = torch.empty(n_pad, 6 * k_pad, device=weight.device, dtype=torch.bfloat16)
    bp = torch.empty(n_pad, device=weight.device)
    b = None if bias is None else dev(bias.detach(), "bias")
    _call("m360_pack_linear_bf16x6", weight, b, n_out, k_in, n_pad, k_pad, wp, bp, STREAM)
    return wp, bp


def split_bf16x6(x: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K] -> the x6 rows [M, 6 K] = [lo | mid | hi | mid | hi | hi] the first layers read (what the encoder writes on the
    device with row format 3; this host-side form is for tests and tools)."""
    hi = x.bfloat16()
    r = torch.where(torch.isfinite(hi.float()), x - hi.float(), torch.zeros_like(x))
    mid = r.bfloat16()
    lo = (r - mid.float()).bfloat16()
    return torch.cat([lo, mid, hi, mid, hi, hi], 1).contiguous()


def linear_bf16_split(x, w_packed, b_packed, act: int = _lib.ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 x [M, k_pad] * bf16 W^T + fp32 bias -> [M, 2 n_pad] bf16 (hi | lo) pair rows (m360_linear_bf16_split)."""
    x, w_packed, b_packed = dev_bf16(x, "x"), dev_bf16(w_packed, "w_packed"), dev(b_packed, "b_packed")
    M, ldx = x.shape
    n_pad, k_pad = w_packed.shape
    if ldx != k_pad:
        raise RuntimeError(f"linear_bf16_split: x has {ldx} columns, packed weight expects {k_pad}")
    y = out if out is not None else torch.empty(M, 2 * n_pad, device=x.device, dtype=torch.bfloat16)
    _call("m360_linear_bf16_split", x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, y.shape[1], STREAM)
    return y


# ----------------------------------------------------------------------------- per-ray scans
def _density2d(density):
    return density[..., 0] if density.dim() == 3 else density


def density_to_weight(t_vals, density, dirs) -> torch.Tensor:
    t_vals, density, dirs = dev(t_vals, "t_vals"), dev(_density2d(density), "density"), dev(dirs, "dirs")
    B, N = density.shape
    w = torch.empty(B, N, device=t_vals.device)
    _call("m360_density_to_weight", t_vals, density, dirs, B, N, w, STREAM)
    return w


def sorted_pdf(bins, weights, num_samples: int, u_rand=None, philox: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """u_rand: uniforms [B, num_samples] of the randomized branch (a tensor wins); philox = (seed, offset): drawn in the kernel."""
    bins, weights = dev(bins, "bins"), dev(weights, "weights")
    B, nb = bins.shape
    if weights.shape[-1] != nb - 1:
        raise RuntimeError(f"sorted_pdf: weights must have {nb - 1} entries per ray, got {weights.shape[-1]}")
    out = torch.empty(B, num_samples, device=bins.device)
    u = None if u_rand is None else dev(u_rand, "u_rand")
    if u is None and philox is not None:
        _call("m360_sorted_pdf_philox", bins, weights, B, nb, num_samples, C.c_ulonglong(philox[0]), C.c_ulonglong(philox[1]), out, STREAM)
    else:
        _call("m360_sorted_pdf", bins, weights, u, B, nb, num_samples, out, STREAM)
    return out


def resample_t(t_vals, weights, resample_padding: float, u_rand=None, num_out: Optional[int] = None,
               philox: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """num_out (extension): number of resampled values per ray; None = t_vals.shape[-1] as in the reference.
    u_rand / philox: as for sorted_pdf."""
    t_vals, weights = dev(t_vals, "t_vals"), dev(weights, "weights")
    B, M = t_vals.shape
    n_out = M if num_out is None else int(num_out)
    out = torch.empty(B, n_out, device=t_vals.device)
    u = None if u_rand is None else dev(u_rand, "u_rand")
    if u is None and philox is not None:
        _call("m360_resample_t_philox", t_vals, weights, B, M - 1, n_out, float(resample_padding), C.c_ulonglong(philox[0]),
              C.c_ulonglong(philox[1]), out, STREAM)
    else:
        _call("m360_resample_t_n", t_vals, weights, u, B, M - 1, n_out, float(resample_padding), out, STREAM)
    return out


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd: bool):
    rgb, density = dev(rgb, "rgb"), dev(_density2d(density), "density")
    t_vals, dirs = dev(t_vals, "t_vals"), dev(dirs, "dirs")
    B, N = density.shape
    d = t_vals.device
    comp, dist, acc, w = torch.empty(B, 3, device=d), torch.empty(B, device=d), torch.empty(B, device=d), torch.empty(B, N, device=d)
    _call("m360_volumetric_rendering", rgb, density, t_vals, dirs, B, N, int(bool(white_bkgd)),
          comp, dist, acc, w, STREAM)
    return comp, dist, acc, w


def to8b(x: torch.Tensor) -> torch.Tensor:
    x = dev(x, "image")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _call("m360_to8b", x, x.numel(), out, STREAM)
    return out


def prop_finish(act, head_w, head_b, density_bias, t_vals, dirs, resample_padding, want_t_new=True, u_rand=None):
    act, head_w, head_b = dev(act, "act"), dev(head_w, "head_w"), dev(head_b, "head_b")
    t_vals, dirs = dev(t_vals, "t_vals"), dev(dirs, "dirs")
    B, M = t_vals.shape
    N = M - 1
    w = torch.empty(B, N, device=act.device)
    t_new = torch.empty_like(t_vals) if want_t_new else None
    u = None if u_rand is None else dev(u_rand, "u_rand")
    _call("m360_prop_finish", act, act.shape[1], head_w, head_b, head_w.numel(), float(density_bias),
          t_vals, dirs, u, B, N, float(resample_padding), w, t_new, STREAM)
    return w, t_new


def nerf_finish(act, head_w, head_b, density_bias, rgb_padding, t_vals, dirs, white_bkgd):
    act, head_w, head_b = dev(act, "act"), dev(head_w, "head_w"), dev(head_b, "head_b")
    t_vals, dirs = dev(t_vals, "t_vals"), dev(dirs, "dirs")
    B, M = t_vals.shape
    N = M - 1
    d = act.device
    comp, dist, acc, w = torch.empty(B, 3, device=d), torch.empty(B, device=d), torch.empty(B, device=d), torch.empty(B, N, device=d)
    _call("m360_nerf_finish", act, act.shape[1], head_w, head_b, head_w.shape[1], float(density_bias),
          float(rgb_padding), t_vals, dirs, B, N, int(bool(white_bkgd)), comp, dist, acc, w,
          STREAM)
    return comp, dist, acc, w


def linear_heads(x, w_packed, b_packed, head_w, store_y: bool = True):
    """Last hidden (sigmoid) layer fused with the `heads` = head_w.shape[0] (1 or 4) output heads (m360_linear_heads):
    -> (y[M,n_pad] (fused rows only written when store_y), head_part[fused_rows, slots, heads], fused_rows)."""
    x, w_packed, b_packed, head_w = dev(x, "x"), dev(w_packed, "w_packed"), dev(b_packed, "b_packed"), dev(head_w, "head_w")
    M, ldx = x.shape
    n_pad, k_pad = w_packed.shape
    heads = head_w.shape[0]
    if head_w.shape[1] != n_pad or ldx != k_pad:
        raise RuntimeError("linear_heads: shapes of x / w_packed / head_w do not match")
    lib = _lib.lib()
    fused, slots = int(lib.m360_linear_heads_fused_rows(M, n_pad, 0)), int(lib.m360_linear_heads_slots(n_pad, 0))
    y = torch.zeros(M, n_pad, device=x.device)
    part = torch.zeros(max(fused, 1), max(slots, 1), heads, device=x.device)
    _call("m360_linear_heads", x, M, ldx, w_packed, b_packed, n_pad, k_pad, _lib.ACT_SIGMOID, y, n_pad, int(bool(store_y)),
          head_w, heads, part, STREAM)
    return y, part, fused


def pair_rows(x: torch.Tensor) -> torch.Tensor:
    """Plain bf16 rows <-> the "paired rows" of include/m360.h (M360_ROWS_PAIRED_IN / _OUT), host-side form for tests and tools: in
    every block of 2 rows x 64 columns of the full 256-row tiles the four 64-byte quarters are transposed (its own inverse); rows
    beyond the last full tile stay as they are.  [hi | lo] pair rows: the same call (their halves are multiples of 64 columns)."""
    M, C = x.shape
    if C % 64:
        raise RuntimeError(f"pair_rows: {C} columns (a multiple of 64)")
    full = (M // 256) * 256
    y = x.clone()
    if full:
        y[:full] = x[:full].reshape(full // 2, 2, C // 64, 2, 32).transpose(1, 3).reshape(full, C)
    return y


def rows_pairable(kind: int, n_pad: int, k_pad: int) -> bool:
    """m360_linear_bf16_rows_pairable: does the call `kind` (_lib.PAIRABLE_*) take paired rows at these pads?"""
    return bool(_lib.lib().m360_linear_bf16_rows_pairable(int(kind), int(n_pad), int(k_pad)))


def mlp_chain_bf16_supported(M: int, width: int, layers: int) -> bool:
    return bool(_lib.lib().m360_mlp_chain_bf16_supported(int(M), int(width), int(layers)))


CHAIN_STATUS_FIELDS = ("launches", "recoveries", "timeouts", "xcc_mismatch", "last_error")


def workspace_init(ws: torch.Tensor) -> None:
    """m360_workspace_init: zero the sticky counters of the status block at the start of a forward / chain workspace."""
    _call("m360_workspace_init", ws, STREAM)


def workspace_status(ws: torch.Tensor) -> dict:
    """m360_workspace_status (waits for the current stream): the counters of the bf16 mode's layer chain since workspace_init -
    launches that ran, launches the gated layer-by-layer re-run repaired, waves whose wait ran out, workgroups found off their slot's XCD,
    and the error word of the last launch (0 = its assumptions held)."""
    import ctypes as C
    out = (C.c_uint * 5)()
    _call("m360_workspace_status", ws, out, STREAM)
    return dict(zip(CHAIN_STATUS_FIELDS, (int(v) for v in out)))


def _chain_args(act0, act1, packs):
    import ctypes as C
    act0, act1 = dev_bf16(act0, "act0"), dev_bf16(act1, "act1")
    M, ld = act0.shape
    L = len(packs)
    width = packs[0][0].shape[0]
    ws = torch.empty(int(_lib.lib().m360_mlp_chain_bf16_workspace(M, L)) // 4 + 4, device=act0.device, dtype=torch.int32)
    wl = (C.c_void_p * L)(*[dev_bf16(w, "w").data_ptr() for w, _ in packs])
    bl = (C.c_void_p * L)(*[dev(b, "b").data_ptr() for _, b in packs])
    return act0, act1, M, ld, L, width, ws, C.cast(wl, C.c_void_p), C.cast(bl, C.c_void_p)


def mlp_chain_bf16(act0, act1, packs):
    """m360_mlp_chain_bf16: the equally shaped ReLU layers `packs` = [(w_packed, b_packed), ...] in one launch on PAIRED rows: layer j reads
    act[j & 1], writes act[(j + 1) & 1]; returns the buffer that holds the result.  Raises when the launch reported that one of its
    assumptions did not hold (this entry point has no re-run: its input is overwritten)."""
    act0, act1, M, ld, L, width, ws, wl, bl = _chain_args(act0, act1, packs)
    import ctypes as C
    opts = _opts_struct()
    _call("m360_mlp_chain_bf16", act0, act1, M, ld, wl, bl, L, width, ws, C.byref(opts), STREAM)
    st = workspace_status(ws)
    if st["last_error"] != 0:
        raise RuntimeError(f"m360_mlp_chain_bf16: the launch reported an error (status {st}): a wait ran out or a workgroup was off its XCD")
    return act0 if L % 2 == 0 else act1


def mlp_chain_bf16x3_safe(x_in, act0, act1, packs3):
    """m360_mlp_chain_bf16x3_safe: the bf16x3 mode's hidden layers in one launch - [hi | lo] rows [M, 2 width] (paired), packs3 =
    [(w_packed3 [width, 3 width], b_packed), ...] of pack_linear_bf16x3 -> (result buffer, status dict of this call)."""
    import ctypes as C
    x_in, act0, act1 = dev_bf16(x_in, "x_in"), dev_bf16(act0, "act0"), dev_bf16(act1, "act1")
    M, ld = act0.shape
    L = len(packs3)
    width = packs3[0][0].shape[0]
    ws = torch.empty(int(_lib.lib().m360_mlp_chain_bf16_workspace(M, L)) // 4 + 4, device=act0.device, dtype=torch.int32)
    wl = (C.c_void_p * L)(*[dev_bf16(w, "w").data_ptr() for w, _ in packs3])
    bl = (C.c_void_p * L)(*[dev(b, "b").data_ptr() for _, b in packs3])
    workspace_init(ws)
    opts = _opts_struct()
    _call("m360_mlp_chain_bf16x3_safe", x_in, act0, act1, M, ld, C.cast(wl, C.c_void_p), C.cast(bl, C.c_void_p), L, width, ws, C.byref(opts), STREAM)
    return (act0 if L % 2 == 0 else act1), workspace_status(ws)


def mlp_chain_bf16_safe(x_in, act0, act1, packs):
    """m360_mlp_chain_bf16_safe: layer 0 reads x_in (never written), layer j writes act[(j + 1) & 1]; a launch that reports an error is redone
    layer by layer by the gated launches queued behind it -> (result buffer, status dict of this call)."""
    x_in = dev_bf16(x_in, "x_in")
    act0, act1, M, ld, L, width, ws, wl, bl = _chain_args(act0, act1, packs)
    workspace_init(ws)
    import ctypes as C
    opts = _opts_struct()
    _call("m360_mlp_chain_bf16_safe", x_in, act0, act1, M, ld, wl, bl, L, width, ws, C.byref(opts), STREAM)
    return (act0 if L % 2 == 0 else act1), workspace_status(ws)


def linear_heads_bf16(x, w_packed, b_packed, head_w, store_y: bool = True, x3: bool = False, paired_in: bool = False):
    """The bf16 (x3 = False: m360_linear_heads_bf16) / bf16x3 (x3 = True: [hi | lo] pair rows, m360_linear_heads_bf16x3) last hidden
    layer with the heads formed on the matrix pipe -> (y, head_part[fused_rows, slots, heads] fp32, fused_rows)."""
    x, w_packed, b_packed, head_w = dev_bf16(x, "x"), dev_bf16(w_packed, "w_packed"), dev(b_packed, "b_packed"), dev(head_w, "head_w")
    M, ldx = x.shape
    n_pad = w_packed.shape[0]
    k_pad = w_packed.shape[1] // (3 if x3 else 1)
    heads = head_w.shape[0]
    if head_w.shape[1] != n_pad or ldx != (2 if x3 else 1) * k_pad:
        raise RuntimeError("linear_heads_bf16: shapes of x / w_packed / head_w do not match")
    lib = _lib.lib()
    mode = 2 if x3 else 1
    fused = int(lib.m360_linear_heads_fused_rows(M, n_pad, mode))
    slots = int(lib.m360_linear_heads_slots_bf16(n_pad, k_pad, mode, int(bool(store_y))))  # 2 or 8 per 256 columns: which kernel takes it
    ldy = (2 if x3 else 1) * n_pad
    y = torch.zeros(M, ldy, device=x.device, dtype=torch.bfloat16)
    part = torch.zeros(max(fused, 1), max(slots, 1), heads, device=x.device)
    _call("m360_linear_heads_bf16x3" if x3 else "m360_linear_heads_bf16", x, M, ldx, w_packed, b_packed, n_pad, k_pad,
          _lib.ACT_SIGMOID | (_lib.ROWS_PAIRED_IN if paired_in else 0), y, ldy, int(bool(store_y)), head_w, heads, part, STREAM)
    return y, part, fused


def nerf_finish_fused(act, head_part, fused_rows, head_w, head_b, density_bias, rgb_padding, t_vals, dirs, white_bkgd):
    act, head_part, head_w, head_b = dev(act, "act"), dev(head_part, "head_part"), dev(head_w, "head_w"), dev(head_b, "head_b")
    t_vals, dirs = dev(t_vals, "t_vals"), dev(dirs, "dirs")
    B, M = t_vals.shape
    N = M - 1
    d = act.device
    comp, dist, acc, w = torch.empty(B, 3, device=d), torch.empty(B, device=d), torch.empty(B, device=d), torch.empty(B, N, device=d)
    _call("m360_nerf_finish_fused", act, 0, act.shape[1], head_part, int(fused_rows), head_part.shape[1], head_w, head_b,
          head_w.shape[1], float(density_bias), float(rgb_padding), t_vals, dirs, B, N, int(bool(white_bkgd)), comp, dist, acc, w,
          STREAM)
    return comp, dist, acc, w


def nerf_finish_outputs(act, head_part, fused_rows, head_w, head_b, density_bias, rgb_padding, t_vals, dirs, near, far,
                        white_bkgd, near_far_calls: int = 1):
    """nerf_finish_fused that also returns t_vals + 1e-6 and s_vals = t_to_s(t_vals, near, far) from the same launch
    (m360_nerf_finish_outputs; model.py:194-196) -> (rgb, distance, acc, weights, t_vals_out, s_vals)."""
    act, head_part, head_w, head_b = dev(act, "act"), dev(head_part, "head_part"), dev(head_w, "head_w"), dev(head_b, "head_b")
    t_vals, dirs, near, far = dev(t_vals, "t_vals"), dev(dirs, "dirs"), dev(near, "near"), dev(far, "far")
    B, M = t_vals.shape
    N = M - 1
    d = act.device
    comp, dist, acc, w = torch.empty(B, 3, device=d), torch.empty(B, device=d), torch.empty(B, device=d), torch.empty(B, N, device=d)
    t_out, s_out = torch.empty_like(t_vals), torch.empty_like(t_vals)
    _call("m360_nerf_finish_outputs", act, 0, act.shape[1], head_part, int(fused_rows), head_part.shape[1], head_w, head_b,
          head_w.shape[1], float(density_bias), float(rgb_padding), t_vals, dirs, near, far, int(near_far_calls), B, N,
          int(bool(white_bkgd)), comp, dist, acc, w, t_out, s_out, STREAM)
    return comp, dist, acc, w, t_out, s_out


def prop_finish_fused(act, head_part, fused_rows, head_w, head_b, density_bias, t_vals, dirs, resample_padding):
    act, head_part, head_w, head_b = dev(act, "act"), dev(head_part, "head_part"), dev(head_w, "head_w"), dev(head_b, "head_b")
    t_vals, dirs = dev(t_vals, "t_vals"), dev(dirs, "dirs")
    B, M = t_vals.shape
    N = M - 1
    w, t_new = torch.empty(B, N, device=act.device), torch.empty_like(t_vals)
    _call("m360_prop_finish_fused", act, 0, act.shape[1], head_part, int(fused_rows), head_part.shape[1], head_w, head_b,
          head_w.numel(), float(density_bias), t_vals, dirs, None, B, N, M, float(resample_padding), w, t_new, STREAM)
    return w, t_new


# ----------------------------------------------------------------------------- ray generation
def generate_rays(cam_to_world, h: int, w: int, focal: float, near: float, far: float, ndc: bool = False,
                  ndc_near: float = 1.0, span=None):
    """-> 6 flattened device tensors (origins, directions, viewdirs [n*h*w,3]; radii, near, far [n*h*w,1]).
    `span=(first, end)`: only the rays of that flat pixel range (a rank's block of chunks of a sharded render)."""
    c2w = dev(cam_to_world, "cam_to_world")
    if c2w.dim() == 2:
        c2w = c2w[None]
    c2w = c2w[:, :3, :4].contiguous()
    n = c2w.shape[0]
    first, end = (0, n * h * w) if span is None else (int(span[0]), int(span[1]))
    total = end - first
    if total < 0:
        raise RuntimeError(f"generate_rays: empty-negative span {span}")
    d = c2w.device
    o, di, v = (torch.empty(total, 3, device=d) for _ in range(3))
    r, ne, fa = (torch.empty(total, 1, device=d) for _ in range(3))
    if first < 0 or end > n * h * w:
        raise RuntimeError(f"generate_rays: span [{first}, {end}) outside the {n * h * w} pixels of {n} camera(s)")
    if total:  # an empty span (a rank without chunks) launches nothing
        _call("m360_generate_rays_span", c2w, n, int(h), int(w), float(focal), float(near), float(far), int(bool(ndc)),
              float(ndc_near), first, total, o, di, v, r, ne, fa, STREAM)
    return o, di, v, r, ne, fa


def convert_to_ndc(origins, directions, focal: float, w: int, h: int, near: float = 1.0):
    origins, directions = dev(origins, "origins"), dev(directions, "directions")
    oo, do = torch.empty_like(origins), torch.empty_like(directions)
    _call("m360_convert_to_ndc", origins, directions, origins.numel() // 3, float(focal), int(w), int(h),
          float(near), oo, do, STREAM)
    return oo, do


# ----------------------------------------------------------------------------- visualisation
def _vis_ws(device) -> torch.Tensor:
    return torch.empty(_lib.lib().m360_visualize_workspace_bytes(), dtype=torch.uint8, device=device)


def depth_to_normals(depth) -> torch.Tensor:
    depth = dev(depth, "depth")
    h, w = depth.shape
    out = torch.empty(h, w, 3, device=depth.device)
    _call("m360_depth_to_normals", depth, h, w, out, STREAM)
    return out


def sinebow(hval) -> torch.Tensor:
    hval = dev(hval, "h")
    out = torch.empty(hval.shape + (3,), device=hval.device)
    _call("m360_sinebow", hval, hval.numel(), out, STREAM)
    return out


def visualize_normals(depth, acc=None) -> torch.Tensor:
    depth = dev(depth, "depth")
    acc = None if acc is None else dev(acc, "acc")
    h, w = depth.shape
    out = torch.empty(h, w, 3, device=depth.device)
    ws = _vis_ws(depth.device)
    _call("m360_visualize_normals", depth, acc, h, w, out, ws, ws.numel(), STREAM)
    return out


def visualize_depth(depth, acc=None, near=None, far=None, modulus: float = 0.0) -> torch.Tensor:
    """near / far falsy (None or 0, as in the reference's `near or ...`) -> taken from the depth map."""
    depth = dev(depth, "depth")
    acc = None if acc is None else dev(acc, "acc")
    h, w = depth.shape
    out = torch.empty(h, w, 3, device=depth.device)
    ws = _vis_ws(depth.device)
    _call("m360_visualize_depth", depth, acc, h, w, float(near or 0.0), float(far or 0.0), int(not near),
          int(not far), float(modulus), out, ws, ws.numel(), STREAM)
    return out


def visualize_depth_ex(depth, acc=None, near=None, far=None, ignore_frac: float = 0.0, curved: bool = False,
                       modulus: float = 0.0, want: str = "vis", near_auto: Optional[bool] = None,
                       far_auto: Optional[bool] = None):
    """m360_visualize_depth_ex.  want = "vis" -> colours [h,w,3]; "value" -> the colormap argument [h,w];
    "planes" -> the automatic (near, far) planes as a device float[2] (nothing rendered).
    near_auto / far_auto: choose that plane from the map.  Default None = the reference's `near or ...` rule (a falsy
    plane is automatic, intern/pose.py:176-177); pass False to use the given value even when it is 0.0 (a plane that a
    curve_fn mapped to exactly 0)."""
    depth = dev(depth, "depth")
    acc = None if acc is None else dev(acc, "acc")
    h, w = depth.shape
    d = depth.device
    ws = torch.empty(_lib.lib().m360_visualize_depth_ex_workspace_bytes(h, w), dtype=torch.uint8, device=d)
    vis = torch.empty(h, w, 3, device=d) if want == "vis" else None
    value = torch.empty(h, w, device=d) if want == "value" else None
    planes = torch.empty(2, device=d) if want == "planes" else None
    near_auto = (not near) if near_auto is None else bool(near_auto)
    far_auto = (not far) if far_auto is None else bool(far_auto)
    _call("m360_visualize_depth_ex", depth, acc, h, w, float(near or 0.0), float(far or 0.0), int(near_auto), int(far_auto),
          float(ignore_frac), int(bool(curved)), float(modulus), vis, value, planes, ws, ws.numel(), STREAM)
    return vis if want == "vis" else (value if want == "value" else planes)


def visualize_composite(colors, acc=None, depth=None) -> torch.Tensor:
    colors = dev(colors, "colors")
    h, w = colors.shape[:2]
    acc = None if acc is None else dev(acc, "acc")
    depth = None if depth is None else dev(depth, "depth")
    out = torch.empty(h, w, 3, device=colors.device)
    _call("m360_visualize_composite", colors, acc, depth, h, w, out, STREAM)
    return out


# ----------------------------------------------------------------------------- losses (row f3)
def _loss_ws(B: int, device, N: int = 0) -> torch.Tensor:
    return torch.empty(_lib.lib().m360_loss_workspace_bytes(int(B), int(N)), dtype=torch.uint8, device=device)


def loss_prop(t, w, t_hat, w_hat, want_grad: bool = False):
    """-> (loss[1], bounds[B,Np], grad_w_hat[B,Np] or None)"""
    t, w, t_hat, w_hat = dev(t, "t"), dev(w, "w"), dev(t_hat, "t_hat"), dev(w_hat, "w_hat")
    B, Nf = w.shape
    Np = w_hat.shape[1]
    d = t.device
    loss, bounds = torch.empty(1, device=d), torch.empty(B, Np, device=d)
    grad = torch.empty(B, Np, device=d) if want_grad else None
    ws = _loss_ws(B, d, Np)
    _call("m360_loss_prop", t, w, t_hat, w_hat, B, Nf, Np, bounds, loss, grad, ws,
          ws.numel(), STREAM)
    return loss, bounds, grad


def loss_prop_given_bounds(bounds, w_hat, want_grad: bool = False):
    """loss_prop with precomputed bounds -> (loss[1], grad_w_hat or None)"""
    bounds, w_hat = dev(bounds, "bounds"), dev(w_hat, "w_hat")
    B, Np = w_hat.shape
    loss = torch.empty(1, device=w_hat.device)
    grad = torch.empty(B, Np, device=w_hat.device) if want_grad else None
    ws = _loss_ws(B, w_hat.device, Np)
    _call("m360_loss_prop", None, bounds, None, w_hat, B, Np, Np, None, loss, grad, ws, ws.numel(),
          STREAM)
    return loss, grad


def loss_dist(s_vals, weights, want_grad: bool = False):
    """-> (loss[1], grad_w[B,N] or None, grad_s[B,N+1] or None)"""
    s_vals, weights = dev(s_vals, "s_vals"), dev(weights, "weights")
    B, N = weights.shape
    d = s_vals.device
    loss = torch.empty(1, device=d)
    gw = torch.empty(B, N, device=d) if want_grad else None
    gs = torch.empty(B, N + 1, device=d) if want_grad else None
    ws = _loss_ws(B, d)
    _call("m360_loss_dist", s_vals, weights, B, N, loss, gw, gs, ws, ws.numel(), STREAM)
    return loss, gw, gs


def loss_nerf(inp, target, want_grad: bool = False):
    """-> (out3 = [10 log10(mse) + 30, psnr, mse], grad_input[B,C] or None)"""
    inp, target = dev(inp, "input"), dev(target, "target")
    B, Cc = inp.shape
    out3 = torch.empty(3, device=inp.device)
    grad = torch.empty_like(inp) if want_grad else None
    ws = _loss_ws(B, inp.device)
    _call("m360_loss_nerf", inp, target, B, Cc, out3, grad, ws, ws.numel(), STREAM)
    return out3, grad
