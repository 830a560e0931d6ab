"""ORACLE — test infrastructure only, never the product path.

A vectorised torch-CPU (fp32) restatement of the ray-marching hot path of
zhangkai0425/mipnerf360, written from the math in SURVEY.md §8a. Each function
cites the reference file:line whose behaviour it reproduces. Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
module; `mipnerf360_amd/` must never do so.

Parity status: PINNED. `tests/golden/*.npz` were produced by importing the
reference itself in the authoring container (`tests/golden/make_golden.py`) and
`tests/test_oracle_golden.py` checks every function below against them
(<= 1e-6 abs on O(1) quantities) - since round 6 including the RANDOMIZED branches
(`jitter_t`, the `u_rand` branch of `sorted_piecewise_constant_pdf`, the stage forwards and the
training steps with draws): fixture G22 is the reference's own randomized run with what
`torch.rand` / `Tensor.uniform_` returned recorded beside its outputs.

Reference behaviours reproduced on purpose:
  * `contract()` uses the Frobenius norm of the WHOLE [B,N,3] tensor
    (intern/parameterization.py:23-29 called at :75),
  * the per-sample Jacobian loop (intern/parameterization.py:76-79) is replaced
    by its closed form  J = I                      if |y| <= 1
                        J = (2/r - 1/r^2) I + (2/r^4 - 2/r^3) y y^T   otherwise,
  * `g(x)` adds 1e-6 in place (intern/parameterization.py:15-21); this module
    never mutates its inputs but applies the same offsets.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, NamedTuple, Optional, Tuple

import numpy as np
import torch

EPS_G = 1e-6  # intern/parameterization.py:18


class Rays(NamedTuple):  # intern/ray.py:6
    origins: torch.Tensor
    directions: torch.Tensor
    viewdirs: torch.Tensor
    radii: torch.Tensor
    near: torch.Tensor
    far: torch.Tensor


def rays_from_numpy(d: Mapping[str, np.ndarray]) -> Rays:
    return Rays(*[torch.from_numpy(np.ascontiguousarray(d[k])).float().clone() for k in Rays._fields])


# 21 unit directions of the icosahedron/dodecahedron basis, intern/encoding.py:9-30
_A, _B, _C, _D = 0.8506508, 0.5257311, 0.809017, 0.309017
_IPE_ROWS = [
    [_A, 0, _B], [_C, 0.5, _D], [_B, _A, 0], [1, 0, 0], [_C, 0.5, -_D], [_A, 0, -_B],
    [_D, _C, -0.5], [0, _B, -_A], [0.5, _D, -_C], [0, 1, 0], [-_B, _A, 0], [-_D, _C, -0.5],
    [0, _B, _A], [-_D, _C, 0.5], [_D, _C, 0.5], [0.5, _D, _C], [0.5, -_D, _C], [0, 0, 1],
    [-0.5, _D, _C], [-_C, 0.5, _D], [-_C, 0.5, -_D]]
IPE_BASIS = torch.tensor(_IPE_ROWS, dtype=torch.float32)


def ipe_basis(dtype=torch.float32):
    """the basis in the dtype of the data (the reference builds it with torch.tensor(<python floats>): in its fp64 runs -
    tests/golden/make_golden.py reference_in_fp64 - the rows are the doubles nearest to the decimal literals)"""
    return IPE_BASIS if dtype == torch.float32 else torch.tensor(_IPE_ROWS, dtype=dtype)


# --------------------------------------------------------------------------- t <-> s
def disparity_eps(x: torch.Tensor, times: int = 1) -> torch.Tensor:
    """1/(x + times*1e-6): value `g` returns on its `times`-th call on the same tensor
    (intern/parameterization.py:15-21, in-place `x += eps`)."""
    y = x
    for _ in range(times):
        y = y + EPS_G
    return 1.0 / y


def sample_t(near: torch.Tensor, far: torch.Tensor, num_samples: int) -> torch.Tensor:
    """Deterministic t_vals [B,N+1] of intern/ray.py:99-101,110:
    t = g(s*g(far) + (1-s)*g(near)), each g adding 1e-6 first."""
    s = torch.linspace(0.0, 1.0, num_samples + 1)
    mix = s * disparity_eps(far) + (1 - s) * disparity_eps(near)
    return 1.0 / (mix + EPS_G)


def jitter_t(t_vals: torch.Tensor, t_rand: torch.Tensor) -> torch.Tensor:
    """Stratified jitter of intern/ray.py:103-108 with caller-supplied uniforms."""
    mids = 0.5 * (t_vals[..., 1:] + t_vals[..., :-1])
    upper = torch.cat([mids, t_vals[..., -1:]], -1)
    lower = torch.cat([t_vals[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def t_to_s(t_vals, near, far, near_calls_before: int = 1, far_calls_before: int = 1):
    """s_vals of intern/parameterization.py:5-8 as evaluated inside
    nerf_net.forward (model.py:196): by then `g` has already been applied
    `*_calls_before` times to near/far by sample_along_rays, and t_to_s itself
    calls g(t), g(near), g(far), g(near) in that order."""
    gt = disparity_eps(t_vals, 1)
    gn1 = disparity_eps(near, near_calls_before + 1)
    gf = disparity_eps(far, far_calls_before + 1)
    gn2 = disparity_eps(near, near_calls_before + 2)
    return (gt - gn1) / (gf - gn2)


# --------------------------------------------------------------------------- frustum -> gaussian
def frustum_moments(t0, t1, radii):
    """Stable conical-frustum moments, intern/parameterization.py:99-107."""
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    den = 3 * mu ** 2 + hw ** 2
    t_mean = mu + (2 * mu * hw ** 2) / den
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / den ** 2)
    r_var = radii ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 * (hw ** 4) / den)
    return t_mean, t_var, r_var


def frustum_moments_unstable(t0, t1, radii):
    """Direct moment formulas, intern/parameterization.py:108-113 (stable=False)."""
    t_mean = (3 * (t1 ** 4 - t0 ** 4)) / (4 * (t1 ** 3 - t0 ** 3))
    r_var = radii ** 2 * (3 / 20 * (t1 ** 5 - t0 ** 5) / (t1 ** 3 - t0 ** 3))
    t_mosq = 3 / 5 * (t1 ** 5 - t0 ** 5) / (t1 ** 3 - t0 ** 3)
    return t_mean, t_mosq - t_mean ** 2, r_var


def lift_to_xyz_diag(d, t_mean, t_var, r_var):
    """Diagonal lift, intern/parameterization.py:44-54 (diag=True) -> (mean[B,N,3], cov_diag[B,N,3])."""
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.clamp_min(torch.sum(d ** 2, dim=-1, keepdim=True), 1e-10)
    d_outer_diag = d ** 2
    null_outer_diag = 1 - d_outer_diag / d_mag_sq
    return mean, t_var[..., None] * d_outer_diag[..., None, :] + r_var[..., None] * null_outer_diag[..., None, :]


def lift_to_xyz(d, t_mean, t_var, r_var):
    """Full-covariance lift, intern/parameterization.py:44-46,55-62 (diag=False)."""
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.clamp_min(torch.sum(d ** 2, dim=-1, keepdim=True), 1e-10)
    d_outer = d[..., :, None] * d[..., None, :]
    eye = torch.eye(3)
    null_outer = eye - d[..., :, None] * (d / d_mag_sq)[..., None, :]
    cov = t_var[..., None, None] * d_outer[..., None, :, :] + r_var[..., None, None] * null_outer[..., None, :, :]
    return mean, cov


def contract_global(x: torch.Tensor, norm=None) -> torch.Tensor:
    """intern/parameterization.py:23-29 — norm over the whole tensor (`norm`: the value to use instead, for a batch
    that is only a shard of the tensor the reference would see)."""
    n = torch.linalg.vector_norm(x) if norm is None else norm
    if n <= 1:
        return x
    return (2 - 1 / n) * (x / n)


def contract_jacobian(y: torch.Tensor) -> torch.Tensor:
    """Closed form of jacobian(contract, y) for 3-vectors y[...,3]
    (what intern/parameterization.py:76-79 computes per sample)."""
    r = torch.linalg.vector_norm(y, dim=-1)[..., None, None]
    eye = torch.eye(3).expand(y.shape[:-1] + (3, 3))
    rs = torch.where(r > 1, r, torch.ones_like(r))
    a = 2 / rs - 1 / rs ** 2
    b = 2 / rs ** 4 - 2 / rs ** 3
    J = a * eye + b * (y[..., :, None] * y[..., None, :])
    return torch.where(r > 1, J, eye)


def gaussian_contract(mean, cov, norm=None):
    """intern/parameterization.py:64-83."""
    mean_c = contract_global(mean, norm)
    J = contract_jacobian(mean_c)
    cov_c = torch.matmul(torch.matmul(J, cov), J.transpose(-1, -2))
    return mean_c, cov_c


def para_rays(t_vals, origins, directions, radii, norm=None):
    """intern/parameterization.py:119-135 (origins added AFTER contraction)."""
    t0, t1 = t_vals[..., :-1], t_vals[..., 1:]
    t_mean, t_var, r_var = frustum_moments(t0, t1, radii)
    mean, cov = lift_to_xyz(directions, t_mean, t_var, r_var)
    mean, cov = gaussian_contract(mean, cov, norm)
    return mean + origins[..., None, :], cov


def mean_sumsq(t_vals, directions, radii) -> torch.Tensor:
    """fp64 sum of squares of the un-contracted means of a shard: what ranks all-reduce to rebuild the global norm."""
    t_mean, t_var, r_var = frustum_moments(t_vals[..., :-1], t_vals[..., 1:], radii)
    mean, _ = lift_to_xyz(directions, t_mean, t_var, r_var)
    return mean.double().square().sum().reshape(1)


# --------------------------------------------------------------------------- encodings
def ipe(mean, cov):
    """intern/encoding.py:33-56 (single-scale integrated positional encoding)."""
    P = ipe_basis(mean.dtype)
    gamma = torch.matmul(mean, P.T)                     # [...,21]
    A = torch.matmul(cov, P.T)                          # [...,3,21]
    sigma = torch.sum(P.T * A, dim=-2)                  # [...,21]
    damp = torch.exp(-0.5 * sigma)
    return torch.cat((damp * torch.sin(gamma), damp * torch.cos(gamma)), -1)


def viewdir_enc(viewdirs, min_deg: int = 0, max_deg: int = 4):
    """intern/encoding.py:69-90 (theta/phi encoding; atan, not atan2)."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)], dtype=viewdirs.dtype)
    x, y, z = viewdirs[..., 0:1], viewdirs[..., 1:2], viewdirs[..., 2:3]
    theta = torch.arccos(z) * scales
    phi = torch.arctan(y / (x + 1e-6)) * scales
    return torch.cat((torch.sin(theta), torch.cos(theta), torch.sin(phi), torch.cos(phi)), -1)


def encode_inputs(mean, cov, viewdirs, min_deg=0, max_deg=4):
    """model.py:85-88 / :173-176 — [B,N,42+16]."""
    enc = ipe(mean, cov)
    vd = viewdir_enc(viewdirs, min_deg, max_deg)[:, None, :].expand(-1, enc.shape[1], -1)
    return torch.cat((enc, vd), -1)


# --------------------------------------------------------------------------- weights / resampling / compositing
def density_to_weight(t_vals, density, dirs):
    """model.py:59-78. density [B,N,1] or [B,N]."""
    if density.dim() == 3:
        density = density[..., 0]
    delta = (t_vals[..., 1:] - t_vals[..., :-1]) * torch.linalg.norm(dirs[..., None, :], dim=-1)
    x = density * delta
    alpha = 1 - torch.exp(-x)
    excl = torch.cat([torch.zeros_like(x[..., :1]), torch.cumsum(x[..., :-1], dim=-1)], dim=-1)
    return alpha * torch.exp(-excl)


def blur_weights(weights, resample_padding):
    """intern/ray.py:137-142."""
    wp = torch.cat([weights[..., :1], weights, weights[..., -1:]], dim=-1)
    wm = torch.maximum(wp[..., :-1], wp[..., 1:])
    return 0.5 * (wm[..., :-1] + wm[..., 1:]) + resample_padding


def sorted_piecewise_constant_pdf(bins, weights, num_samples, u_rand: Optional[torch.Tensor] = None, wsum_ulps: int = 0):
    """intern/ray.py:12-57. Deterministic u unless `u_rand` ([B,num_samples]
    uniforms in [0,1)) is given, in which case the randomized branch (:30-35,
    including its `u + u` doubling) is followed with those uniforms.
    The O(N^2) mask is replaced by a sorted search (same bracket: the last
    cdf_i <= u).
    wsum_ulps (conditioning probe for tests, 0 = the reference): the weight sum moved by that many fp32 ulps before it is
    used - what another summation order of torch.sum could return.  Where a flat stretch of the cdf sits within an ulp of a
    u (the randomized branch clamps HALF of its u to 1 - eps, so a cdf that saturates at ~1 is such a stretch) the bracket,
    and with it the sample, depends on that ulp: such samples are not defined by the reference beyond this set of answers."""
    eps = 1e-5
    f32eps = float(torch.finfo(torch.float32).eps)
    wsum = torch.sum(weights, dim=-1, keepdim=True)
    for _ in range(abs(int(wsum_ulps))):
        wsum = torch.nextafter(wsum, torch.full_like(wsum, float("inf") if wsum_ulps > 0 else float("-inf")))
    pad = torch.maximum(torch.zeros_like(wsum), eps - wsum)  # NaN-propagating, like the reference's torch.maximum
    weights = weights + pad / weights.shape[-1]
    wsum = wsum + pad
    pdf = weights / wsum
    cdf = torch.cumsum(pdf[..., :-1], dim=-1)
    cdf = torch.minimum(torch.ones_like(cdf), cdf)
    edge = torch.zeros_like(wsum)  # [...,1] even when there is a single interval (empty cumsum)
    cdf = torch.cat([edge, cdf, edge + 1.0], dim=-1)
    if u_rand is None:
        u = torch.linspace(0.0, 1.0 - f32eps, num_samples)
        u = u.expand(cdf.shape[:-1] + (num_samples,)).contiguous()
    else:
        s = 1 / num_samples
        base = (torch.arange(num_samples) * s)[None, :]
        u = base + base + u_rand * (s - f32eps)
        u = torch.clamp_max(u, 1.0 - f32eps)
    idx = torch.searchsorted(torch.nan_to_num(cdf, nan=2.0).contiguous(), u, right=True) - 1
    last = cdf.shape[-1] - 1
    i0 = idx.clamp(0, last)
    i1 = (idx + 1).clamp(0, last)
    b0, b1 = torch.gather(bins, -1, i0), torch.gather(bins, -1, i1)
    c0, c1 = torch.gather(cdf, -1, i0), torch.gather(cdf, -1, i1)
    # Rows whose cdf is not a finite non-decreasing sequence (NaN / Inf / negative weights; fixture G18): the sorted
    # search does not describe the reference there.  Its bracket is defined through the comparison table
    # m[i,j] = (u_j >= cdf_i) (intern/ray.py:43-50): x0_j = max over i of (x_i where m, else x_0), x1_j = min over i of
    # (x_i where not m, else x_last) - with NaN comparing false and torch.max / torch.min propagating NaN.
    bad = ~(torch.isfinite(cdf).all(-1) & (cdf[..., 1:] >= cdf[..., :-1]).all(-1))
    if bool(bad.any()):
        cb, bb, ub = cdf[bad], bins[bad], u[bad]
        m = ub[:, None, :] >= cb[:, :, None]

        def bracket(x):
            lo = torch.where(m, x[:, :, None], x[:, :1, None]).max(dim=-2).values
            hi = torch.where(~m, x[:, :, None], x[:, -1:, None]).min(dim=-2).values
            return lo, hi

        b0, b1, c0, c1 = b0.clone(), b1.clone(), c0.clone(), c1.clone()
        (b0[bad], b1[bad]), (c0[bad], c1[bad]) = bracket(bb), bracket(cb)
    tt = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    return b0 + tt * (b1 - b0)


def resample_t(t_vals, weights, resample_padding, u_rand=None, num_out=None):
    """intern/ray.py:136-149 (new t only).  num_out: extension for unequal proposal / NeRF sample counts
    (the reference always draws t_vals.shape[-1] values, :147)."""
    w = blur_weights(weights, resample_padding)
    return sorted_piecewise_constant_pdf(t_vals, w, num_out or t_vals.shape[-1], u_rand)


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """intern/ray.py:155-191."""
    t_mids = 0.5 * (t_vals[..., :-1] + t_vals[..., 1:])
    weights = density_to_weight(t_vals, density, dirs)
    comp_rgb = (weights[..., None] * rgb).sum(dim=-2)
    acc = weights.sum(dim=-1)
    distance = (weights * t_mids).sum(dim=-1) / acc
    distance = torch.clamp(torch.nan_to_num(distance), t_vals[:, 0], t_vals[:, -1])
    if white_bkgd:
        comp_rgb = comp_rgb + (1.0 - acc[..., None])
    return comp_rgb, distance, acc, weights


# --------------------------------------------------------------------------- MLPs
def _lin(x, sd, prefix):
    return torch.nn.functional.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _r16(x):
    """round-to-nearest-even to bf16 and back: emulates the opt-in bf16 MLP's storage precision."""
    return x.bfloat16().float()


def _lin16(x, sd, prefix):
    """bf16-rounded inputs and weights, fp32 products/accumulation, fp32 bias (the bf16 MFMA contract)."""
    return torch.nn.functional.linear(x, _r16(sd[prefix + ".weight"]), sd[prefix + ".bias"])


def _split16(x):
    """two bf16 terms of an fp32 tensor: hi = bf16(x), lo = bf16(x - hi) (the build's "bf16x3" mode, an extension)"""
    hi = _r16(x)
    return hi, _r16(x - hi)


def _x3(x):
    """what a bf16x3 layer stores and the next one reads: hi + lo (16 significant bits)"""
    hi, lo = _split16(x)
    return hi + lo


def _lin16x3(x, sd, prefix):
    """bf16x3 contract: x w = xh wh + xl wh + xh wl with exact bf16 products and fp32 accumulation, fp32 bias"""
    xh, xl = _split16(x)
    wh, wl = _split16(sd[prefix + ".weight"])
    return (xh @ wh.t() + xl @ wh.t() + xh @ wl.t()) + sd[prefix + ".bias"]


def _split16x3(x):
    """three bf16 terms hi + mid + lo = all 24 bits of an fp32 value (a non-finite hi carries the whole value)"""
    hi = _r16(x)
    r = torch.where(torch.isfinite(hi), x - hi, torch.zeros_like(x))
    mid = _r16(r)
    return hi, mid, _r16(r - mid)


def _lin16x6(x, sd, prefix):
    """first layer of the build's bf16 / bf16x3 modes ("x6", include/m360.h): features and weights as three bf16 terms each, six
    exact bf16 products xl wh + xm wm + xh wl + xm wh + xh wm + xh wh with fp32 accumulation, fp32 bias - the fp32 product up to
    2^-24 terms"""
    xh, xm, xl = _split16x3(x)
    wh, wm, wl = _split16x3(sd[prefix + ".weight"])
    return (xl @ wh.t() + xm @ wm.t() + xh @ wl.t() + xm @ wh.t() + xh @ wm.t() + xh @ wh.t()) + sd[prefix + ".bias"]


def prop_mlp(x, sd, bf16=False):
    """model.py:43-53.  bf16=True emulates the reduced-precision extension (hidden activations stored as bf16), bf16=2 the
    bf16x3 extension (two bf16 terms per value, three products per multiply); the first layers see more bits than the hidden ones:
    bf16: `_lin16x3` (16 bits of features and weights) with one bf16 term out; bf16x3: `_lin16x6` (all 24 bits)."""
    if bf16 == 2:
        x = _x3(torch.relu(_lin16x6(x, sd, "prop_net.model.0")))
        for i in (2, 4):
            x = _x3(torch.relu(_lin16x3(x, sd, f"prop_net.model.{i}")))
        x = _x3(torch.sigmoid(_lin16x3(x, sd, "prop_net.model.6")))
        return _lin(x, sd, "prop_net.model.8")
    if bf16:
        x = _r16(torch.relu(_lin16x3(x, sd, "prop_net.model.0")))   # the bf16 mode's first layer: two terms in, one out
        for i in (2, 4):
            x = _r16(torch.relu(_lin16(x, sd, f"prop_net.model.{i}")))
        x = _r16(torch.sigmoid(_lin16(x, sd, "prop_net.model.6")))
        return _lin(x, sd, "prop_net.model.8")
    for i in (0, 2, 4):
        x = torch.relu(_lin(x, sd, f"prop_net.model.{i}"))
    x = torch.sigmoid(_lin(x, sd, "prop_net.model.6"))
    return _lin(x, sd, "prop_net.model.8")


def nerf_mlp(x, sd, bf16=False):
    """model.py:131-158 (bf16: see prop_mlp)."""
    if bf16 == 2:
        x = _x3(torch.relu(_lin16x6(x, sd, "nerf_net.model.0")))
        for i in range(2, 14, 2):
            x = _x3(torch.relu(_lin16x3(x, sd, f"nerf_net.model.{i}")))
        x = _x3(torch.sigmoid(_lin16x3(x, sd, "nerf_net.model.14")))
    elif bf16:
        x = _r16(torch.relu(_lin16x3(x, sd, "nerf_net.model.0")))
        for i in range(2, 14, 2):
            x = _r16(torch.relu(_lin16(x, sd, f"nerf_net.model.{i}")))
        x = _r16(torch.sigmoid(_lin16(x, sd, "nerf_net.model.14")))
    else:
        for i in range(0, 14, 2):
            x = torch.relu(_lin(x, sd, f"nerf_net.model.{i}"))
        x = torch.sigmoid(_lin(x, sd, "nerf_net.model.14"))
    raw_density = torch.sigmoid(_lin(x, sd, "nerf_net.final_density.0"))
    raw_rgb = torch.sigmoid(_lin(x, sd, "nerf_net.final_color.0"))
    return raw_density, raw_rgb


def to_torch_state_dict(sd: Mapping[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: (torch.from_numpy(np.ascontiguousarray(v)).float() if isinstance(v, np.ndarray) else v.detach().float().cpu())
            for k, v in sd.items()}


# --------------------------------------------------------------------------- stages
class Hyper(NamedTuple):
    num_samples: int = 128
    density_bias: float = -1.0
    rgb_padding: float = 0.001
    resample_padding: float = 0.01
    white_bkgd: bool = False
    viewdir_min_deg: int = 0
    viewdir_max_deg: int = 4
    num_samples_fine: int = 0  # extension ("64+128"); 0 = num_samples, the reference's behaviour
    mlp_bf16: int = 0          # extension: 1 = emulate the bf16 MLP mode (bf16 storage, fp32 accumulation), 2 = bf16x3


def prop_forward(rays: Rays, sd, hp: Hyper, t_rand=None):
    """prop_net.forward, model.py:80-94 -> (t_vals[B,N+1], weights[B,N])."""
    t_vals = sample_t(rays.near, rays.far, hp.num_samples)
    if t_rand is not None:
        t_vals = jitter_t(t_vals, t_rand)
    t_vals = t_vals.expand(rays.origins.shape[0], -1).contiguous()
    mean, cov = para_rays(t_vals, rays.origins, rays.directions, rays.radii)
    x = encode_inputs(mean, cov, rays.viewdirs, hp.viewdir_min_deg, hp.viewdir_max_deg)
    raw = prop_mlp(x, sd, hp.mlp_bf16)
    density = torch.nn.functional.softplus(raw + hp.density_bias)
    return t_vals, density_to_weight(t_vals, density, rays.directions)


def nerf_forward(rays: Rays, t_vals, coarse_weights, sd, hp: Hyper, u_rand=None):
    """nerf_net.forward, model.py:163-200 -> (rgb, dist, acc, t_vals+1e-6, weights, s_vals)."""
    t_new = resample_t(t_vals, coarse_weights, hp.resample_padding, u_rand,
                       (hp.num_samples_fine + 1) if hp.num_samples_fine else None)
    mean, cov = para_rays(t_new, rays.origins, rays.directions, rays.radii)
    x = encode_inputs(mean, cov, rays.viewdirs, hp.viewdir_min_deg, hp.viewdir_max_deg)
    raw_density, raw_rgb = nerf_mlp(x, sd, hp.mlp_bf16)
    rgb = raw_rgb * (1 + 2 * hp.rgb_padding) - hp.rgb_padding
    density = torch.nn.functional.softplus(raw_density + hp.density_bias)
    comp_rgb, distance, acc, weights = volumetric_rendering(rgb, density, t_new, rays.directions, hp.white_bkgd)
    s_vals = t_to_s(t_new, rays.near, rays.far)
    return comp_rgb, distance, acc, t_new + EPS_G, weights, s_vals


def prop_forward_from_t(rays: Rays, sd, hp: Hyper, t_vals, norm):
    """proposal stage of a shard: given sample positions and the global norm -> (weights, resampled t)."""
    mean, cov = para_rays(t_vals, rays.origins, rays.directions, rays.radii, norm)
    x = encode_inputs(mean, cov, rays.viewdirs, hp.viewdir_min_deg, hp.viewdir_max_deg)
    density = torch.nn.functional.softplus(prop_mlp(x, sd, hp.mlp_bf16) + hp.density_bias)
    w = density_to_weight(t_vals, density, rays.directions)
    return w, resample_t(t_vals, w, hp.resample_padding, None, (hp.num_samples_fine + 1) if hp.num_samples_fine else None)


def nerf_forward_from_t(rays: Rays, sd, hp: Hyper, t_new, norm):
    """NeRF stage of a shard on given resampled positions -> (rgb, distance, acc)."""
    mean, cov = para_rays(t_new, rays.origins, rays.directions, rays.radii, norm)
    x = encode_inputs(mean, cov, rays.viewdirs, hp.viewdir_min_deg, hp.viewdir_max_deg)
    raw_density, raw_rgb = nerf_mlp(x, sd, hp.mlp_bf16)
    rgb = raw_rgb * (1 + 2 * hp.rgb_padding) - hp.rgb_padding
    density = torch.nn.functional.softplus(raw_density + hp.density_bias)
    comp_rgb, distance, acc, _ = volumetric_rendering(rgb, density, t_new, rays.directions, hp.white_bkgd)
    return comp_rgb, distance, acc


def forward(rays: Rays, sd, hp: Hyper):
    """mipNeRF360.forward, model.py:247-252."""
    with torch.no_grad():
        t_hat, w_hat = prop_forward(rays, sd, hp)
        out = nerf_forward(rays, t_hat, w_hat, sd, hp)
    return out[0], out[1], out[2]


def to8b(img: np.ndarray) -> np.ndarray:
    """intern/utils.py:17-20."""
    return (255 * np.clip(np.nan_to_num(img), 0, 1)).astype(np.uint8)


def render_image(rays: Rays, height: int, width: int, sd, hp: Hyper, chunks: int = 4096):
    """mipNeRF360.render_image, model.py:254-274 (chunk partition preserved —
    the global contraction norm makes results chunk-dependent)."""
    n = rays.origins.shape[0]
    rgbs, dists, accs = [], [], []
    for i in range(0, n, chunks):
        chunk = Rays(*[f[i:i + chunks] for f in rays])
        r, d, a = forward(chunk, sd, hp)
        rgbs.append(r), dists.append(d), accs.append(a)
    rgb8 = to8b(torch.cat(rgbs, 0).reshape(height, width, 3).numpy())
    return rgb8, torch.cat(dists, 0).reshape(height, width).numpy(), torch.cat(accs, 0).reshape(height, width).numpy()


# --------------------------------------------------------------------------- ray generation (row f1)
def convert_to_ndc(origins: np.ndarray, directions: np.ndarray, focal, w, h, near=1.0):
    """intern/ray.py:59-79 (NumPy, fp32 arrays with Python-float scalars like the reference)."""
    t = -(near + origins[..., 2]) / (directions[..., 2] + 1e-15)
    p = origins + t[..., None] * directions
    iz = p[..., 2] + 1e-15
    dzr = directions[..., 2] + 1e-15
    sx, sy = (2 * focal) / w, (2 * focal) / h
    o = np.stack([-sx * (p[..., 0] / iz), -sy * (p[..., 1] / iz), 1 + 2 * near / iz], -1)
    d = np.stack([-sx * (directions[..., 0] / dzr - p[..., 0] / iz), -sy * (directions[..., 1] / dzr - p[..., 1] / iz),
                  -2 * near / iz], -1)
    return o, d


def _neighbour_dist(m: np.ndarray, axis: int) -> np.ndarray:
    """||m[i] - m[i+1]|| along `axis` (n-1 entries) padded to n with the SECOND-to-last difference: the
    reference appends `dx[:, -2:-1]`, i.e. entry n-3 of the n-1 differences (dataset.py:129-131)."""
    a = np.take(m, range(0, m.shape[axis] - 1), axis=axis)
    b = np.take(m, range(1, m.shape[axis]), axis=axis)
    dist = np.sqrt(np.sum((a - b) ** 2, -1))
    last = np.take(dist, [dist.shape[axis] - 2], axis=axis)
    return np.concatenate([dist, last], axis)


def generate_rays(cam_to_world: np.ndarray, h: int, w: int, focal, near, far, ndc: bool = False) -> Dict[str, np.ndarray]:
    """dataset.py:109-145 (pinhole) and, for ndc=True, dataset.py:364-387; flattened to [n*h*w, .]."""
    c2w = np.asarray(cam_to_world, dtype=np.float32)[:, :3, :4]
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="xy")
    cam = np.stack([(x - w * 0.5 + 0.5) / focal, -(y - h * 0.5 + 0.5) / focal, -np.ones_like(x)], -1)
    dirs = (cam[None, :, :, None, :] * c2w[:, None, None, :, :3]).sum(-1)
    origins = np.broadcast_to(c2w[:, None, None, :, 3], dirs.shape)
    viewdirs = dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
    if ndc:
        origins, dirs = convert_to_ndc(origins, dirs, focal, w, h)
        spread = 0.5 * (_neighbour_dist(origins, 1) + _neighbour_dist(origins, 2))
    else:
        spread = _neighbour_dist(dirs, 1)
    radii = spread[..., None] * 2 / np.sqrt(12)
    ones = np.ones_like(radii)
    out = dict(origins=origins, directions=dirs, viewdirs=viewdirs, radii=radii, near=ones * near, far=ones * far)
    return {k: np.ascontiguousarray(v, dtype=np.float32).reshape(-1, v.shape[-1]) for k, v in out.items()}


# --------------------------------------------------------------------------- visualisation (row f2)
def _conv3_same(z: np.ndarray, k: np.ndarray) -> np.ndarray:
    """scipy.signal.convolve2d(z, k, mode='same') for a 3x3 kernel: true convolution, zero fill."""
    zp = np.pad(z, 1)
    out = np.zeros_like(z, dtype=np.result_type(z, k))
    for i in range(3):
        for j in range(3):
            out += k[i, j] * zp[2 - i:2 - i + z.shape[0], 2 - j:2 - j + z.shape[1]]
    return out


def depth_to_normals(depth: np.ndarray) -> np.ndarray:
    """intern/pose.py:112-121."""
    blur, edge = np.array([1, 2, 1]) / 4, np.array([-1, 0, 1]) / 2
    dy = _conv3_same(depth, blur[None, :] * edge[:, None])
    dx = _conv3_same(depth, blur[:, None] * edge[None, :])
    inv = 1 / np.sqrt(1 + dx ** 2 + dy ** 2)
    return np.stack([dx * inv, dy * inv, inv], -1)


def sinebow(h):
    """intern/pose.py:122-125."""
    f = lambda x: np.sin(np.pi * x) ** 2  # noqa: E731
    return np.stack([f(3 / 6 - h), f(5 / 6 - h), f(7 / 6 - h)], -1)


def visualize_normals(depth: np.ndarray, acc: Optional[np.ndarray]) -> np.ndarray:
    """intern/pose.py:127-146 (scaling=None)."""
    mask = ~np.isnan(depth)
    x, y = np.meshgrid(np.arange(depth.shape[1]), np.arange(depth.shape[0]), indexing="xy")
    scaling = np.sqrt((np.var(x[mask]) + np.var(y[mask])) / 2 / np.var(depth[mask].astype(np.float64)))
    normals = depth_to_normals(scaling * depth.astype(np.float64))
    vis = np.isnan(normals) + np.nan_to_num((normals + 1) / 2, nan=0)
    if acc is not None:
        vis = vis * acc[:, :, None] + (1 - acc)[:, :, None]
    return vis


def turbo_lut() -> np.ndarray:
    """The 256 x 3 'turbo' table the reference gets from matplotlib (cm.get_cmap('turbo'), intern/pose.py:197)."""
    import matplotlib
    return np.asarray(matplotlib.colormaps["turbo"](np.arange(256))[:, :3])


def visualize_depth(depth: np.ndarray, acc=None, near=None, far=None, modulus: float = 0.0, ignore_frac: float = 0.0,
                    curve_fn=None, colormap=None) -> np.ndarray:
    """intern/pose.py:148-212 (default curve -log(x + eps), default colormaps turbo / sinebow unless given)."""
    eps = np.finfo(np.float32).eps
    acc = np.ones_like(depth) if acc is None else acc
    acc = np.where(np.isnan(depth), np.zeros_like(acc), acc)
    order = np.argsort(depth.reshape(-1), kind="stable")  # NaNs last, like the reference's argsort
    flat = depth.reshape(-1)[order]
    cum = np.cumsum(acc.reshape(-1)[order])
    keep = flat[(cum >= cum[-1] * ignore_frac) & (cum <= cum[-1] * (1 - ignore_frac))]
    near = near or keep[0] - eps
    far = far or keep[-1] + eps
    curve = curve_fn or (lambda v: -np.log(v + eps))  # noqa: E731
    d, cn, cf = curve(depth), curve(near), curve(far)
    if modulus > 0:
        value = np.mod(d, modulus) / modulus
        rgb = colormap(value)[:, :, :3] if colormap else sinebow(value)
    else:
        value = np.nan_to_num(np.clip((d - np.minimum(cn, cf)) / np.abs(cf - cn), 0, 1))
        if colormap:
            rgb = colormap(value)[:, :, :3]
        else:
            idx = np.minimum((value * 256).astype(int), 255)
            rgb = turbo_lut()[idx]
    return rgb * acc[:, :, None] + (1 - acc)[:, :, None]


# ------------------------------------------------------------------------------------------ losses (row f3)
def prop_bounds(t: torch.Tensor, w: torch.Tensor, t_hat: torch.Tensor) -> torch.Tensor:
    """intern/distillation.py:4-33.  The mask at :29 is [B, Nf] and indexes `fine_weights[..., mask]`, which flattens
    over the rays: bounds[:, i] is the overlap sum of the WHOLE BATCH, the same for every ray (no gradient)."""
    t0, t1, lo, hi = t[..., :-1], t[..., 1:], t_hat[..., :-1], t_hat[..., 1:]
    overlap = ~((t0[:, None, :] > hi[:, :, None]) | (t1[:, None, :] < lo[:, :, None]))  # [B, Np, Nf]
    per_ray = (overlap * w[:, None, :].double()).sum(-1)                                   # [B, Np]
    return per_ray.sum(0, keepdim=True).float().expand(w.shape[0], -1).contiguous().detach()


def loss_prop_given(w_hat: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """intern/distillation.py:35-51."""
    return (torch.relu(bounds - w_hat).square() / (w_hat + 1e-6)).sum() / bounds.shape[0]


def loss_prop(t, w, t_hat, w_hat) -> torch.Tensor:
    """intern/loss.py:6-21."""
    return loss_prop_given(w_hat, prop_bounds(t, w, t_hat))


def loss_prop_grad(w_hat: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """closed form of d loss_prop / d w_hat (what autograd returns for distillation.py:48-49)."""
    r, den = torch.relu(bounds - w_hat), w_hat + 1e-6
    return -(2 * r * den + r * r) / (den * den) / bounds.shape[0]


def loss_dist(s_vals: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """intern/regularization.py:3-19 (the double Python loop as one outer difference), summed over rays."""
    m = (s_vals[..., :-1] + s_vals[..., 1:]) / 2
    pair = (weights[:, :, None] * weights[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum()
    return pair + (1 / 3) * (weights ** 2 * (s_vals[..., 1:] - s_vals[..., :-1])).sum()


def loss_dist_grads(s_vals: torch.Tensor, weights: torch.Tensor):
    """closed-form gradients of loss_dist -> (grad_s [B, N+1], grad_w [B, N]); sign(0) = 0 like torch.abs'."""
    m = (s_vals[..., :-1] + s_vals[..., 1:]) / 2
    d = m[:, :, None] - m[:, None, :]
    ds = s_vals[..., 1:] - s_vals[..., :-1]
    grad_w = 2 * (weights[:, None, :] * d.abs()).sum(-1) + (2 / 3) * weights * ds
    gm = 2 * weights * (weights[:, None, :] * torch.sign(d)).sum(-1)
    grad_s = torch.zeros_like(s_vals)
    grad_s[:, :-1] += 0.5 * gm - (1 / 3) * weights ** 2
    grad_s[:, 1:] += 0.5 * gm + (1 / 3) * weights ** 2
    return grad_s, grad_w


def mse_to_psnr(mse):
    """intern/loss.py:57-59."""
    return -10.0 * torch.log10(mse)


def loss_nerf(inp: torch.Tensor, target: torch.Tensor):
    """intern/loss.py:23-40 -> (10 log10(mse) + 30, psnr); mse is the squared error summed over rgb / batch."""
    mse = ((inp[..., :3] - target[..., :3]) ** 2).sum() / inp.shape[0]
    return -mse_to_psnr(mse) + 30, mse_to_psnr(mse)


def loss_nerf_grad(inp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    mse = ((inp[..., :3] - target[..., :3]) ** 2).sum() / inp.shape[0]
    return (10 / np.log(10)) / mse * 2 * (inp[..., :3] - target[..., :3]) / inp.shape[0]


# ------------------------------------------------------------------------------- training step (row f3)
def _trainable(sd):
    return {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}


def prop_step_gradients(rays: Rays, sd, hp: Hyper, t_rand=None, u_rand=None):
    """train.py:55-62: Loss_prop(t, w, t_hat, w_hat).backward() -> (loss, {parameter name: gradient}) by autograd
    through the restated forward (only prop_net.* receive a gradient: t and w are detached, train.py:57-58).
    t_rand / u_rand: the draws of a randomized model (fixture G22), None = deterministic."""
    p = _trainable(sd)
    t_hat, w_hat = prop_forward(rays, p, hp, t_rand=t_rand)
    with torch.no_grad():
        _, _, _, t, w, _ = nerf_forward(rays, t_hat, w_hat, p, hp, u_rand=u_rand)
    loss = loss_prop_given(w_hat, prop_bounds(t, w, t_hat.detach()))
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in p.items() if k.startswith("prop_net")}


def nerf_step_gradients(rays: Rays, sd, hp: Hyper, pixels: torch.Tensor, dist_weight: float = 0.01, t_rand=None, u_rand=None):
    """train.py:69-80: (Loss_nerf + dist_weight * Loss_dist).backward() -> (loss_nerf, loss_dist, gradients of nerf_net.*)."""
    p = _trainable(sd)
    with torch.no_grad():
        t_hat, w_hat = prop_forward(rays, p, hp, t_rand=t_rand)
    rgb, _, _, _, fine_w, s_vals = nerf_forward(rays, t_hat, w_hat, p, hp, u_rand=u_rand)
    ln, _ = loss_nerf(rgb, pixels)
    ld = loss_dist(s_vals, fine_w)
    (ln + dist_weight * ld).backward()
    return ln.detach(), ld.detach(), {k: v.grad for k, v in p.items() if k.startswith("nerf_net")}


def nerf_output_gradients(rays: Rays, sd, hp: Hyper, c_dist=None, c_acc=None):
    """gradients of sum(distance * c_dist) + sum(acc * c_acc) w.r.t. nerf_net.* (the reference can differentiate acc;
    for distance its in-place g() makes autograd raise, so that half is the mathematical definition only)."""
    p = _trainable(sd)
    with torch.no_grad():
        t_hat, w_hat = prop_forward(rays, p, hp)
    _, dist, acc, _, _, _ = nerf_forward(rays, t_hat, w_hat, p, hp)
    loss = 0.0
    if c_dist is not None:
        loss = loss + (dist * c_dist).sum()
    if c_acc is not None:
        loss = loss + (acc * c_acc).sum()
    loss.backward()
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if k.startswith("nerf_net")}
