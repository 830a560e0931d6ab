"""Deterministic synthetic inputs for the ray-marching hot path.

Everything here is NumPy-only and seeded with PCG64 so that the authoring
container (where the golden fixtures are generated from the reference) and the
GPU box (where parity tests and bench.py run) see bit-identical ray batches and
weights without shipping a 30 MB checkpoint.

Shapes follow the reference's `Rays` tuple (intern/ray.py:6) and the
`state_dict` layout of `mipNeRF360` (model.py:43-53, 131-158).
The ray distributions follow SURVEY.md §8d:

* ``garden``: NDC-style rays as produced for nerf_360 scenes, near=0, far=1
  (config.py:64-82, dataset.py:364-387 in the reference).
* ``lego``: 400x400 pinhole camera on a radius-4 sphere, near=2, far=6,
  un-normalised directions (dataset.py:109-145,176 in the reference).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

RAY_FIELDS = ("origins", "directions", "viewdirs", "radii", "near", "far")


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(int(seed)))


def garden_rays(num_rays: int, seed: int = 1) -> dict:
    """NDC-style ray batch (near 0 / far 1), fp32, as a dict of numpy arrays."""
    g = _rng(seed)
    B = int(num_rays)
    ox = g.uniform(-1.0, 1.0, size=(B, 1))
    oy = g.uniform(-1.0, 1.0, size=(B, 1))
    origins = np.concatenate([ox, oy, -np.ones((B, 1))], axis=1)
    dx = g.uniform(-0.5, 0.5, size=(B, 1))
    dy = g.uniform(-0.5, 0.5, size=(B, 1))
    directions = np.concatenate([dx, dy, 2.0 * np.ones((B, 1))], axis=1)
    # camera-space pinhole view directions, z<0 hemisphere, unit length
    vx = g.uniform(-0.6, 0.6, size=(B, 1))
    vy = g.uniform(-0.4, 0.4, size=(B, 1))
    v = np.concatenate([vx, vy, -np.ones((B, 1))], axis=1)
    viewdirs = v / np.linalg.norm(v, axis=1, keepdims=True)
    radii = g.uniform(1e-3, 3e-3, size=(B, 1))
    near = np.zeros((B, 1))
    far = np.ones((B, 1))
    out = dict(origins=origins, directions=directions, viewdirs=viewdirs,
               radii=radii, near=near, far=far)
    return {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in out.items()}


def lego_rays(num_rays: int, seed: int = 1, hw: int = 400, focal: float = 555.6) -> dict:
    """Blender-style pinhole rays (near 2 / far 6), fp32."""
    g = _rng(seed)
    B = int(num_rays)
    # camera on a radius-4 sphere looking at the origin
    theta = g.uniform(0.0, 2.0 * math.pi)
    phi = g.uniform(0.15 * math.pi, 0.45 * math.pi)
    cam = 4.0 * np.array([math.sin(phi) * math.cos(theta),
                          math.sin(phi) * math.sin(theta),
                          math.cos(phi)])
    fwd = -cam / np.linalg.norm(cam)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    rot = np.stack([right, upv, -fwd], axis=1)  # camera-to-world
    px = g.integers(0, hw, size=B).astype(np.float64) + 0.5
    py = g.integers(0, hw, size=B).astype(np.float64) + 0.5
    cam_dirs = np.stack([(px - hw * 0.5) / focal, -(py - hw * 0.5) / focal,
                         -np.ones(B)], axis=1)
    directions = cam_dirs @ rot.T          # ||d|| in [1, ~1.12], not normalised
    viewdirs = directions / np.linalg.norm(directions, axis=1, keepdims=True)
    origins = np.broadcast_to(cam, (B, 3)).copy()
    radii = np.full((B, 1), 2.0 / (math.sqrt(12.0) * focal))
    near = np.full((B, 1), 2.0)
    far = np.full((B, 1), 6.0)
    out = dict(origins=origins, directions=directions, viewdirs=viewdirs,
               radii=radii, near=near, far=far)
    return {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in out.items()}


def make_rays(kind: str, num_rays: int, seed: int = 1) -> dict:
    if kind == "garden":
        return garden_rays(num_rays, seed)
    if kind == "lego":
        return lego_rays(num_rays, seed)
    if kind == "mixed":
        return mixed_rays(num_rays, seed)
    raise ValueError(f"unknown synthetic ray kind {kind!r}")


def state_dict_spec(hidden_proposal: int = 256, hidden_nerf: int = 1024,
                    viewdir_min_deg: int = 0, viewdir_max_deg: int = 4):
    """(name, shape) pairs of the reference checkpoint layout."""
    in_ch = 21 * 2 + (viewdir_max_deg - viewdir_min_deg) * 2 * 2
    spec = []
    hp, hn = int(hidden_proposal), int(hidden_nerf)
    dims = [(hp, in_ch), (hp, hp), (hp, hp), (hp, hp), (1, hp)]
    for i, (o, k) in zip((0, 2, 4, 6, 8), dims):
        spec.append((f"prop_net.model.{i}.weight", (o, k)))
        spec.append((f"prop_net.model.{i}.bias", (o,)))
    dims = [(hn, in_ch)] + [(hn, hn)] * 7
    for i, (o, k) in zip(range(0, 16, 2), dims):
        spec.append((f"nerf_net.model.{i}.weight", (o, k)))
        spec.append((f"nerf_net.model.{i}.bias", (o,)))
    spec.append(("nerf_net.final_density.0.weight", (1, hn)))
    spec.append(("nerf_net.final_density.0.bias", (1,)))
    spec.append(("nerf_net.final_color.0.weight", (3, hn)))
    spec.append(("nerf_net.final_color.0.bias", (3,)))
    return spec


def make_state_dict(hidden_proposal: int = 256, hidden_nerf: int = 1024, seed: int = 0,
                    viewdir_min_deg: int = 0, viewdir_max_deg: int = 4) -> "OrderedDict[str, np.ndarray]":
    """Kaiming-uniform-distributed weights / default-Linear-distributed biases.

    Same distributions as the reference's initialisation (model.py:8-12:
    kaiming_uniform_ with a=0 -> U(-sqrt(6/fan_in), +sqrt(6/fan_in)); bias
    U(-1/sqrt(fan_in), 1/sqrt(fan_in))) but drawn from a build-owned PCG64
    stream, so only the seed has to be committed.
    """
    g = _rng(seed)
    sd = OrderedDict()
    fan_in = None
    for name, shape in state_dict_spec(hidden_proposal, hidden_nerf,
                                       viewdir_min_deg, viewdir_max_deg):
        if name.endswith(".weight"):
            fan_in = shape[1]
            bound = math.sqrt(6.0 / fan_in)
        else:
            bound = 1.0 / math.sqrt(fan_in)
        sd[name] = g.uniform(-bound, bound, size=shape).astype(np.float32)
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# "trained-like" weights (fixture G19): high-contrast colours, peaked proposal weights, saturated and empty rays
# ---------------------------------------------------------------------------------------------------------------------
# the 21 unit directions of the reference's positional encoding (intern/encoding.py:9-30)
_A, _B, _C, _D = 0.8506508, 0.5257311, 0.809017, 0.309017
IPE_BASIS = np.array([
    [_A, 0, _B], [_C, 0.5, _D], [_B, _A, 0], [1, 0, 0], [_C, 0.5, -_D], [_A, 0, -_B],
    [_D, _C, -0.5], [0, _B, -_A], [0.5, _D, -_C], [0, 1, 0], [-_B, _A, 0], [-_D, _C, -0.5],
    [0, _B, _A], [-_D, _C, 0.5], [_D, _C, 0.5], [0.5, _D, _C], [0.5, -_D, _C], [0, 0, 1],
    [-0.5, _D, _C], [-_C, 0.5, _D], [-_C, 0.5, -_D]], dtype=np.float64)


def mixed_rays(num_rays: int, seed: int = 1) -> dict:
    """Pinhole rays like `lego_rays` but with PER-RAY bounds: the NeRF head of the reference bounds the density to
    softplus(sigmoid(.) - 1) in [0.313, 0.693] (model.py:150-158,185), so the accumulated opacity of a ray is a function
    of its length |d| (far - near) alone: 1 - exp(-0.5 |d| (far - near)) give or take.  A quarter of the rays is short
    (far - near in [0.02, 0.2]: acc < 0.15, "nearly empty"), a quarter long (far - near in [12, 40]: acc > 0.97,
    "saturated"), the rest ordinary (2 ... 8)."""
    out = lego_rays(num_rays, seed)
    g = _rng(seed + 7919)
    B = int(num_rays)
    cls = g.integers(0, 4, size=B)
    near = g.uniform(0.5, 3.0, size=B)
    length = np.where(cls == 0, g.uniform(0.02, 0.2, size=B),
                      np.where(cls == 1, g.uniform(12.0, 40.0, size=B), g.uniform(2.0, 8.0, size=B)))
    out["near"] = np.ascontiguousarray(near.reshape(B, 1).astype(np.float32))
    out["far"] = np.ascontiguousarray((near + length).reshape(B, 1).astype(np.float32))
    return out


def batch_geometry(rays: dict, num_samples: int) -> dict:
    """Where the proposal stage of ONE chunk puts its samples after the reference's whole-chunk contraction
    (intern/ray.py:99-101, intern/parameterization.py:23-29,99-101,135), in fp64: the contraction scale, the contracted
    sample positions' centroid, the unit axis the rays march along and the half extent of a ray along it.  A trained
    checkpoint is fitted to the geometry of its scene and chunk size; `make_structured_state_dict` uses this to put its
    density shells where the rays are."""
    o = rays["origins"].astype(np.float64)
    d = rays["directions"].astype(np.float64)
    near, far = rays["near"].astype(np.float64), rays["far"].astype(np.float64)
    s = np.linspace(0.0, 1.0, int(num_samples) + 1)[None, :]
    eps = 1e-6
    t = 1.0 / (s / (far + eps) + (1.0 - s) / (near + eps) + eps)
    mu, hw = (t[:, :-1] + t[:, 1:]) / 2, (t[:, 1:] - t[:, :-1]) / 2
    t_mean = mu + 2 * mu * hw ** 2 / (3 * mu ** 2 + hw ** 2)
    mean = d[:, None, :] * t_mean[..., None]
    n_g = math.sqrt(float((mean ** 2).sum()))
    scale = (2.0 - 1.0 / n_g) / n_g if n_g > 1.0 else 1.0
    x = o[:, None, :] + scale * mean
    axis = d.mean(0)
    axis /= np.linalg.norm(axis)
    along = (x @ axis)
    half = 0.5 * float(np.median(np.abs(along[:, -1] - along[:, 0])))
    return dict(scale=scale, centroid=x.reshape(-1, 3).mean(0), axis=axis, half_extent=max(half, 1e-9))


def mean_features(rays: dict, geo: dict, viewdir_min_deg: int = 0, viewdir_max_deg: int = 4) -> np.ndarray:
    """The encoder's 58 features (intern/encoding.py:33-56,69-90) at the chunk's centroid, view-direction part averaged
    over the rays, in fp64 (the exp(-sigma / 2) factor of the integrated encoding taken as 1)."""
    gamma0 = IPE_BASIS @ geo["centroid"]
    v = rays["viewdirs"].astype(np.float64)
    theta = np.arccos(np.clip(v[:, 2:3], -1.0, 1.0))
    phi = np.arctan(v[:, 1:2] / (v[:, 0:1] + 1e-6))
    sc = np.array([2.0 ** i for i in range(viewdir_min_deg, viewdir_max_deg)])[None, :]
    vd = np.concatenate([np.sin(theta * sc), np.cos(theta * sc), np.sin(phi * sc), np.cos(phi * sc)], 1)
    return np.concatenate([np.sin(gamma0), np.cos(gamma0), vd.mean(0)])


def make_structured_state_dict(hidden_proposal: int, hidden_nerf: int, seed: int, rays: dict, num_samples: int, *,
                               first_gain: float = 4.0, hidden_gain: float = 1.3, bias_gain: float = 3.0,
                               prop_head_gain: float = 3.0, shell_density: float = 150.0, sharpness: float = 40.0,
                               viewdir_shift: float = 0.6, colour_units: int = 12, colour_gain: float = 5.0,
                               colour_sharpness: float = 3.0,
                               viewdir_min_deg: int = 0, viewdir_max_deg: int = 4) -> "OrderedDict[str, np.ndarray]":
    """Build-owned "trained-like" weights in the reference's state_dict layout (model.py:43-53,131-158).

    `make_state_dict` draws every layer at its initialisation scale, where both nets are nearly constant functions of
    position and direction (a deep random ReLU net maps all inputs onto almost the same activation pattern): every ray
    renders the same grey and the proposal weights are flat.  This generator starts from the same draws and
      * scales the first layers (position / direction sensitivity), the hidden layers (wider pre-activations, partly
        saturated sigmoids), the proposal head (densities over decades) and the biases;
      * routes three designed units through the proposal net that turn a coordinate along the rays into density SHELLS:
        unit j of the first layer is relu(sharpness * (u - off_j)), where u in [-1, 1] is the position along the chunk's
        mean ray direction (a linear combination of the sin / cos features, linearised at the chunk's centroid, see
        `batch_geometry`) shifted per ray by the view-direction features; the units pass through the hidden layers
        unchanged, become on / off switches in the sigmoid layer and enter the density head with +, -, + shell_density:
        a thin shell, a gap, then a solid.  The proposal weights of a ray then have one or two sharp peaks
        (max / mean >> 20), or none when the ray misses the shells;
      * routes `colour_units` designed units through the NeRF net the same way: unit j is a random direction in feature
        space (positions and view directions), centred on the chunk's mean features, and enters the colour and density
        heads with weights of size colour_gain: colours and densities that vary over their whole range from ray to ray
        and along a ray.
    Only numpy and the seed are involved: the GPU box regenerates the same weights (fixtures carry a checksum)."""
    sd = make_state_dict(hidden_proposal, hidden_nerf, seed, viewdir_min_deg, viewdir_max_deg)
    n_ipe = 2 * IPE_BASIS.shape[0]
    in_ch = sd["prop_net.model.0.weight"].shape[1]
    f32 = np.float32
    for name in list(sd):
        w = sd[name].astype(np.float64)
        if name.endswith(".bias"):
            w = w * bias_gain
        elif name in ("prop_net.model.0.weight", "nerf_net.model.0.weight"):
            w = w * first_gain
        elif name == "prop_net.model.8.weight":
            w = w * prop_head_gain
        elif name.startswith("nerf_net.final_"):
            pass
        else:
            w = w * hidden_gain
        sd[name] = w.astype(f32)
    geo = batch_geometry(rays, num_samples)
    f0 = mean_features(rays, geo, viewdir_min_deg, viewdir_max_deg)
    # u(x) = axis . (x - centroid) / half_extent, written in the encoder's features: axis = sum_k a_k P_k (least squares),
    # P_k . (x - x0) ~ sin(gamma_k - gamma0_k) = sin(gamma_k) cos(gamma0_k) - cos(gamma_k) sin(gamma0_k)
    a = np.linalg.lstsq(IPE_BASIS.T, geo["axis"], rcond=None)[0]
    lin = np.zeros(in_ch)
    lin[:n_ipe // 2] = a * f0[n_ipe // 2:n_ipe] / geo["half_extent"]
    lin[n_ipe // 2:n_ipe] = -a * f0[:n_ipe // 2] / geo["half_extent"]
    g = _rng(seed + 104729)
    v = g.normal(size=in_ch - n_ipe)
    v *= viewdir_shift / np.linalg.norm(v)
    lin[n_ipe:] = v
    if int(hidden_proposal) >= 8:
        offsets = (-0.35, -0.1, 0.45)
        signs = (1.0, -1.0, 1.0)
        for j, off in enumerate(offsets):
            sd["prop_net.model.0.weight"][j, :] = (sharpness * lin).astype(f32)
            sd["prop_net.model.0.bias"][j] = f32(-sharpness * (off + float(lin[n_ipe:] @ f0[n_ipe:])))
            for layer in (2, 4, 6):
                sd[f"prop_net.model.{layer}.weight"][j, :] = 0
                sd[f"prop_net.model.{layer}.weight"][j, j] = 1
                sd[f"prop_net.model.{layer}.bias"][j] = 0
            sd["prop_net.model.6.bias"][j] = -4.0
            sd["prop_net.model.8.weight"][0, j] = f32(signs[j] * shell_density)
        off_state = sum(signs) * shell_density / (1.0 + math.exp(4.0))
        sd["prop_net.model.8.bias"][0] = f32(-5.0 - off_state)
    J = min(int(colour_units), int(hidden_nerf) // 4)
    margin = 6.0
    for j in range(J):
        r = g.normal(size=in_ch)
        r[:n_ipe] *= 0.0
        r[n_ipe:] *= colour_sharpness / math.sqrt(in_ch - n_ipe)
        r = r + (g.normal() * colour_sharpness) * lin * np.concatenate([np.ones(n_ipe), np.zeros(in_ch - n_ipe)])
        sd["nerf_net.model.0.weight"][j, :] = r.astype(f32)
        sd["nerf_net.model.0.bias"][j] = f32(margin - float(r[n_ipe:] @ f0[n_ipe:]))
        for layer in range(2, 16, 2):
            sd[f"nerf_net.model.{layer}.weight"][j, :] = 0
            sd[f"nerf_net.model.{layer}.weight"][j, j] = 1
            sd[f"nerf_net.model.{layer}.bias"][j] = 0
        sd["nerf_net.model.14.bias"][j] = f32(-margin)
        sd["nerf_net.final_color.0.weight"][:, j] = (colour_gain * g.normal(size=3)).astype(f32)
        sd["nerf_net.final_density.0.weight"][0, j] = f32(colour_gain * g.normal())
    return sd


def state_dict_checksum(sd) -> np.ndarray:
    """One fp64 number per tensor (sum of v * (1 + index mod 7)): fixtures built on regenerated full-width weights carry it
    so that a test on another machine can tell whether its regenerated weights are the ones the fixture was made with."""
    out = []
    for v in sd.values():
        x = np.asarray(v, dtype=np.float64).ravel()
        out.append(float((x * (1.0 + (np.arange(x.size) % 7))).sum()))
    return np.array(out, dtype=np.float64)
