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
