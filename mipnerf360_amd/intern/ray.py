"""Mirror of the reference's intern/ray.py (hot-path functions only)."""
from __future__ import annotations

from collections import namedtuple

import torch

from .. import ops

Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "near", "far"))  # intern/ray.py:6


def namedtuple_map(fn, tup):
    """intern/ray.py:8-10."""
    return type(tup)(*map(fn, tup))


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized=True):
    """intern/ray.py:12-57 (default randomized=True as in the reference; `weights` is not modified)."""
    # randomized: the kernel draws its uniforms itself from torch's generator state (ops.philox_state) - no torch.rand tensor
    return ops.sorted_pdf(bins, weights, num_samples, philox=ops.philox_state(bins.device) if randomized else None)


def convert_to_ndc(origins, directions, focal, w, h, near=1.0):
    """intern/ray.py:59-79.  NumPy in -> NumPy out like the reference; device tensors in -> device tensors out."""
    import numpy as np
    if isinstance(origins, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError("convert_to_ndc: no HIP device available; mipnerf360_amd has no CPU path")
        o, d = ops.convert_to_ndc(torch.from_numpy(np.ascontiguousarray(origins, dtype=np.float32)).cuda(),
                                  torch.from_numpy(np.ascontiguousarray(directions, dtype=np.float32)).cuda(),
                                  focal, w, h, near)
        return o.cpu().numpy(), d.cpu().numpy()
    return ops.convert_to_ndc(origins, directions, focal, w, h, near)


def generate_rays(cam_to_world, h, w, focal, near, far, ndc=False, span=None):
    """On-device equivalent of NeRFDataset.generate_rays + flatten_to_pytorch (dataset.py:109-145,147-150)
    and, with ndc=True, of LLFF.generate_rays (dataset.py:364-387) -> Rays of [n*h*w, .] device tensors.
    `span=(first, end)` (extension): only the rays of that flat pixel range, bit-identical to those rows of the
    full call - what one rank of a ray-sharded frame render generates for its own block of chunks."""
    return Rays(*ops.generate_rays(cam_to_world, h, w, focal, near, far, ndc, span=span))


def sample_along_rays(origins, directions, radii, num_samples, near, far, randomized):
    """intern/ray.py:81-116 -> (t_vals[B,N+1], (means[B,N,3], covs[B,N,3,3]))."""
    t_vals = ops.sample_t(near, far, num_samples, philox=ops.philox_state(near.device) if randomized else None)
    return t_vals, ops.para_rays(t_vals, origins, directions, radii)


def resample_along_rays(origins, directions, radii, t_vals, weights, randomized, resample_padding):
    """intern/ray.py:118-153."""
    new_t = ops.resample_t(t_vals, weights, resample_padding, philox=ops.philox_state(t_vals.device) if randomized else None)
    return new_t, ops.para_rays(new_t, origins, directions, radii)


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """intern/ray.py:155-191 -> (comp_rgb[B,3], distance[B], acc[B], weights[B,N])."""
    return ops.volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd)
