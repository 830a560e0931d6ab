"""Mirror of the visualisation half of the reference's intern/pose.py (lines 112-212); the camera-path
helpers of that file (spiral / spherical poses, lines 1-110) are host-side NumPy run once per video and stay
out of scope.  NumPy in -> NumPy out like the reference; device tensors in -> device tensors out."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _to_dev(a):
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device available; mipnerf360_amd has no CPU path")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _wrap(fn, first, *rest, **kw):
    if isinstance(first, np.ndarray):
        conv = [None if r is None else (_to_dev(r) if isinstance(r, np.ndarray) else r) for r in rest]
        return fn(_to_dev(first), *conv, **kw).cpu().numpy()
    return fn(first, *rest, **kw)


def depth_to_normals(depth):
    """intern/pose.py:112-121."""
    return _wrap(ops.depth_to_normals, depth)


def sinebow(h):
    """intern/pose.py:122-125."""
    return _wrap(ops.sinebow, h)


def visualize_normals(depth, acc, scaling=None):
    """intern/pose.py:127-146.  Like the reference, only scaling=None produces a result (its body sits under
    `if scaling is None:` and otherwise falls through returning None)."""
    if scaling is not None:
        return None
    return _wrap(ops.visualize_normals, depth, acc)


def visualize_depth(depth, acc=None, near=None, far=None, ignore_frac=0, curve_fn=None, modulus=0, colormap=None):
    """intern/pose.py:148-212 with the reference's default curve (-log(x + eps)) and colormaps (matplotlib
    'turbo' for modulus == 0, sinebow otherwise).  Custom curve_fn / colormap callables and ignore_frac > 0
    (a global weighted quantile: a device-wide sort) are not implemented on the device path."""
    if curve_fn is not None or colormap is not None:
        raise NotImplementedError("visualize_depth: only the reference's default curve_fn / colormap are implemented")
    if ignore_frac:
        raise NotImplementedError("visualize_depth: ignore_frac > 0 is not implemented (callers in the reference never pass it)")
    near = None if near is None else float(np.asarray(near).reshape(-1)[0])
    far = None if far is None else float(np.asarray(far).reshape(-1)[0])
    return _wrap(ops.visualize_depth, depth, acc, near=near, far=far, modulus=float(modulus))


# everything else of the reference's intern/pose.py (host-side helpers outside the hot path) falls through to the
# reference's own file when install_dropin(reference_root=...) was given one
from . import _fallback  # noqa: E402

__getattr__ = _fallback.make_getattr("pose", __name__)
