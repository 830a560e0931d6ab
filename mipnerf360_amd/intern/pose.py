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
    """intern/pose.py:148-212.  Defaults (curve -log(x + eps), matplotlib 'turbo' / sinebow) run entirely on the device,
    `ignore_frac > 0` included (device sort + numpy's sequential float32 cumsum).  `curve_fn` / `colormap` are Python
    callables in the reference's API, i.e. the caller's own host code: they are applied where the reference applies them
    (to the depth map and the two planes / to the normalised value map) on host copies, everything around them - plane
    selection, normalisation, modulus, blending with acc - stays on the device."""
    np_in = isinstance(depth, np.ndarray)
    d = _to_dev(depth) if np_in else depth
    a = None if acc is None else (_to_dev(acc) if isinstance(acc, np.ndarray) else acc)
    near = None if near is None else float(np.asarray(near).reshape(-1)[0])
    far = None if far is None else float(np.asarray(far).reshape(-1)[0])

    def back(t):
        return t.cpu().numpy() if np_in else t

    if curve_fn is None and colormap is None:
        if not ignore_frac:
            return back(ops.visualize_depth(d, a, near=near, far=far, modulus=float(modulus)))
        return back(ops.visualize_depth_ex(d, a, near, far, ignore_frac=float(ignore_frac), modulus=float(modulus)))
    curved = curve_fn is not None
    auto = {}
    if curved:
        if not near or not far:  # the automatic planes are chosen on the UNcurved map, then curved like it
            planes = ops.visualize_depth_ex(d, a, near, far, ignore_frac=float(ignore_frac), want="planes").cpu().numpy()
            near, far = near or float(planes[0]), far or float(planes[1])
        host = d.cpu().numpy()
        dc = _to_dev(np.asarray(curve_fn(host), dtype=np.float32))
        near, far = float(curve_fn(np.float32(near))), float(curve_fn(np.float32(far)))
        auto = dict(near_auto=False, far_auto=False)  # both planes are known now; a curved value of exactly 0.0 is legal
    else:
        dc = d
    if colormap is None:
        return back(ops.visualize_depth_ex(dc, a, near, far, ignore_frac=float(ignore_frac), curved=curved, modulus=float(modulus),
                                           **auto))
    value = ops.visualize_depth_ex(dc, a, near, far, ignore_frac=float(ignore_frac), curved=curved, modulus=float(modulus),
                                   want="value", **auto).cpu().numpy()
    colors = np.asarray(colormap(value), dtype=np.float32)[:, :, :3]
    return back(ops.visualize_composite(_to_dev(colors), a, d))


# everything else of the reference's intern/pose.py (host-side helpers outside the hot path) falls through to the
# reference's own file when install_dropin(reference_root=...) was given one
from . import _fallback  # noqa: E402

__getattr__ = _fallback.make_getattr("pose", __name__)
