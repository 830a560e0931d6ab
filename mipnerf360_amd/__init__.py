"""mipnerf360_amd — MI355X (gfx950) native ray-marching hot path of mip-NeRF 360.

Drop-in for the hot path of zhangkai0425/mipnerf360: `mipnerf360_amd.model` mirrors the
reference's `model.py`, `mipnerf360_amd.intern.*` its `intern/ray.py`,
`intern/parameterization.py`, `intern/encoding.py` and `intern/utils.py:to8b`.
All compute goes through libm360.so (hand-written HIP, C-ABI in include/m360.h).
"""
from __future__ import annotations

import sys

__version__ = "0.1.0"


def install_dropin(reference_root=None, mutate_like_reference: bool = True) -> None:
    """Make `import model` / `from intern.ray import ...` (the reference's module names, as used
    by its train.py / test.py / video.py) resolve to this package.

    `reference_root`: the reference checkout.  Its scripts also import host-side helpers that are outside the hot path
    (`intern.scheduler.lr_decay`, the camera paths of `intern.pose`, `intern.utils.normalize / to_float`); given the root,
    those keep resolving to the reference's own files: `intern.__path__` is extended by `<root>/intern` (modules this
    package has no mirror for) and names on the allowlist of out-of-scope host helpers fall through lazily
    (mipnerf360_amd/intern/_fallback.py).

    `mutate_like_reference` (default True HERE, because the drop-in exists to run the reference's own scripts): models
    built afterwards also reproduce the reference's in-place `g()` side effect on `rays.near` / `rays.far` (+3e-6 / +2e-6
    per forward pair, intern/parameterization.py:15-21), so train.py's three pairs per iteration drift exactly like the
    reference's (fixture G14) and training curves are comparable.  False: caller tensors are never mutated and every
    forward behaves like the reference's FIRST forward on fresh rays - the default of `mipnerf360_amd.model` used directly."""
    import os
    from . import intern, model
    from .intern import _fallback, distillation, encoding, loss, parameterization, pose, ray, regularization, utils
    if reference_root is not None:
        ref_intern = os.path.join(os.path.abspath(reference_root), "intern")
        if not os.path.isdir(ref_intern):
            raise FileNotFoundError(f"{ref_intern} is not a directory")
        _fallback.set_reference_root(reference_root)
        if ref_intern not in intern.__path__:
            intern.__path__.append(ref_intern)
    model.MUTATE_LIKE_REFERENCE = bool(mutate_like_reference)
    sys.modules["model"] = model
    sys.modules["intern"] = intern
    sys.modules["intern.ray"] = ray
    sys.modules["intern.parameterization"] = parameterization
    sys.modules["intern.encoding"] = encoding
    sys.modules["intern.utils"] = utils
    sys.modules["intern.loss"] = loss
    sys.modules["intern.distillation"] = distillation
    sys.modules["intern.regularization"] = regularization
    sys.modules["intern.pose"] = pose  # visualisation half only (visualize_depth / visualize_normals)
