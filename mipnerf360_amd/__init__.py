"""mipnerf360_amd — MI355X (gfx950) native ray-marching hot path of mip-NeRF 360.

Drop-in for the hot path of zhangkai0425/mipnerf360: `mipnerf360_amd.model` mirrors the
reference's `model.py`, `mipnerf360_amd.intern.*` its `intern/ray.py`,
`intern/parameterization.py`, `intern/encoding.py` and `intern/utils.py:to8b`.
All compute goes through libm360.so (hand-written HIP, C-ABI in include/m360.h).
"""
from __future__ import annotations

import sys

__version__ = "0.1.0"


def install_dropin() -> None:
    """Make `import model` / `from intern.ray import ...` (the reference's module names, as used
    by its train.py / test.py / video.py) resolve to this package."""
    from . import intern, model
    from .intern import distillation, encoding, loss, parameterization, pose, ray, regularization, utils
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
