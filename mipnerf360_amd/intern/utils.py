"""Mirror of the one hot-path function of the reference's intern/utils.py."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def to8b(img):
    """intern/utils.py:17-20: (255 * clip(nan_to_num(img), 0, 1)).astype(uint8).
    Accepts a NumPy array (returned as NumPy, like the reference) or a device tensor."""
    if isinstance(img, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError("to8b: no HIP device available; mipnerf360_amd has no CPU path")
        t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda()
        return ops.to8b(t).cpu().numpy()
    return ops.to8b(img)


# everything else of the reference's intern/utils.py (host-side helpers outside the hot path) falls through to the
# reference's own file when install_dropin(reference_root=...) was given one
from . import _fallback  # noqa: E402

__getattr__ = _fallback.make_getattr("utils", __name__)
