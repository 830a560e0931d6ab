"""Mirror of the reference's intern/encoding.py."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops

_A, _B, _C, _D = 0.8506508, 0.5257311, 0.809017, 0.309017


class PositionalEncoding(nn.Module):
    """intern/encoding.py:5-61.  `P` is a plain attribute (not in the state_dict), kept for
    API compatibility; the kernel holds the same 21 directions in constant memory."""

    def __init__(self):
        super().__init__()
        self.P = torch.tensor([
            [_A, 0, _B], [_C, 0.5, _D], [_B, _A, 0], [1, 0, 0], [_C, 0.5, -_D], [_A, 0, -_B],
            [_D, _C, -0.5], [0, _B, -_A], [0.5, _D, -_C], [0, 1, 0], [-_B, _A, 0], [-_D, _C, -0.5],
            [0, _B, _A], [-_D, _C, 0.5], [_D, _C, 0.5], [0.5, _D, _C], [0.5, -_D, _C], [0, 0, 1],
            [-0.5, _D, _C], [-_C, 0.5, _D], [-_C, 0.5, -_D]], requires_grad=False)

    def forward(self, mean, cov):
        return ops.ipe(mean, cov)


class ViewdirectionEncoding(nn.Module):
    """intern/encoding.py:63-90."""

    def __init__(self, viewdir_min_deg, viewdir_max_deg):
        super().__init__()
        self.min_deg, self.max_deg = int(viewdir_min_deg), int(viewdir_max_deg)
        self.scales = torch.tensor([2 ** i for i in range(self.min_deg, self.max_deg)], dtype=torch.float32,
                                   requires_grad=False)

    def forward(self, viewdirs):
        return ops.viewdir_enc(viewdirs, self.min_deg, self.max_deg)
