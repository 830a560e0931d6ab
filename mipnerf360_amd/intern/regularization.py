"""Mirror of the reference's intern/regularization.py (distortion loss), differentiable."""
from __future__ import annotations

import torch

from .. import ops


class _LossDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_vals, weights):
        need = s_vals.requires_grad or weights.requires_grad
        loss, gw, gs = ops.loss_dist(s_vals.detach(), weights.detach(), want_grad=need)
        ctx.save_for_backward(*([gw, gs] if need else []))
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        gw, gs = ctx.saved_tensors
        return grad_out * gs, grad_out * gw


def loss_dist(s_vals, weights):
    """intern/regularization.py:3-19: sum over rays of sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 ds_i
    (one kernel instead of the N^2 Python loop; gradients w.r.t. weights and s_vals are analytic)."""
    return _LossDist.apply(s_vals, weights)
