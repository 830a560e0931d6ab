"""Mirror of the reference's intern/distillation.py (proposal / envelope loss), differentiable in w_hat."""
from __future__ import annotations

import torch

from .. import ops


def bounds(t_vals_fine, fine_weights, t_vals_coarse):
    """intern/distillation.py:4-33 (detached, like the reference): for every proposal interval the summed
    NeRF weights of the fine intervals that overlap it."""
    B, Np = t_vals_coarse.shape[0], t_vals_coarse.shape[1] - 1
    dummy = torch.ones(B, Np, device=t_vals_coarse.device)
    return ops.loss_prop(t_vals_fine.detach(), fine_weights.detach(), t_vals_coarse.detach(), dummy)[1]


class _LossProp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coarse_weights, bnd):
        need = coarse_weights.requires_grad
        loss, grad = ops.loss_prop_given_bounds(bnd.detach(), coarse_weights.detach(), want_grad=need)
        ctx.save_for_backward(*([grad] if need else []))
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return grad_out * g, None


def loss_prop(coarse_weights, bounds):
    """intern/distillation.py:35-51."""
    return _LossProp.apply(coarse_weights, bounds)
