"""Mirror of the reference's intern/loss.py: Loss_prop, Loss_nerf, Loss_dist, mse_to_psnr, backed by the HIP loss
kernels (m360_loss_*) with analytic gradients, usable from the reference's train.py with tensors on the HIP device."""
from __future__ import annotations

import torch

from .. import ops
from .regularization import loss_dist


class _LossPropFull(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w, t_hat, w_hat):
        need = w_hat.requires_grad
        loss, _, grad = ops.loss_prop(t.detach(), w.detach(), t_hat.detach(), w_hat.detach(), want_grad=need)
        ctx.save_for_backward(*([grad] if need else []))
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return None, None, None, grad_out * g  # bounds are detached in the reference (distillation.py:31)


def Loss_prop(t, w, t_hat, w_hat):
    """intern/loss.py:6-21."""
    return _LossPropFull.apply(t, w, t_hat, w_hat)


class _LossNerf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target):
        out3, grad = ops.loss_nerf(inp.detach()[..., :3].contiguous(), target.detach()[..., :3].contiguous(),
                                   want_grad=inp.requires_grad)
        ctx.save_for_backward(*([grad] if inp.requires_grad else []))
        ctx.cols = inp.shape[-1]
        ctx.mark_non_differentiable(out3)
        return out3[0], out3

    @staticmethod
    def backward(ctx, grad_loss, _grad_out3):
        (g,) = ctx.saved_tensors
        if ctx.cols > 3:
            g = torch.cat([g, torch.zeros(g.shape[0], ctx.cols - 3, device=g.device)], -1)
        return grad_loss * g, None


def Loss_nerf(input, target):  # noqa: A002  (argument names of the reference)
    """intern/loss.py:23-40 -> (10 log10(mse) + 30, psnr)."""
    loss, out3 = _LossNerf.apply(input, target)
    return loss, out3[1]


def Loss_dist(s_vals, weights):
    """intern/loss.py:42-54."""
    return loss_dist(s_vals=s_vals, weights=weights)


def mse_to_psnr(mse):
    """intern/loss.py:57-59 (scalar helper, tensor op of the caller's device)."""
    return -10.0 * torch.log10(mse)
