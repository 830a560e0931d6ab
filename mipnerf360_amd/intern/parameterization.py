"""Mirror of the reference's intern/parameterization.py (same names, argument meaning).

Differences, both documented in DESIGN.md:
  * nothing is mutated in place (`g` in the reference does `x += 1e-6`); the same numeric
    offsets are applied inside the kernels;
  * `gaussian_contract` uses the closed-form Jacobian instead of a per-sample autograd loop.
Tensors must live on a HIP device; there is no CPU implementation.
"""
from __future__ import annotations

from .. import ops


def t_to_s(t_vals, near, far):
    """intern/parameterization.py:5-8: (g(t) - g(near)) / (g(far) - g(near)), with the
    reference's call order (the second g(near) sees near + 2e-6)."""
    return ops.t_to_s(t_vals, near, far, near_calls=0, far_calls=0)


def s_to_t(s_vals, near, far):
    """intern/parameterization.py:10-13."""
    return ops.s_to_t(s_vals, near, far)


def g(x):
    """intern/parameterization.py:15-21: 1/(x + 1e-6) (x itself is left untouched)."""
    return ops.g(x)


def contract(x):
    """intern/parameterization.py:23-29: norm taken over the WHOLE tensor."""
    return ops.contract(x)


def gaussian_to_xyz(d, t_mean, t_var, r_var, diag=False):
    """intern/parameterization.py:31-62 -> (mean[B,N,3], cov[B,N,3,3]), or with diag=True the diagonal cov[B,N,3]
    (a branch the reference's own hot path never takes)."""
    if diag:
        return ops.gaussian_to_xyz_diag(d, t_mean, t_var, r_var)
    return ops.gaussian_to_xyz(d, t_mean, t_var, r_var)


def gaussian_contract(mean, cov):
    """intern/parameterization.py:64-83."""
    return ops.gaussian_contract(mean, cov)


def conical_frustum_to_gaussian(d, t0, t1, base_radius, diag, stable=True):
    """intern/parameterization.py:85-117; stable=False takes the direct moment formulas (:108-113).  diag=True raises
    like the reference does there (its gaussian_contract needs the full 3 x 3 covariance, :76-81)."""
    if diag:
        raise RuntimeError("conical_frustum_to_gaussian(diag=True): gaussian_contract needs a full covariance (the reference "
                           "raises at this point as well)")
    t_mean, t_var, r_var = ops.frustum_moments(t0, t1, base_radius, stable=bool(stable))
    mean, cov = ops.gaussian_to_xyz(d, t_mean, t_var, r_var)
    return ops.gaussian_contract(mean, cov)


def para_rays(t_vals, origins, directions, radii, diag=False):
    """intern/parameterization.py:119-135 (origins are added after the contraction)."""
    if diag:
        raise RuntimeError("para_rays(diag=True): gaussian_contract needs a full covariance (the reference raises here as well)")
    return ops.para_rays(t_vals, origins, directions, radii)
