"""Same-named mirrors of the reference's `intern.ray`, `intern.parameterization`,
`intern.encoding` and `intern.utils` modules, backed by libm360 HIP kernels."""
from . import distillation, encoding, loss, parameterization, pose, ray, regularization, utils  # noqa: F401
