import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_cov_within_reference_error(got, ref32, ref64, c=3.0, what=""):
    """Tolerance of a covariance check, derived instead of asserted.  The reference's fp32 covariances are themselves
    inexact: t_var = hw^2/3 - (4/15) hw^4 (12 mu^2 - hw^2) / (3 mu^2 + hw^2)^2 (intern/parameterization.py:102-105)
    cancels almost completely for thin frusta (near = 0 / N = 128: the result is ~1e-5 of its terms).  `ref64` is the
    reference run in double precision on the same fp32 ray values (tests/golden/make_golden.py, reference_in_fp64), so
    e_ref = max |ref32 - ref64| / scale  (scale = largest entry of the sample's 3 x 3 matrix) is the reference's OWN
    rounding error; `got` must (a) be no further than c * e_ref from the fp64 values and (b) agree with the reference's
    fp32 values within (c + 1) * e_ref * scale, entry by entry.  Returns (e_got, e_ref)."""
    got, ref32, ref64 = (np.asarray(a, dtype=np.float64) for a in (got, ref32, ref64))
    assert got.shape == ref32.shape == ref64.shape, (what, got.shape, ref32.shape, ref64.shape)
    scale = np.abs(ref64).max(axis=(-1, -2), keepdims=True)
    scale = np.maximum(scale, np.finfo(np.float64).tiny)
    e_ref = float((np.abs(ref32 - ref64) / scale).max())
    e_got = float((np.abs(got - ref64) / scale).max())
    floor = 4 * np.finfo(np.float32).eps  # a result that happens to be exact in fp32 must not make the bound vanish
    bound = c * max(e_ref, floor)
    assert e_got <= bound, f"{what}: {e_got:.3e} from the fp64 values, the reference itself {e_ref:.3e} (allowed {c} x)"
    assert np.all(np.abs(got - ref32) <= (bound + e_ref) * scale), f"{what}: differs from the reference's fp32 values by more than its own error budget"
    return e_got, e_ref


G19_STAGE_NAMES = ("w_hat", "rgb", "dist", "acc", "t_vals", "fine_w", "s_vals")


def g19_case(g, tag):
    """(B, n, white_bkgd, hp, hn, seed), rays dict and state dict of one case of fixture G19: the reduced-width cases store
    their weights, the full-width ones regenerate them from the seed and check the stored per-tensor checksums (to 1e-9
    relative: another machine's libm / BLAS may move a weight by an ulp, which the tolerances below do not notice)."""
    from mipnerf360_amd import synthetic
    B, n, wb, hp, hn, seed = (int(x) for x in g[tag + "_cfg"])
    rays = {k: g[f"{tag}_rays_{k}"] for k in synthetic.RAY_FIELDS}
    pre = tag + "_sd."
    sd = {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}
    if not sd:
        sd = synthetic.make_structured_state_dict(hp, hn, seed, rays, n)
        np.testing.assert_allclose(synthetic.state_dict_checksum(sd), g[tag + "_sdsum"], rtol=1e-9, atol=1e-9)
    return (B, n, bool(wb), hp, hn, seed), rays, sd


def assert_within_reference_error(got, ref32, ref64, c=4.0, floor=2e-6, relative_above_one=False, what=""):
    """Tolerance of a whole-path check on ill-conditioned ("trained-like", fixture G19) weights, derived instead of
    asserted: `ref64` is the reference run in double precision on the same fp32 inputs and weights, so
    e_ref = max |ref32 - ref64| is the reference's OWN fp32 error; `got` must be no further than c * max(e_ref, floor)
    from the fp64 values.  relative_above_one: errors are divided by max(1, |ref64|) (distances).  Returns (e_got, e_ref)."""
    got, ref32, ref64 = (np.asarray(a, dtype=np.float64) for a in (got, ref32, ref64))
    assert got.shape == ref32.shape == ref64.shape, (what, got.shape, ref32.shape, ref64.shape)
    scale = np.maximum(1.0, np.abs(ref64)) if relative_above_one else 1.0
    e_ref = float((np.abs(ref32 - ref64) / scale).max())
    e_got = float((np.abs(got - ref64) / scale).max())
    assert np.isfinite(e_got) and e_got <= c * max(e_ref, floor), \
        f"{what}: {e_got:.3e} from the reference's fp64 values; the reference's own fp32 run is {e_ref:.3e} away (allowed {c} x, floor {floor:.0e})"
    return e_got, e_ref


def oracle_stages(rays_np, sd_np, num_samples, white_bkgd, dtype="float32", mlp_bf16=0):
    """Both stage forwards of the oracle in fp32 or in fp64 (the same code on double inputs with torch's default dtype switched,
    exactly how make_golden.py runs the reference in fp64) -> {name: numpy array} over G19_STAGE_NAMES + t_hat."""
    import torch

    from oracle import ref_path as O
    dt = getattr(torch, dtype)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        rays = O.Rays(*[torch.from_numpy(np.ascontiguousarray(rays_np[k])).to(dt) for k in O.Rays._fields])
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dt) for k, v in sd_np.items()}
        hp = O.Hyper(num_samples=num_samples, white_bkgd=white_bkgd, mlp_bf16=mlp_bf16)
        with torch.no_grad():
            t_hat, w_hat = O.prop_forward(rays, sd, hp)
            out = O.nerf_forward(rays, t_hat, w_hat, sd, hp)
    finally:
        torch.set_default_dtype(old)
    res = {nm: v.numpy() for nm, v in zip(G19_STAGE_NAMES, (w_hat,) + tuple(out))}
    res["t_hat"] = t_hat.numpy()
    return res


def g21_case(g, kind):
    from mipnerf360_amd import synthetic
    B, n, wb, hp, hn = (int(x) for x in g[kind + "_cfg"])
    pre = kind + "_sd."
    sd = {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}
    rays = {k: g[f"{kind}_rays_{k}"] for k in synthetic.RAY_FIELDS}
    return (B, n, bool(wb), hp, hn), rays, sd, g[kind + "_pixels"]


def assert_grad_within_reference_error(got, ref32, ref64, c=4.0, floor=2e-4, what=""):
    """A parameter gradient on ill-conditioned (G19 / G21) weights: no further from the reference's fp64 gradient than c x the
    reference's own fp32 gradient is, all relative to the tensor's largest entry; floor = the flat regime's 2e-4."""
    got, ref32, ref64 = (np.asarray(a, dtype=np.float64) for a in (got, ref32, ref64))
    scale = max(float(np.abs(ref64).max()), 1e-30)
    e_ref = float(np.abs(ref32 - ref64).max()) / scale
    e_got = float(np.abs(got - ref64).max()) / scale
    assert np.isfinite(e_got) and e_got <= c * max(e_ref, floor), f"{what}: {e_got:.3e} of the tensor's scale from the fp64 gradient; the reference's fp32 gradient is {e_ref:.3e} away"
    return e_got, e_ref
