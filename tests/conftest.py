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
