#!/usr/bin/env python3
"""Randomised bit-identity soak: the one-wave ring kernel (diag variant 200) against the ping-pong kernel (variant 3) on random
full-tile shapes - both accumulate the same 32-deep MFMA k-steps in the same order, so every output element must be equal."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mipnerf360_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=16)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--special", action="store_true", help="plant NaN / Inf / huge / denormal entries in the activations")
ap.add_argument("--x3", action="store_true", help="the bf16x3 forms (diag variants 301 / 300): [hi | lo] rows in and out, three products per block")
a = ap.parse_args()
dev = torch.device("cuda:0")
diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libm360_diag.so"))
vp = ctypes.c_void_p
diag.m360_diag_linear_bf16.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
rng = np.random.default_rng(a.seed)
bad = 0
for it in range(a.shapes):
    m = 256 * int(rng.integers(1, 700 if not a.x3 else 300))
    n = 256 * int(rng.integers(1, 5))
    k = (128 * int(rng.integers(2, 10))) if not a.x3 else (64 * int(rng.integers(2, 12)))
    pad_x, pad_y = 8 * int(rng.integers(0, 3)), 8 * int(rng.integers(0, 3))   # row strides that are not the row length
    g = torch.Generator(device=dev).manual_seed(it)
    xf = torch.rand(m, k, device=dev, generator=g) * 2 - 1
    if a.special:   # a few NaN / +-Inf / huge entries: the NaN-preserving ReLU and the bf16 conversion must agree bit for bit as well
        rows = torch.randint(0, m, (16,), device=dev, generator=g)
        cols = torch.randint(0, k, (16,), device=dev, generator=g)
        vals = torch.tensor([float("nan"), -float("nan"), float("inf"), -float("inf"), 3e38, -3e38, 1e-40, -0.0] * 2, device=dev)
        xf[rows, cols] = vals
    w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1) * (6.0 / k) ** 0.5
    b = torch.rand(n, device=dev, generator=g) - 0.5
    xm = 2 if a.x3 else 1   # row-length multiplier of the [hi | lo] layout
    x = torch.zeros(m, xm * k + pad_x, device=dev, dtype=torch.bfloat16)
    if a.x3:
        x[:, :2 * k] = ops.split_bf16x3(xf)
        wp, bp = ops.pack_linear_bf16x3(w, b, n, k)
        variants = (301, 300, 300)
    else:
        x[:, :k] = xf.bfloat16()
        wp, bp = ops.pack_linear_bf16(w, b, n, k)
        variants = (3, 200, 200)
    outs = []
    for variant in variants:
        y = torch.full((m, xm * n + pad_y), float("nan"), device=dev, dtype=torch.bfloat16)
        rc = diag.m360_diag_linear_bf16(x.data_ptr(), m, xm * k + pad_x, wp.data_ptr(), bp.data_ptr(), n, k, y.data_ptr(), xm * n + pad_y, variant, k,
                                        torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append(y)
    nn = xm * n
    d1 = int((outs[0][:, :nn].view(torch.int16) != outs[1][:, :nn].view(torch.int16)).sum())
    d2 = int((outs[1][:, :nn].view(torch.int16) != outs[2][:, :nn].view(torch.int16)).sum())
    untouched = bool(torch.isnan(outs[1][:, nn:].float()).all()) if pad_y else True
    print(f"{'bf16x3 ' if a.x3 else ''}M={m} N={n} K={k} ldx={xm * k + pad_x} ldy={nn + pad_y}: {d1} elements differ from the ping-pong kernel, {d2} between two "
          f"launches, padding untouched: {untouched}", flush=True)
    bad += (d1 != 0) + (d2 != 0) + (not untouched)
print("FAILED" if bad else "OK", flush=True)
sys.exit(1 if bad else 0)
