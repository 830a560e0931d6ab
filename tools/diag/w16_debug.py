#!/usr/bin/env python3
"""Where does the one-wave 16x16x32 ring kernel (diag variant 200) differ from the ping-pong kernel (variant 3) / from itself?
Both accumulate the same 32-deep MFMA k-steps in the same order, so the outputs should be bit-identical."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mipnerf360_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--k", type=int, default=256)
ap.add_argument("--variant", type=int, default=200)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(a.m, a.k, device=dev, generator=g) * 2 - 1).bfloat16()
w = (torch.rand(a.n, a.k, device=dev, generator=g) * 2 - 1) * (6.0 / a.k) ** 0.5
b = torch.rand(a.n, device=dev, generator=g) - 0.5
wp, bp = ops.pack_linear_bf16(w, b, a.n, a.k)
diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libm360_diag.so"))
vp = ctypes.c_void_p
diag.m360_diag_linear_bf16.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]


def run(variant, extra):
    y = torch.full((a.m, a.n), float("nan"), device=dev, dtype=torch.bfloat16)
    print(f"variant {variant}: x {x.data_ptr():#x} w {wp.data_ptr():#x} b {bp.data_ptr():#x} y {y.data_ptr():#x} .. {y.data_ptr() + y.numel() * 2:#x}", flush=True)
    rc = diag.m360_diag_linear_bf16(x.data_ptr(), a.m, a.k, wp.data_ptr(), bp.data_ptr(), a.n, a.k, y.data_ptr(), a.n, variant, extra,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return y


ref = run(3, a.k)
ref64 = (x[:2048].double() @ wp.double().T + bp.double()).clamp_min(0)
print("pp vs fp64 (first 2048 rows):", float((ref[:2048].double() - ref64).abs().max()))
for r in range(a.reps):
    y = run(a.variant, 0)
    bad = (y.view(torch.int16) != ref.view(torch.int16))
    nb = int(bad.sum())
    print(f"rep {r}: {nb} of {y.numel()} elements differ from the ping-pong kernel; NaN: {int(torch.isnan(y.float()).sum())}")
    if nb:
        idx = bad.nonzero()
        rows, cols = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
        print("  rows % 256 histogram (16-row blocks):", np.bincount((rows % 256) // 16, minlength=16).tolist())
        print("  cols % 256 histogram (32-col pieces):", np.bincount((cols % 256) // 32, minlength=8).tolist())
        print("  cols % 32 histogram (4-col groups):", np.bincount((cols % 32) // 4, minlength=8).tolist())
        print("  row tiles touched:", np.unique(rows // 256)[:20].tolist(), "... of", a.m // 256, "| col tiles:", np.unique(cols // 256).tolist())
        for i in range(min(6, nb)):
            rr, cc = int(rows[i]), int(cols[i])
            print(f"   y[{rr},{cc}] = {float(y[rr, cc]):.6f} vs {float(ref[rr, cc]):.6f}")
