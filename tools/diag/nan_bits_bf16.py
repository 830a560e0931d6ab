#!/usr/bin/env python3
"""How does v_mfma_f32_16x16x32_bf16 hand a NaN operand on (sign, payload)?  Prints the fraction of NaN outputs with the sign bit set
for a NaN activation and for a NaN weight, through m360_linear_bf16 (none / ReLU) and m360_linear_bf16x3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for M, n, k in ((512, 256, 256), (100, 64, 64)):
    x = (torch.rand(M, k, device=dev, generator=g) * 2 - 1)
    w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1)
    b = torch.zeros(n, device=dev)
    for which in ("x", "w"):
        xx, ww = x.clone(), w.clone()
        if which == "x":
            xx[5, 3] = float("nan")
        else:
            ww[7, 3] = float("nan")
        wp, bp = ops.pack_linear_bf16(ww, b, n, k)
        for act, name in ((_lib.ACT_NONE, "none"), (_lib.ACT_RELU, "relu")):
            y = ops.linear_bf16(xx.bfloat16(), wp, bp, act).float()
            sel = y[5] if which == "x" else y[:, 7]
            nan = torch.isnan(sel)
            neg = (sel.view(torch.int32) < 0) & nan
            print(f"bf16   M={M} n={n} k={k} NaN in {which} act={name}: NaN outputs {int(nan.sum())}/{sel.numel()}, with sign bit {int(neg.sum())}")
        wp3, bp3 = ops.pack_linear_bf16x3(ww, b, n, k)
        y3 = ops.join_bf16x3(ops.linear_bf16x3(ops.split_bf16x3(xx), wp3, bp3, _lib.ACT_RELU))
        sel = y3[5] if which == "x" else y3[:, 7]
        print(f"bf16x3 M={M} n={n} k={k} NaN in {which} act=relu: NaN outputs {int(torch.isnan(sel).sum())}/{sel.numel()}")
