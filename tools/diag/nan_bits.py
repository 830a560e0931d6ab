#!/usr/bin/env python3
"""Which NaN bit patterns come out of the fp32 MFMA path (diagnostic for the NaN-preserving ReLU, fixture G18)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mipnerf360_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for M, n, k in ((1024, 256, 64), (90, 96, 64), (1024, 512, 32)):
    x = torch.rand(M, k, device=dev, generator=g) * 2 - 1
    w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1) * (6.0 / k) ** 0.5
    b = torch.rand(n, device=dev, generator=g) - 0.5
    wp, bp = ops.pack_linear(w, b)
    for name, bits in (("+qNaN 7FC00000", 0x7FC00000), ("-qNaN FFC00000", -4194304), ("+NaN payload 7FC12345", 0x7FC12345), ("-sNaN FF800001", -8388607)):
        bad = x.clone()
        bad.view(torch.int32)[5, 3] = bits
        for act, an in ((_lib.ACT_NONE, "none"), (_lib.ACT_RELU, "relu"), (_lib.ACT_SIGMOID, "sigmoid")):
            y = ops.linear(bad, wp, bp, act)
            row = y[5].view(torch.int32).cpu()
            pats = {}
            for v in row.tolist():
                pats[v & 0xFFFFFFFF] = pats.get(v & 0xFFFFFFFF, 0) + 1
            top = sorted(pats.items(), key=lambda kv: -kv[1])[:4]
            print(f"M={M} n={n} k={k} in={name} act={an}: nan units {int(torch.isnan(y[5]).sum())}/{y.shape[1]}; patterns " +
                  ", ".join(f"{p:08X}x{c}" for p, c in top))
