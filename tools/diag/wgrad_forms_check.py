#!/usr/bin/env python3
"""m360_linear_wgrad_bf16: the one-wave form (default) against the 8-wave form (0) and fp64, where they differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
for M, n, k in ((32832, 256, 1024), (4096, 1024, 1024), (524288, 1024, 1024), (320, 256, 1024)):
    g = torch.Generator().manual_seed(M)
    dz = torch.randn(M, n, generator=g).to(dev).bfloat16()
    x = torch.relu(torch.randn(M, k, generator=g)).to(dev).bfloat16()
    ref = dz.double().t() @ x.double() if M <= 65536 else None
    out = {}
    for form in (0, 1):
        ops.set_wgrad_bf16_form(form)
        gw, gb = ops.linear_wgrad_bf16(dz, x)
        gw2, gb2 = ops.linear_wgrad_bf16(dz, x)
        out[form] = (gw, gb)
        print(M, n, k, "form", form, "nan w", int(torch.isnan(gw).sum()), "nan b", int(torch.isnan(gb).sum()), "deterministic", bool(torch.equal(gw, gw2)),
              "err vs fp64", None if ref is None else float((gw.double() - ref).abs().max() / ref.abs().max()))
    d = (out[0][0] - out[1][0]).abs()
    bad = torch.nonzero(~torch.isfinite(out[1][0]) | (d > 1e-3 * out[0][0].abs().max()))
    print("   max |form0 - form1| / scale", float(d[torch.isfinite(d)].max() / out[0][0].abs().max()), "bad entries", bad.shape[0], bad[:6].tolist(),
          "bias diff", float((out[0][1] - out[1][1]).abs().max()))
    if bad.shape[0]:
        rows, cols = bad[:, 0], bad[:, 1]
        print("   bad rows: min", int(rows.min()), "max", int(rows.max()), "distinct", len(set(rows.tolist())), "| cols: min", int(cols.min()), "max", int(cols.max()), "distinct", len(set(cols.tolist())))
ops.set_wgrad_bf16_form(1)
