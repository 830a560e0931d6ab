#!/usr/bin/env python3
"""m360_linear_wgrad_bf16 at 1024 x 1024 on 524 288 rows with and without the bias gradient: 10 calls each (run under rocprofv3 --kernel-trace --stats
to see which kernel the difference is in)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 524288
dz = torch.randn(M, 1024, device=dev).bfloat16()
x = torch.relu(torch.randn(M, 1024, device=dev)).bfloat16()
for want in (True, False, True, False):
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_wgrad_bf16(dz, x, want_bias=want)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("want_bias", want, "median ms", sorted(ts)[5])
