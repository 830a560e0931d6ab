#!/usr/bin/env python3
"""bf16 weight-gradient kernel (m360_linear_wgrad_bf16, 1024 x 1024) at several row counts, warm (same operands again: a 32768-row pair is
128 MB - Infinity-Cache resident) - is the kernel bound by the latency of first-touch HBM reads (one stage of prefetch lead) or by its own loop?"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402

if os.environ.get("M360_WGRAD_FORM"):  # 0: the 8-wave kernel (the one the M360_TN16_ABL ablations of the diagnostics build act on), 1: the one-wave form
    ops.set_wgrad_bf16_form(int(os.environ["M360_WGRAD_FORM"]))

dev = torch.device("cuda:0")
for M in ((524288,) if (os.environ.get("M360_TN16_ABL") or os.environ.get("M360_TNW_ABL")) else (16384, 32768, 65536, 131072, 524288)):
    dz = torch.randn(M, 1024, device=dev).bfloat16()
    x = torch.relu(torch.randn(M, 1024, device=dev)).bfloat16()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_wgrad_bf16(dz, x, want_bias=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts[2:])
    print(json.dumps({"M": M, "ms": round(ms, 4), "tflops": round(2.0 * M * 1024 * 1024 / ms / 1e9, 1), "operand_MB": round(2 * M * 1024 * 2 / 2**20),
                      "stages_per_workgroup": M // 64 // 16, "us_per_stage": round(ms * 1e3 / max(M // 64 // 16, 1), 3)}), flush=True)
    del dz, x
