import os, sys, statistics, json
sys.path.insert(0, os.getcwd())
import torch
from mipnerf360_amd import ops
dev = torch.device("cuda:0")
for (n, k) in ((256, 256), (1024, 256), (256, 1024)):
    M = 524288
    dz = torch.randn(M, n, device=dev).bfloat16(); x = torch.relu(torch.randn(M, k, device=dev)).bfloat16()
    ref = None
    for form in (0, 1, 0, 1):
        ops.set_wgrad_bf16_form(form)
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gw, _ = ops.linear_wgrad_bf16(dz, x, want_bias=False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if ref is None: ref = gw.clone()
        err = float((gw - ref).abs().max() / ref.abs().max())
        ms = statistics.median(ts[2:])
        print(json.dumps({"n": n, "k": k, "form": form, "ms": round(ms, 4), "GBps": round((M * (n + k) * 2) / ms / 1e6, 1), "tflops": round(2.0 * M * n * k / ms / 1e9, 1), "rel_vs_form0": err}), flush=True)
