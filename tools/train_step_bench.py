#!/usr/bin/env python3
"""Time the reference's train.py loop body (train.py:53-84) on the HIP mirrors at the BASELINE shape
(4096 rays x 128 samples, hidden 256 / 1024, fp32), plus the two gradient GEMMs on their own.

    python tools/train_step_bench.py [--rays 4096] [--samples 128] [--iters 3]

Prints one JSON line.  Not the headline metric (bench.py is); evidence for DESIGN.md's training section.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import ops, synthetic  # noqa: E402
from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--gemm-only", action="store_true")
    ap.add_argument("--mlp-dtype", default="fp32", choices=("fp32", "bf16"), help="precision the model trains in (round 5: bf16)")
    ap.add_argument("--no-gemms", action="store_true", help="skip the stand-alone gradient GEMMs (kernel profiles of the iteration alone)")
    ap.add_argument("--overlap", type=int, default=None, help="bf16: 1 = the ReLU mask on a second stream (m360_side_t) beside the weight gradient (default), 0 = one stream")
    a = ap.parse_args()
    if a.overlap is not None:
        from mipnerf360_amd import ops as _ops
        _ops.set_backward_overlap(bool(a.overlap))
    dev = torch.device("cuda:0")
    out = {"rays": a.rays, "samples": a.samples, "mlp_dtype": a.mlp_dtype}
    if a.overlap is not None:
        out["backward_overlap"] = a.overlap
    b16 = a.mlp_dtype == "bf16"

    # --- the two gradient GEMMs of one 1024 x 1024 layer
    M = a.rays * a.samples
    if not a.no_gemms:
        dt = torch.bfloat16 if b16 else torch.float32
        dz = torch.randn(M, 1024, device=dev).to(dt)
        x = torch.relu(torch.randn(M, 1024, device=dev)).to(dt)
        w = torch.randn(1024, 1024, device=dev) / 32
        wt = (ops.pack_linear_bf16_transposed if b16 else ops.pack_linear_transposed)(w)
        dx = torch.empty(M, 1024, device=dev, dtype=dt)
        wgrad, dgrad = (ops.linear_wgrad_bf16, ops.linear_dgrad_bf16) if b16 else (ops.linear_wgrad, ops.linear_dgrad)
        flops = 2.0 * M * 1024 * 1024
        ms = timed(lambda: wgrad(dz, x), a.iters)
        out["wgrad_ms"], out["wgrad_tflops"] = round(ms, 3), round(flops / ms / 1e9, 1)
        ms = timed(lambda: wgrad(dz, x, want_bias=False), a.iters)
        out["wgrad_nobias_ms"] = round(ms, 3)
        ms = timed(lambda: dgrad(dz, wt, x, out=dx), a.iters)
        out["dgrad_ms"], out["dgrad_tflops"] = round(ms, 3), round(flops / ms / 1e9, 1)
        ms = timed(lambda: dgrad(dz, wt, None, out=dx), a.iters)
        out["dgrad_nomask_ms"] = round(ms, 3)
        del dz, x, dx
    if a.gemm_only:
        print(json.dumps(out))
        return

    model = mipNeRF360(randomized=False, num_samples=a.samples, device=dev, mlp_dtype=a.mlp_dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(256, 1024, seed=0).items()})
    model.train()
    r = synthetic.make_rays("garden", a.rays, seed=1)
    rays = Rays(*[torch.from_numpy(r[f]).to(dev) for f in synthetic.RAY_FIELDS])
    pixels = torch.rand(a.rays, 3, device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)

    def prop_step():
        t_hat, w_hat = model.prop_net.forward(rays)
        with torch.no_grad():
            _, _, _, t, w_, _ = model.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        loss = Loss_prop(t=t, w=w_, t_hat=t_hat, w_hat=w_hat)
        opt.zero_grad()
        loss.backward()
        opt.step()

    def nerf_step():
        with torch.no_grad():
            t_hat, w_hat = model.prop_net.forward(rays)
        rgb, _, _, _, fw, sv = model.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        ln, _ = Loss_nerf(input=rgb, target=pixels)
        ld = Loss_dist(s_vals=sv, weights=fw)
        opt.zero_grad()
        (ln + 0.01 * ld).backward()
        opt.step()

    def iteration():            # train.py:53-84: two proposal updates, one NeRF update
        prop_step()
        prop_step()
        nerf_step()

    out["prop_step_ms"] = round(timed(prop_step, a.iters), 2)
    out["nerf_step_ms"] = round(timed(nerf_step, a.iters), 2)
    it = timed(iteration, a.iters)
    out["iteration_ms"] = round(it, 2)
    out["train_rays_per_s"] = round(a.rays / it * 1e3, 1)
    # FLOPs of the NeRF update: forward (both nets) + backward of the NeRF net (2x its forward, minus layer-0 dgrad)
    n = a.samples
    out["nerf_step_tflops"] = round((a.rays * n * (423424 + 3 * 14807040)) / out["nerf_step_ms"] / 1e9, 1)
    out["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
