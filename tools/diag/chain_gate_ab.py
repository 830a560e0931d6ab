#!/usr/bin/env python3
"""What the safety of the bf16 mode's layer chain costs (round 5): ms per forward at 4096 x 128 (and 8192 x 256 with --c5), alternating,
 (a) chain + six gated re-run launches behind it (the product), (b) chain alone, unchecked (M360_TUNE_CHAIN_UNGATED, diagnostics build: run with M360_LIB=mipnerf360_amd/libm360_diag.so; A/B only),
 (c) chain through hipLaunchCooperativeKernel + gated launches, (d) no chain: six launches.  No event recorder attached."""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops, synthetic  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--c5", action="store_true")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N = (8192, 256) if a.c5 else (4096, 128)
m = mipNeRF360(num_samples=N, hidden_proposal=256, hidden_nerf=1024, mlp_dtype="bf16", device=dev, randomized=False).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(256, 1024, seed=0).items()})
r = synthetic.make_rays("garden", B, seed=1)
rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])


def run(cfg):
    fault, coop, chain = cfg
    ops.set_chain_debug(0, fault)
    ops.set_chain_cooperative(bool(coop))
    was = ops.set_hidden_chain(bool(chain))
    try:
        with torch.no_grad():
            for _ in range(3):
                out = m(rays)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                out = m(rays)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.iters * 1e3
    finally:
        ops.set_hidden_chain(was)
        ops.set_chain_debug(0, 0)
        ops.set_chain_cooperative(False)
    return ms, [o.clone() for o in out]


configs = {"chain+gated": (0, 0, 1), "chain_unchecked": (-1, 0, 1), "cooperative+gated": (0, 1, 1), "six_launches": (0, 0, 0)}
ref = None
res = {k: [] for k in configs}
for _ in range(a.rounds):
    for name, cfg in configs.items():
        ms, out = run(cfg)
        if ref is None:
            ref = out
        same = all(torch.equal(x, y) for x, y in zip(out, ref))
        res[name].append(round(ms, 4))
        assert same, name
print(json.dumps({"B": B, "N": N, "iters": a.iters, "ms_per_forward": res, "median": {k: statistics.median(v) for k, v in res.items()},
                  "chain_status": m.chain_status(), "same_bits": True}), flush=True)
