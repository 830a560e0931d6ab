#!/usr/bin/env python3
"""Row blocks of the NeRF MLP in the bf16 mode (m360_set_row_blocks / m360_set_row_block_streams) against layer by layer, WITHOUT the
event recorder bench.py attaches (two events per launch weigh on 100-200 launches per forward): ms per rendering forward of
4096 x 128 (and 8192 x 256 with --c5), full width, alternating configurations on one box; outputs compared bit for bit."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import ops, synthetic  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--c5", action="store_true")
ap.add_argument("--mode", default="bf16")
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--configs", default="0:1,49152:1,24576:2,32768:2,49152:2,16384:2,0:1,24576:2")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N = (8192, 256) if a.c5 else (4096, 128)
sd = synthetic.make_state_dict(256, 1024, seed=0)
r = synthetic.make_rays("garden", B, seed=1)
rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])
m = mipNeRF360(num_samples=N, device=dev, mlp_dtype=a.mode)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m.eval()


def timed(iters):
    with torch.no_grad():
        for _ in range(5):
            m(rays)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            m(rays)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


ref = None
for cfg in a.configs.split(","):
    rows, streams = (int(v) for v in cfg.split(":"))
    ops.set_row_blocks(rows)
    from mipnerf360_amd import _lib
    _lib.lib().m360_set_row_block_streams(streams)
    with torch.no_grad():
        out = [t.clone() for t in m(rays)]
    if ref is None:
        ref = out
    same = all(torch.equal(x, y) for x, y in zip(out, ref))
    print(json.dumps({"mode": a.mode, "rays": B, "samples": N, "row_blocks": rows, "streams": streams, "ms_per_forward": round(timed(a.iters), 4),
                      "same_bits_as_first": same}), flush=True)
