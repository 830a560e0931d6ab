#!/usr/bin/env python3
"""Does a captured HIP graph of one rendering forward (18 launches) run faster than the same launches issued from the host?
Prints eager vs graph-replay ms per forward for the three MLP precisions (4096 x 128, full width)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402

dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(256, 1024, seed=0)
r = synthetic.make_rays("garden", 4096, seed=1)
rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for mode, iters in (("bf16", 100), ("bf16x3", 40), ("fp32", 15)):
    m = mipNeRF360(num_samples=128, device=dev, mlp_dtype=mode)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    out = {"mode": mode}
    with torch.no_grad():
        ref = [t.clone() for t in m(rays)]
        out["eager_ms"] = round(timed(lambda: m(rays), iters), 4)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    m(rays)                      # workspace and packing for this stream, outside the capture
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                res = m(rays)
            torch.cuda.synchronize()
            out["graph_ms"] = round(timed(g.replay, iters), 4)
            out["graph_equals_eager"] = all(torch.equal(a, b) for a, b in zip(res, ref))
        except Exception as e:  # noqa: BLE001
            out["graph_error"] = f"{type(e).__name__}: {str(e)[:300]}"
    print(json.dumps(out), flush=True)
    del m
    torch.cuda.empty_cache()
