#!/usr/bin/env python3
"""BASELINE.json configs[2]: one full 1237 x 822 nerf_360/garden-sized frame (1 016 814 rays), rays generated
on the device from a synthetic forward-facing pose (NDC, near 0 / far 1), chunks of 4096, full-width fp32
MLPs, random-init weights.  Reports seconds per frame and rays/s for "128+128" (the reference's equal
counts) and "64+128" (hierarchical extension).  Not the headline bench (bench.py is), a measurement tool."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1237)
    ap.add_argument("--height", type=int, default=822)
    ap.add_argument("--chunks", type=int, default=4096)
    ap.add_argument("--configs", type=str, default="128+128,64+128")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(256, 1024, seed=0).items()}
    pose = np.concatenate([np.eye(3), np.array([[0.05], [-0.02], [0.1]])], 1).astype(np.float32)
    focal = 0.9 * args.width
    out = []
    for cfg in args.configs.split(","):
        n_prop, n_fine = (int(x) for x in cfg.split("+"))
        m = mipNeRF360(num_samples=n_prop, hidden_proposal=256, hidden_nerf=1024, device=dev,
                       num_samples_fine=None if n_fine == n_prop else n_fine)
        m.load_state_dict(sd)
        m.render_view(pose, 64, 64, focal, 0.0, 1.0, ndc=True, chunks=args.chunks)  # warm-up (packing, workspace)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rgb8, dist, acc = m.render_view(pose, args.height, args.width, focal, 0.0, 1.0, ndc=True, chunks=args.chunks)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = args.width * args.height
        assert rgb8.shape == (args.height, args.width, 3) and np.isfinite(dist).all() and np.isfinite(acc).all()
        out.append({"config": cfg, "rays": rays, "seconds_per_frame": round(dt, 3), "rays_per_s": round(rays / dt, 1),
                    "chunks": args.chunks, "n_chunks": (rays + args.chunks - 1) // args.chunks})
        del m
    print(json.dumps({"workload": f"{args.width}x{args.height} frame, pose -> uint8 frame on the host", "results": out}))


if __name__ == "__main__":
    main()
