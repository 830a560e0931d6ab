#!/usr/bin/env python3
"""What the reference's test.py does after its data loader (test.py:28-58), without cv2 / dataset files:
load a reference checkpoint (`model_*.pt`, a plain state_dict), render views, colour-map depth and normals,
write PNGs.  Poses come from a small forward-facing sweep (or `--poses file.npy`, [V, 3, 4] camera-to-world).

    python tools/render_checkpoint.py model.pt out_dir --width 200 --height 150 --views 3 [--chunks 4096] [--ndc]

Everything between the 3 x 4 pose and the uint8 images runs on the device: ray generation, the two-stage forward,
to8b, visualize_depth / visualize_normals.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import checkpoint, png  # noqa: E402
from mipnerf360_amd.intern.pose import visualize_depth, visualize_normals  # noqa: E402
from mipnerf360_amd.intern.utils import to8b  # noqa: E402


def sweep_poses(views: int) -> np.ndarray:
    out = []
    for v in range(views):
        a = 0.15 * (v - (views - 1) / 2)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        out.append(np.concatenate([rot, np.array([[0.3 * a], [0.0], [0.0]], np.float32)], 1))
    return np.stack(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("out_dir")
    ap.add_argument("--width", type=int, default=200)
    ap.add_argument("--height", type=int, default=150)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--poses", type=str, default=None)
    ap.add_argument("--focal", type=float, default=None)
    ap.add_argument("--near", type=float, default=0.0)
    ap.add_argument("--far", type=float, default=1.0)
    ap.add_argument("--ndc", action="store_true")
    ap.add_argument("--chunks", type=int, default=4096)
    ap.add_argument("--num-samples", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = checkpoint.load_reference_checkpoint(a.checkpoint, device=dev, num_samples=a.num_samples)
    poses = np.load(a.poses).astype(np.float32) if a.poses else sweep_poses(a.views)
    focal = a.focal or 0.9 * a.width
    os.makedirs(a.out_dir, exist_ok=True)
    t0 = time.perf_counter()
    for i, pose in enumerate(poses):
        img, dist, acc = model.render_view(pose, a.height, a.width, focal, a.near, a.far, ndc=a.ndc, chunks=a.chunks)
        png.write_png(os.path.join(a.out_dir, f"rgb_{i:04d}.png"), img)
        png.write_png(os.path.join(a.out_dir, f"dist_{i:04d}.png"), to8b(visualize_depth(dist, acc, a.near, a.far)))
        png.write_png(os.path.join(a.out_dir, f"norm_{i:04d}.png"), to8b(visualize_normals(dist, acc)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"views": len(poses), "width": a.width, "height": a.height, "seconds": round(dt, 3),
                      "rays_per_s": round(len(poses) * a.width * a.height / dt, 1), "out_dir": a.out_dir}))


if __name__ == "__main__":
    main()
