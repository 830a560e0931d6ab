#!/usr/bin/env python3
"""Multi-GPU path check, one process per GPU over RCCL (backend "nccl"):

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
         tools/dist_check.py

Every rank renders the same small synthetic frame with `render_image_sharded` (chunk-granular ray sharding +
one all-gather of the [rays,5] pixel block) and compares it bit for bit with its own single-GPU `render_image`.
Works with N = 1 as well (the all-gather then has a single participant).

  ... tools/dist_check.py --backend gloo --same-gpu     (tests/: N ranks of the HIP renderer on ONE GPU; RCCL refuses
  two ranks on one device, so the collectives go through gloo with host staging - the compute path is unchanged)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402
from mipnerf360_amd.distributed import render_image_sharded  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses cuda:0")
    ap.add_argument("--full-width", action="store_true", help="256 / 1024 hidden units instead of 64 / 128")
    args = ap.parse_args()
    local_rank = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    h, w, n, chunks = 40, 53, 32, 256
    hp, hn = (256, 1024) if args.full_width else (64, 128)
    sd = synthetic.make_state_dict(hp, hn, seed=3)
    m = mipNeRF360(num_samples=n, hidden_proposal=hp, hidden_nerf=hn, device=dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    r = synthetic.make_rays("garden", h * w, seed=21)
    rays = Rays(*[torch.from_numpy(r[k]) for k in synthetic.RAY_FIELDS])
    single = m.render_image(rays, h, w, chunks=chunks)
    sharded = render_image_sharded(m, rays, h, w, chunks=chunks)
    ok = all(np.array_equal(a, b) for a, b in zip(single, sharded))
    # the same from a pose: every rank generates only the rays of its own block of chunks (m360_generate_rays_span)
    from mipnerf360_amd import ops
    from mipnerf360_amd.distributed import render_view_sharded
    pose = np.concatenate([np.eye(3), np.array([[0.05], [-0.02], [0.1]])], 1).astype(np.float32)
    view1 = m.render_view(pose, h, w, 0.9 * w, 0.0, 1.0, ndc=True, chunks=chunks)
    rgb_s, dist_s, acc_s = render_view_sharded(m, pose, h, w, 0.9 * w, 0.0, 1.0, ndc=True, chunks=chunks)
    ok = ok and np.array_equal(view1[0], ops.to8b(rgb_s).reshape(h, w, 3).cpu().numpy())
    ok = ok and np.array_equal(view1[1], dist_s.reshape(h, w).cpu().numpy()) and np.array_equal(view1[2], acc_s.reshape(h, w).cpu().numpy())
    # one logical batch split over the ranks, global contraction norm rebuilt with an all-reduce
    from mipnerf360_amd.distributed import forward_sharded
    world, rank = dist.get_world_size(), dist.get_rank()
    B = 96 * world
    rb = synthetic.make_rays("garden", B, seed=33)
    rb["origins"] = rb["origins"] * 3.0      # means outside the unit ball: the norm matters
    whole = Rays(*[torch.from_numpy(rb[k]).to(dev) for k in synthetic.RAY_FIELDS])
    mine = Rays(*[f[rank * 96:(rank + 1) * 96] for f in whole])
    with torch.no_grad():
        ref = m(whole)
    got = forward_sharded(m, mine)
    err = max(float((g - r[rank * 96:(rank + 1) * 96]).abs().max()) for g, r in zip(got, ref))
    ok = ok and err <= (0.0 if world == 1 else 2e-6)
    if rank == 0:
        print(f"dist_check forward_sharded max |diff| vs whole-batch forward: {err:.2e}")
    flag = torch.tensor([1 if ok else 0], device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if dist.get_rank() == 0:
        print(f"dist_check world={dist.get_world_size()} sharded==single: {bool(flag.item())}")
    dist.destroy_process_group()
    if not flag.item():
        raise SystemExit(1)


if __name__ == "__main__":
    main()
