#!/usr/bin/env python3
"""End-to-end training on the HIP path with the reference's own loop body (train.py:53-84): a student
mip-NeRF 360 (random init) is fitted to pixels rendered by a fixed teacher network on synthetic rays.
No dataset is needed; the point is that model forward, losses, backward and the AdamW update all run on
the mirrors and that the optimisation actually descends.

    python tools/train_demo.py [--steps 150] [--rays 1024] [--samples 32] [--hidden 64 128]

Prints one JSON line with the PSNR trajectory (student render vs teacher pixels).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402
from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402


def run(steps=150, rays_n=1024, samples=32, hp=64, hn=128, lr=2e-3, dist_weight=0.01, seed=0, device="cuda:0", log_every=25,
        randomized=False, kind="garden", teacher_kind="kaiming", save=None, white_bkgd=False, mlp_dtype="fp32"):
    """teacher_kind="structured": the teacher carries the trained-like weights of fixture G19 (high-contrast colours, density
    shells), so the student is fitted to an image with signal.  save: write the trained student's state_dict in the reference's
    checkpoint layout (train.py:98-103) - fixture G20 is such a file, rendered by the reference itself."""
    dev = torch.device(device)
    torch.manual_seed(seed)
    kw = dict(randomized=False, num_samples=samples, hidden_proposal=hp, hidden_nerf=hn, white_bkgd=white_bkgd, device=dev)
    teacher = mipNeRF360(**kw)
    # train.py's default is randomized=True (config.py:14); mlp_dtype="bf16": the student trains on the bf16 matrix pipe (bf16 tape and
    # gradients in flight, fp32 accumulation, fp32 master weights + AdamW) - the teacher's pixels stay fp32
    student = mipNeRF360(**dict(kw, randomized=randomized, mlp_dtype=mlp_dtype))
    r = synthetic.make_rays(kind, rays_n, seed=300 + seed)
    t_sd = (synthetic.make_structured_state_dict(hp, hn, 100 + seed, r, samples) if teacher_kind == "structured"
            else synthetic.make_state_dict(hp, hn, seed=100 + seed))
    teacher.load_state_dict({k: torch.from_numpy(v) for k, v in t_sd.items()})
    student.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(hp, hn, seed=200 + seed).items()})
    rays = Rays(*[torch.from_numpy(r[f]).to(dev) for f in synthetic.RAY_FIELDS])
    with torch.no_grad():
        pixels, _, _ = teacher(rays)
    student.train()
    opt = torch.optim.AdamW(student.parameters(), lr=lr, weight_decay=0.0)
    traj = []
    t0 = time.perf_counter()
    for step in range(steps):
        for _ in range(2):                                              # train.py:55-65
            t_hat, w_hat = student.prop_net.forward(rays)
            _, _, _, t, w, _ = student.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
            loss_prop = Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
            opt.zero_grad()
            loss_prop.backward()
            opt.step()
        t_hat, w_hat = student.prop_net.forward(rays)                   # train.py:68-81
        final_rgbs, _, _, _, fine_weights, s_vals = student.nerf_net.forward(rays, t_vals=t_hat.detach(),
                                                                             coarse_weights=w_hat.detach())
        loss_nerf, psnr = Loss_nerf(input=final_rgbs, target=pixels)
        loss_dist = Loss_dist(s_vals=s_vals, weights=fine_weights)
        loss_all = loss_nerf + dist_weight * loss_dist
        opt.zero_grad()
        loss_all.backward()
        opt.step()
        if step % log_every == 0 or step == steps - 1:
            traj.append({"step": step, "psnr": round(float(psnr), 3), "loss_prop": round(float(loss_prop.detach()), 4),
                         "loss_dist": round(float(loss_dist.detach()), 5)})
    torch.cuda.synchronize()
    out = {"steps": steps, "rays": rays_n, "samples": samples, "hidden": [hp, hn], "seconds": round(time.perf_counter() - t0, 2),
           "kind": kind, "teacher": teacher_kind, "rays_seed": 300 + seed, "white_bkgd": bool(white_bkgd), "mlp_dtype": mlp_dtype,
           "target_rgb_std_over_rays": [round(float(v), 4) for v in pixels.std(0)], "trajectory": traj}
    if save:
        from mipnerf360_amd import checkpoint
        student.eval()
        with torch.no_grad():
            rgb, _, acc = student(rays)
        out["student_rgb_std_over_rays"] = [round(float(v), 4) for v in rgb.std(0)]
        out["student_acc_range"] = [round(float(acc.min()), 4), round(float(acc.max()), 4)]
        os.makedirs(os.path.dirname(os.path.abspath(save)), exist_ok=True)
        torch.save(checkpoint.to_reference_state_dict(student), save)
        out["saved"] = save
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--hidden", type=int, nargs=2, default=[64, 128])
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--randomized", action="store_true", help="stratified jitter in both stages, as train.py does by default")
    ap.add_argument("--kind", default="garden", choices=("garden", "lego", "mixed"))
    ap.add_argument("--teacher", default="kaiming", choices=("kaiming", "structured"))
    ap.add_argument("--white-bkgd", action="store_true")
    ap.add_argument("--save", default=None, help="write the trained student's state_dict (reference checkpoint layout) here")
    ap.add_argument("--mlp-dtype", default="fp32", choices=("fp32", "bf16"), help="precision the student trains in")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.rays, a.samples, a.hidden[0], a.hidden[1], a.lr, randomized=a.randomized, kind=a.kind,
                         teacher_kind=a.teacher, save=a.save, white_bkgd=a.white_bkgd, mlp_dtype=a.mlp_dtype)))


if __name__ == "__main__":
    main()
