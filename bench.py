#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s at 128 samples/ray (BASELINE.json).

One "step" = one full two-stage forward of the hot path (mipNeRF360.forward, model.py:247-252
of the reference) over one synthetic garden-like batch of 4096 rays x 128 samples with the
full-width fp32 proposal (4x256) + NeRF (8x1024) MLPs — BASELINE.json configs[1].  Rays and
weights are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N ...

For N > 1 every rank renders its own 4096-ray batch (weak scaling; rays shard with no data-path
collective) and the rendered pixels ([4096,5] fp32 per rank) are all-gathered over RCCL each step,
as the path's one exchange step.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_GPU = 4096
SAMPLES = 128
HP, HN = 256, 1024
FLOPS_PER_SAMPLE = 2 * (58 * 256 + 3 * 256 * 256 + 256) + 2 * (58 * 1024 + 7 * 1024 * 1024 + 4 * 1024)  # 15,230,464
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (never the 2:1-sparsity figure)


def cpu_baseline(sd_np, rays_np, n_rays):
    """The oracle (CPU restatement of the reference path, kind="port") on the first n_rays of the batch."""
    import torch
    from oracle import ref_path as O
    sd = O.to_torch_state_dict(sd_np)
    sub = {k: v[:n_rays] for k, v in rays_np.items()}
    hp = O.Hyper(num_samples=SAMPLES)
    O.forward(O.rays_from_numpy({k: v[:32] for k, v in rays_np.items()}), sd, hp)  # page in / thread pool warm-up
    t0 = time.perf_counter()
    out = O.forward(O.rays_from_numpy(sub), sd, hp)
    dt = time.perf_counter() - t0
    return n_rays / dt, dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays of the batch timed on the host CPU (0 = skip)")
    ap.add_argument("--mlp-dtype", choices=("fp32", "bf16"), default="fp32",
                    help="fp32 = the headline configuration (BASELINE configs[1]); bf16 = opt-in reduced-precision MLP "
                         "(configs[4] kernel family) - reported with dtype bf16, never comparable to the fp32 line")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from mipnerf360_amd import _lib, synthetic
    from mipnerf360_amd.intern.ray import Rays
    from mipnerf360_amd.model import mipNeRF360

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lib = _lib.lib()
    sd_np = synthetic.make_state_dict(HP, HN, seed=0)
    rays_np = synthetic.make_rays("garden", RAYS_PER_GPU, seed=1 + rank)
    bf16 = args.mlp_dtype == "bf16"
    model = mipNeRF360(randomized=False, num_samples=SAMPLES, hidden_proposal=HP, hidden_nerf=HN, white_bkgd=False,
                       device=dev, mlp_dtype=args.mlp_dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    rays = Rays(*[torch.from_numpy(rays_np[k]).to(dev) for k in synthetic.RAY_FIELDS])
    gathered = torch.empty(world * RAYS_PER_GPU, 5, device=dev) if world > 1 else None

    def step():
        with torch.no_grad():  # rendering, as in render_image (model.py:261); with grad enabled the mirrors keep a training tape
            rgb, d, a = model(rays)  # the public forward: (rgb[B,3], distance[B], acc[B])
        if world > 1:
            pixels = torch.cat([rgb, d[:, None], a[:, None]], 1)  # 20 B per ray
            dist.all_gather_into_tensor(gathered, pixels)
        return rgb, d, a

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.check(lib.m360_prof_enable(16 * max(args.steps, 1)), "m360_prof_enable")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel (hidden x hidden fp32-MFMA linear), HIP events recorded on
    # the launch stream inside the timed region by libm360 itself
    durs = []
    ms, M_, n_, k_ = C.c_float(), C.c_long(), C.c_int(), C.c_int()
    for i in range(lib.m360_prof_count()):
        _lib.check(lib.m360_prof_read(i, C.byref(ms), C.byref(M_), C.byref(n_), C.byref(k_)), "m360_prof_read")
        if n_.value == HN and abs(k_.value) == HN and M_.value == RAYS_PER_GPU * SAMPLES:  # k < 0 marks bf16 launches
            durs.append(ms.value)
    lib.m360_prof_enable(0)
    roofline = None
    if durs:
        flops = 2.0 * RAYS_PER_GPU * SAMPLES * HN * HN
        avg_ms = sum(durs) / len(durs)
        achieved = flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from a separate rocprofv3 --pmc run
        if os.path.exists(tpath) and not bf16:
            traffic = json.load(open(tpath)).get("linear_f32_mfma_1024x1024_bytes_per_launch")
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        kname = "linear_bf16_mfma_persist_kernel" if bf16 else "linear_f32_mfma_persist_kernel"
        roofline = {"bound": "mfma", "kernel": f"{kname} (1024x1024 layer, M=524288)", "achieved": round(achieved, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "traffic": traffic, "launches": len(durs), "avg_launch_ms": round(avg_ms, 4),
                    "flops_per_launch": flops}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_rays = world * RAYS_PER_GPU * args.steps
    value = total_rays / elapsed
    line = {
        "metric": "rendered rays/sec at 128 samples/ray",
        "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": "nerf_360/garden-like synthetic NDC ray batch (near 0 / far 1), 4096 rays x 128 samples/ray "
                               "per GPU, proposal 4x256 + NeRF 8x1024 MLPs in " + ("bf16 (fp32 accumulate)" if bf16 else "fp32") + " on MFMA, random-init Kaiming weights "
                               "(BASELINE.json configs[1])",
                   "rays_per_gpu": RAYS_PER_GPU, "samples_per_ray": SAMPLES,
                   "parallelism": f"rays sharded over {world} GPU(s), replicated weights" +
                                  (", RCCL all-gather of [4096,5] pixels per step" if world > 1 else ""),
                   "flops_per_ray": FLOPS_PER_SAMPLE * SAMPLES,
                   "whole_path_tflops": round(value * FLOPS_PER_SAMPLE * SAMPLES / 1e12, 2)},
        "roofline": roofline,
    }
    if world == 1 and args.cpu_rays > 0:
        n_cpu = min(args.cpu_rays, RAYS_PER_GPU)
        cpu_rps, cpu_s, o = cpu_baseline(sd_np, rays_np, n_cpu)
        line["cpu_baseline"] = {"value": round(cpu_rps, 2), "unit": "rays/s", "cores": torch.get_num_threads(),
                                "kind": "port",
                                "sample": f"first {n_cpu} rays of the same 4096x128 batch as one chunk, 1 pass = {cpu_s:.1f} s "
                                          f"(oracle/ref_path.py, vectorised torch-CPU fp32, closed-form Jacobian; the "
                                          f"unmodified reference measured 17.9-19.1 rays/s on 8 cores, BASELINE.md)",
                                "host_cpus": os.cpu_count()}
        # parity + PSNR of the same sub-batch rendered as its own chunk on the GPU
        sub = Rays(*[f[:n_cpu].contiguous() for f in rays])
        with torch.no_grad():
            g_rgb, g_dist, g_acc = model(sub)
        mse = float(((g_rgb.cpu() - o[0]) ** 2).mean())
        line["parity"] = {"max_abs_rgb": float((g_rgb.cpu() - o[0]).abs().max()),
                          "max_abs_acc": float((g_acc.cpu() - o[2]).abs().max()),
                          "psnr_vs_cpu_db": round(-10.0 * float(np.log10(max(mse, 1e-20))), 2), "rays": n_cpu}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
