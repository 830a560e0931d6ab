#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s at 128 samples/ray (BASELINE.json).

One "step" = one full two-stage forward of the hot path (mipNeRF360.forward, model.py:247-252
of the reference) over one synthetic garden-like batch of 4096 rays x 128 samples with the
full-width fp32 proposal (4x256) + NeRF (8x1024) MLPs — BASELINE.json configs[1].  Rays and
weights are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N ...

For N > 1 every rank renders its own batch (weak scaling; rays shard with no data-path collective) and
the rendered pixels ([rays,5] fp32 per rank) are all-gathered over RCCL each step, as the path's one
exchange step.  Rank 0 prints ONE JSON line.

Other named workloads (never the default line): `--config c5` = BASELINE configs[4]'s per-GPU shape
(8192 rays x 256 samples, bf16 MLP), `--mlp-dtype bf16` = configs[1] with the opt-in bf16 MLP.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HP, HN = 256, 1024
FLOPS_PER_SAMPLE = 2 * (58 * 256 + 3 * 256 * 256 + 256) + 2 * (58 * 1024 + 7 * 1024 * 1024 + 4 * 1024)  # 15,230,464
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (never the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)

CONFIGS = {
    # name: (rays per GPU, samples, default MLP dtype, metric, workload description)
    "c2": (4096, 128, "fp32", "rendered rays/sec at 128 samples/ray",
           "nerf_360/garden-like synthetic NDC ray batch (near 0 / far 1), 4096 rays x 128 samples/ray per GPU, "
           "proposal 4x256 + NeRF 8x1024 MLPs in {mlp} on MFMA, random-init Kaiming weights (BASELINE.json configs[1])"),
    "c5": (8192, 256, "bf16", "rendered rays/sec at 256 samples/ray",
           "nerf_360/bicycle-like synthetic NDC ray batch (near 0 / far 1), 8192 rays x 256 samples/ray per GPU, "
           "proposal 4x256 + NeRF 8x1024 MLPs in {mlp} on MFMA, random-init Kaiming weights (BASELINE.json configs[4] "
           "shape on each GPU)"),
}

# source files that define the dominant kernel: profiles/traffic.json is only trusted while they are unchanged
TRAFFIC_SOURCES = ("mipnerf360_amd/csrc/m360_linear_hd.hip.h", "mipnerf360_amd/csrc/m360_linear_hd_gen.inc",
                   "mipnerf360_amd/csrc/m360_linear.hip", "mipnerf360_amd/csrc/m360_common.hip.h")


def kernel_source_sha():
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def cpu_baseline(sd_np, rays_np, n_rays, samples, passes=3):
    """The oracle (CPU restatement of the reference path, kind="port") on the first n_rays of the batch:
    median of `passes` timed passes after one small warm-up."""
    from oracle import ref_path as O
    sd = O.to_torch_state_dict(sd_np)
    sub = {k: v[:n_rays] for k, v in rays_np.items()}
    hp = O.Hyper(num_samples=samples)
    O.forward(O.rays_from_numpy({k: v[:32] for k, v in rays_np.items()}), sd, hp)  # page in / thread pool warm-up
    times, out = [], None
    for _ in range(passes):
        t0 = time.perf_counter()
        out = O.forward(O.rays_from_numpy(sub), sd, hp)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return n_rays / dt, times, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 = the headline workload (BASELINE configs[1]); c5 = BASELINE configs[4]'s shape (bf16)")
    ap.add_argument("--cpu-rays", type=int, default=None,
                    help="rays of the batch timed on the host CPU, 3 passes (default 1024 for c2, 0 = skip)")
    ap.add_argument("--mlp-dtype", choices=("fp32", "bf16"), default=None,
                    help="override the workload's MLP precision (c2: fp32 = the headline; bf16 = opt-in reduced-precision "
                         "MLP, reported with dtype bf16 and never comparable to the fp32 line)")
    args = ap.parse_args()
    n_rays, samples, mlp_default, metric, workload = CONFIGS[args.config]
    mlp_dtype = args.mlp_dtype or mlp_default
    if args.cpu_rays is None:
        args.cpu_rays = 1024 if args.config == "c2" else 0

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

    sd_np = synthetic.make_state_dict(HP, HN, seed=0)
    rays_np = synthetic.make_rays("garden", n_rays, seed=1 + rank)
    bf16 = mlp_dtype == "bf16"
    model = mipNeRF360(randomized=False, num_samples=samples, hidden_proposal=HP, hidden_nerf=HN, white_bkgd=False,
                       device=dev, mlp_dtype=mlp_dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.eval()  # rendering: weights are packed once, outside the timed region
    rays = Rays(*[torch.from_numpy(rays_np[k]).to(dev) for k in synthetic.RAY_FIELDS])
    gathered = torch.empty(world * n_rays, 5, device=dev) if world > 1 else None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        with torch.no_grad():  # rendering, as in render_image (model.py:261); with grad enabled the mirrors keep a training tape
            rgb, d, a = model(rays)  # the public forward: (rgb[B,3], distance[B], acc[B])
        if i is not None:
            ev[i][1].record()
        if world > 1:
            pixels = torch.cat([rgb, d[:, None], a[:, None]], 1)  # 20 B per ray
            dist.all_gather_into_tensor(gathered, pixels)
        if i is not None:
            ev[i][2].record()
        return rgb, d, a

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = _lib.Prof(40 * max(args.steps, 1))  # caller-owned HIP-event recorder: every kernel of the stage drivers
    model.set_prof(prof)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    model.set_prof(None)
    compute_ms = [e[0].elapsed_time(e[1]) for e in ev]
    gather_ms = [e[1].elapsed_time(e[2]) for e in ev]
    step_ms = [e[0].elapsed_time(e[2]) for e in ev]
    per_rank = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([statistics.median(compute_ms), statistics.median(gather_ms)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "compute_ms_median": round(float(t[0]), 3), "all_gather_ms_median": round(float(t[1]), 3)}
                    for r, t in enumerate(allr)]

    # ---- per-kernel numbers from the event records (HIP events on the launch stream, inside the timed region)
    recs = prof.records()
    prof.close()
    S = n_rays * samples
    lin_kind = _lib.K_LINEAR_BF16 if bf16 else _lib.K_LINEAR
    durs = [r["ms"] for r in recs if r["kind"] == lin_kind and r["n_pad"] == HN and r["k_pad"] == HN and r["M"] == S]
    roofline = None
    if durs:
        flops = 2.0 * S * HN * HN
        avg_ms = sum(durs) / len(durs)
        achieved = flops / (avg_ms * 1e-3) / 1e12
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from separate rocprofv3 --pmc passes
        if os.path.exists(tpath) and not bf16 and args.config == "c2":
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha256") == kernel_source_sha():
                traffic = tj.get("linear_f32_mfma_1024x1024_bytes_per_launch")
            else:
                traffic_note = "profiles/traffic.json was measured on different kernel sources (stale): not reported"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        kname = "linear_bf16_pp_kernel" if bf16 else "linear_f32_hd_kernel"
        roofline = {"bound": "mfma", "kernel": f"{kname} (1024x1024 layer, M={S})", "achieved": round(achieved, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "traffic": traffic, "launches": len(durs), "avg_launch_ms": round(avg_ms, 4),
                    "median_launch_ms": round(statistics.median(durs), 4), "flops_per_launch": flops}
        if traffic_note:
            roofline["traffic_note"] = traffic_note
    # HBM-bound kernels of the path: algorithmic bytes / mean launch duration.  The finishers read the
    # last layer's partial head sums [S, slots, heads] fp32 for the rows the fused epilogue covered, activation rows else.
    el = 2 if bf16 else 4
    in_pad = 64
    lib = _lib.lib()

    def finish_in_bytes(width, heads):
        fused = int(lib.m360_linear_heads_fused_rows(S, width, int(bf16)))
        return fused * int(lib.m360_linear_heads_slots(width, int(bf16))) * heads * 4 + (S - fused) * width * el

    hbm_kernels = {}
    fused_last = [r["ms"] for r in recs if r["kind"] == _lib.K_LINEAR_HEADS and r["n_pad"] == HN]
    if fused_last:
        ms = sum(fused_last) / len(fused_last)
        hbm_kernels["nerf_last_layer_fused_heads"] = {"avg_launch_ms": round(ms, 4), "tflops": round(2.0 * S * HN * HN / ms / 1e9, 1),
                                                      "launches": len(fused_last), "note": "MFMA-bound; listed for completeness"}
    for kind, name, nbytes in (
            (_lib.K_ENCODE, "encode_features", S * in_pad * el + n_rays * (48 + 4 * (samples + 1))),
            (_lib.K_PROP_FINISH, "prop_finish", finish_in_bytes(HP, 1) + n_rays * (4 * (samples + 1) + 12 + 4 * samples + 4 * (samples + 1))),
            (_lib.K_NERF_FINISH, "nerf_finish", finish_in_bytes(HN, 4) + n_rays * (4 * (samples + 1) + 12 + 20))):
        d = [r["ms"] for r in recs if r["kind"] == kind]
        if d:
            ms = sum(d) / len(d)
            hbm_kernels[name] = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes": nbytes,
                                 "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3), "launches": len(d)}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_rays = world * n_rays * args.steps
    value = total_rays / elapsed
    med_ms = statistics.median(step_ms)
    line = {
        "metric": metric,
        "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "ms_per_step_median": round(med_ms, 3), "value_at_median_step": round(world * n_rays / (med_ms * 1e-3), 1),
        "config": {"workload": workload.format(mlp="bf16 (fp32 accumulate)" if bf16 else "fp32"),
                   "name": args.config, "rays_per_gpu": n_rays, "samples_per_ray": samples,
                   "parallelism": f"rays sharded over {world} GPU(s), replicated weights" +
                                  (f", RCCL all-gather of [{n_rays},5] pixels per step" if world > 1 else ""),
                   "flops_per_ray": FLOPS_PER_SAMPLE * samples,
                   "whole_path_tflops": round(value * FLOPS_PER_SAMPLE * samples / 1e12, 2)},
        "roofline": roofline,
        "hbm_kernels": hbm_kernels,
    }
    if per_rank:
        line["per_rank"] = per_rank
    if world == 1 and args.cpu_rays > 0:
        n_cpu = min(args.cpu_rays, n_rays)
        cpu_rps, cpu_times, o = cpu_baseline(sd_np, rays_np, n_cpu, samples)
        line["cpu_baseline"] = {"value": round(cpu_rps, 2), "unit": "rays/s", "cores": torch.get_num_threads(),
                                "kind": "port",
                                "sample": f"first {n_cpu} rays of the same {n_rays}x{samples} batch as one chunk, median of "
                                          f"{len(cpu_times)} passes ({', '.join(f'{t:.1f}' for t in cpu_times)} s) "
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
