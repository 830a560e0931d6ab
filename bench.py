#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s at 128 samples/ray (BASELINE.json).

One "step" = one full two-stage forward of the hot path (mipNeRF360.forward, model.py:247-252
of the reference) over one synthetic garden-like batch of 4096 rays x 128 samples with the
full-width fp32 proposal (4x256) + NeRF (8x1024) MLPs — BASELINE.json configs[1].  Rays and
weights are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python bench.py --gpus 8                      # starts 8 fresh rank processes itself (torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N ...  # or launched from outside: RANK / LOCAL_RANK / WORLD_SIZE from the env

Launching: with WORLD_SIZE unset and --gpus N > 1 this process touches no GPU; it starts `python -m torch.distributed.run`
as a CHILD (which starts one fresh process per rank), relays rank 0's single JSON line and exits with the child's code.
Nothing is ever exec'ed in place (`--dry-launch` prints the child command instead of running it).

Default workload (c2, weak scaling): every rank renders its own 4096-ray batch (rays shard with no data-path
collective) and the rendered pixels (20 B per ray) are all-gathered over RCCL each step, as the path's one exchange
step.  Rank 0 prints ONE JSON line; for N > 1 it carries `rccl.ranks`, the world size read back from an all-reduce.
After the timed region the same processes render ONE 1237 x 822 frame sharded over the N ranks at 64 proposal + 128 NeRF
samples per ray (BASELINE configs[2] as written; configs[3] for N > 1, strong scaling; `strong_scaling_frame` in the line,
`--frame-steps 0` to skip), and at N = 1 the process then times the other single-GPU workloads for a few steps each
(`named_workloads`: configs[4]'s shape in bf16, configs[1] in bf16 and bf16x3, one train.py iteration; `--no-named` to skip).

Other named workloads (never the default line):
  --config c4          BASELINE configs[3] as the timed workload: frames of 1237 x 822 rays, each rank generates and
                       renders the rays of its own block of 4096-ray chunks, one pixel all-gather per frame, overlapped
                       with the next frame's compute on a side stream (the serial numbers are reported beside it);
  --config c5          BASELINE configs[4]'s per-GPU shape (8192 rays x 256 samples, bf16 MLP);
  --mlp-dtype bf16     configs[1] with the opt-in bf16 MLP.
  --backend gloo       (diagnostics) ranks may share a GPU, collectives staged through the host: exercises the launcher
                       and the sharded code on a 1-GPU box; never a scaling measurement.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HP, HN = 256, 1024
FLOPS_PER_SAMPLE = 2 * (58 * 256 + 3 * 256 * 256 + 256) + 2 * (58 * 1024 + 7 * 1024 * 1024 + 4 * 1024)  # 15,230,464
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (never the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
FRAME_W, FRAME_H, FRAME_CHUNKS = 1237, 822, 4096  # BASELINE configs[2]/[3]: nerf_360/garden at 4x downscale

CONFIGS = {
    # name: (rays per GPU, samples, default MLP dtype, metric, workload description)
    "c2": (4096, 128, "fp32", "rendered rays/sec at 128 samples/ray",
           "nerf_360/garden-like synthetic NDC ray batch (near 0 / far 1), 4096 rays x 128 samples/ray per GPU, "
           "proposal 4x256 + NeRF 8x1024 MLPs in {mlp} on MFMA, random-init Kaiming weights (BASELINE.json configs[1])"),
    "c4": (FRAME_CHUNKS, 128, "fp32", "rendered rays/sec at 128 samples/ray",
           "nerf_360/garden-sized 1237x822 frame (1 016 814 rays, synthetic forward-facing NDC pose, near 0 / far 1) in "
           "249 chunks of 4096 rays x 128 samples/ray, whole chunks sharded over the GPUs, rays generated on each GPU for "
           "its own chunks, proposal 4x256 + NeRF 8x1024 MLPs in {mlp} on MFMA, random-init Kaiming weights, one pixel "
           "all-gather per frame (BASELINE.json configs[3]; configs[2] at 1 GPU)"),
    "c5": (8192, 256, "bf16", "rendered rays/sec at 256 samples/ray",
           "nerf_360/bicycle-like synthetic NDC ray batch (near 0 / far 1), 8192 rays x 256 samples/ray per GPU, "
           "proposal 4x256 + NeRF 8x1024 MLPs in {mlp} on MFMA, random-init Kaiming weights (BASELINE.json configs[4] "
           "shape on each GPU)"),
}

# source files that define the dominant kernel: profiles/traffic.json is only trusted while they are unchanged
TRAFFIC_SOURCES = ("mipnerf360_amd/csrc/m360_linear_hd.hip.h", "mipnerf360_amd/csrc/m360_linear_hd_gen.inc",
                   "mipnerf360_amd/csrc/m360_linear.hip", "mipnerf360_amd/csrc/m360_common.hip.h")


# ... and the dominant kernel of --mlp-dtype bf16 (the one-wave ring kernel)
TRAFFIC_SOURCES_BF16 = ("mipnerf360_amd/csrc/m360_linear_bf16_w16.hip.h", "mipnerf360_amd/csrc/m360_linear_bf16_w16_gen.inc",
                        "mipnerf360_amd/csrc/m360_linear.hip", "mipnerf360_amd/csrc/m360_common.hip.h")


# ... and the gradient GEMMs of the training path (weight gradient: the tn kernels; input gradient: the forward kernels on transposed packings)
TRAFFIC_SOURCES_TRAIN = ("mipnerf360_amd/csrc/m360_linear.hip", "mipnerf360_amd/csrc/m360_common.hip.h", "mipnerf360_amd/csrc/m360_linear_tn.hip.h",
                         "mipnerf360_amd/csrc/m360_linear_tn_bf16.hip.h", "mipnerf360_amd/csrc/m360_linear_tn_bf16_w.hip.h",
                         "mipnerf360_amd/csrc/m360_linear_persist.hip.h", "mipnerf360_amd/csrc/m360_linear_bf16_w16.hip.h",
                         "mipnerf360_amd/csrc/m360_linear_bf16_w16_gen.inc")


def training_gemm_traffic(dtype_key):
    """{'wgrad': {...}, 'dgrad': {...}[, 'relu_mask': {...}]} of profiles/traffic.json's training_gemms for 'bf16' / 'fp32' (fabric-side counter
    bytes per launch from separate --pmc passes, tools/gpu_session.sh pmctrain), or {} when absent or measured on other kernel sources."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}
    tg = json.load(open(tpath)).get("training_gemms") or {}
    if tg.get("kernel_source_sha256") != kernel_source_sha(TRAFFIC_SOURCES_TRAIN):
        return {}
    return tg.get(dtype_key) or {}


def kernel_source_sha(sources=None):
    h = hashlib.sha256()
    for rel in (sources or TRAFFIC_SOURCES):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; c4: 3 frames)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5; c4: 1 frame)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 = the headline workload (BASELINE configs[1]); c4 = one frame sharded over the GPUs "
                         "(configs[3], strong scaling); c5 = BASELINE configs[4]'s shape (bf16)")
    ap.add_argument("--cpu-rays", type=int, default=None,
                    help="rays of the batch timed on the host CPU (default: the whole 4096-ray batch for c2, 0 = skip)")
    ap.add_argument("--mlp-dtype", choices=("fp32", "bf16", "bf16x3"), default=None,
                    help="override the workload's MLP precision (c2: fp32 = the headline; bf16 = opt-in reduced-precision "
                         "MLP, reported with dtype bf16 and never comparable to the fp32 line; bf16x3 = two bf16 terms per "
                         "value, three MFMA passes per product: inside the fp32 render tolerance, reported with dtype bf16x3)")
    ap.add_argument("--frame-steps", type=int, default=None,
                    help="c2 only: frames of the 1237x822 strong-scaling workload rendered AFTER the timed region and "
                         "reported as `strong_scaling_frame` (default 1; 0 = skip)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL (the product); gloo = diagnostics only (ranks may share a GPU, host-staged collectives)")
    ap.add_argument("--frame-size", type=str, default=None, metavar="WxH",
                    help="(diagnostics) frame size of the c4 workload / strong_scaling_frame instead of 1237x822; the line "
                         "then says so in config.workload")
    ap.add_argument("--no-named", action="store_true",
                    help="c2 / fp32 / 1 GPU only: skip the `named_workloads` block (the other single-GPU configs, a few steps each)")
    ap.add_argument("--no-chain", action="store_true", help="(diagnostics, bf16 mode) M360_TUNE_NO_HIDDEN_CHAIN per call: six launches for the six hidden NeRF layers")
    ap.add_argument("--plain-rows", action="store_true",
                    help="(diagnostics, bf16 modes) M360_TUNE_PLAIN_ROWS per call: plain instead of paired rows between the layers - same bits; "
                         "the line then carries config.plain_rows")
    ap.add_argument("--fail-frame-setup-on-rank", type=int, default=-1, metavar="R",
                    help="(test hook) rank R raises while setting the frame leg up: every rank must then skip the leg together (no rank is left "
                         "waiting in a collective) and the run must still end with its headline line and rc 0")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the child command as JSON and exit")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------- launcher
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(args, argv, port):
    passed = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + passed


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process tree can use, counted WITHOUT initialising HIP (the parent must never touch a GPU): the KFD topology
    lists every node with its `simd_count` (0 = a CPU node); ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES narrow the set like the runtime does.  None when the topology cannot be read (no driver)."""
    try:
        nodes = sorted(os.listdir(sysfs), key=lambda v: int(v) if v.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(sysfs, node, "properties")) as f:
                props = dict(line.split(None, 1) for line in f if len(line.split(None, 1)) == 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(listed))
    return n


def launch(args, argv) -> int:
    """Parent of an N > 1 run started without a launcher.  Imports no torch, touches no GPU: the ranks are fresh child
    processes of `torch.distributed.run`, itself a child of this process (never an exec in place)."""
    cmd = child_command(args, argv, free_port())
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "ranks": args.gpus}), flush=True)
        return 0
    if args.backend == "nccl":  # RCCL wants one device per rank: refuse here, before N processes have initialised a GPU each
        have = visible_gpu_count()
        if have is not None and args.gpus > have:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible here (KFD topology / *_VISIBLE_DEVICES); "
                  f"RCCL needs one device per rank (--backend gloo lets ranks share a GPU, diagnostics only)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    # own session: if this process is told to stop (a driver's timeout), the whole rank tree goes with it
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)

    def stop(signum, _frame):
        try:
            os.killpg(proc.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        raise SystemExit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop)
    line = None
    for out in proc.stdout:  # ranks' stderr goes straight through; stdout is scanned for rank 0's JSON line
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0:
        print(f"bench.py: {args.gpus}-rank child run failed with exit code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(sd_np, rays_np, n_rays, samples):
    """The oracle (CPU restatement of the reference path, kind="port") on the first n_rays of the batch as ONE chunk:
    two timed passes for >= 2048 rays (about 20 s each on the GPU box's 128 threads; the single passes of rounds 1-3 read
    175 / 191 / 210 rays/s, +-10 %: both times are reported), else the median of 3, after a small warm-up."""
    from oracle import ref_path as O
    sd = O.to_torch_state_dict(sd_np)
    sub = {k: v[:n_rays] for k, v in rays_np.items()}
    hp = O.Hyper(num_samples=samples)
    O.forward(O.rays_from_numpy({k: v[:32] for k, v in rays_np.items()}), sd, hp)  # page in / thread pool warm-up
    times, out = [], None
    for _ in range(2 if n_rays >= 2048 else 3):
        t0 = time.perf_counter()
        out = O.forward(O.rays_from_numpy(sub), sd, hp)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return n_rays / dt, times, out


# ------------------------------------------------------------------------------------------------- rank process
def rccl_log_summary(path):
    """What RCCL itself logged while it built this rank's communicator (NCCL_DEBUG=INFO, subsystems INIT + GRAPH): the rank count it
    printed, its channel count and how many peer connections went over which transport ("via P2P/IPC" = xGMI / PCIe peer access,
    "via SHM", "via NET/...").  Informational, never fatal: {} when there is no log of ours or nothing in it parses."""
    import re
    out = {}
    try:
        if not path or not os.path.exists(path):
            return out
        text = open(path, errors="replace").read()
        via = {}
        for m in re.finditer(r"\bvia ([A-Za-z0-9_/]+)", text):
            via[m.group(1)] = via.get(m.group(1), 0) + 1
        if via:
            out["transports"] = via
        m = re.search(r"nranks (\d+)", text)
        if m:
            out["log_nranks"] = int(m.group(1))
        ch = re.findall(r"(\d+) coll channels", text)
        if ch:
            out["coll_channels"] = int(ch[-1])
        m = re.search(r"(RCCL version[^\n]*|NCCL version[^\n]*)", text)
        if m:
            out["log_version"] = m.group(1).strip()[:120]
        out["log_lines"] = text.count("\n")
    except Exception as e:  # noqa: BLE001
        out = {"log_error": f"{type(e).__name__}: {e}"[:200]}
    return out


class Comm:
    """The few collectives bench.py itself needs (barrier, max / gather of timings), on RCCL or - diagnostics - gloo."""

    def __init__(self, world, rank, dev, backend):
        import torch
        import torch.distributed as dist
        self.world, self.rank, self.dev, self.backend, self.dist, self.torch = world, rank, dev, backend, dist, torch
        self.info = None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            log = None
            if backend == "nccl":
                # RCCL's own account of what it set up (ranks, channels, the transport of every peer connection) goes into the line: its
                # INIT-level log is written to a file of ours unless the caller already directs NCCL_DEBUG somewhere
                if "NCCL_DEBUG" not in os.environ and "NCCL_DEBUG_FILE" not in os.environ:
                    log = f"/tmp/m360_rccl_rank{rank}_{os.getpid()}.log"
                    os.environ["NCCL_DEBUG"], os.environ["NCCL_DEBUG_SUBSYS"], os.environ["NCCL_DEBUG_FILE"] = "INFO", "INIT,GRAPH", log
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
            ones = torch.ones(1, device=self.cdev)
            dist.all_reduce(ones)  # the world size as the collective library saw it
            self.info = {"backend": backend, "ranks": int(ones.item())}
            if backend == "nccl":
                self.info["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                torch.cuda.synchronize()
                self.info.update(rccl_log_summary(log))
            else:
                self.info["note"] = "diagnostics backend: collectives staged through the host, NOT a scaling measurement"

    @property
    def cdev(self):
        return self.dev if self.backend == "nccl" else self.torch.device("cpu")

    def barrier(self):
        if self.world > 1:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.dev.index])
            else:
                self.dist.barrier()

    def fence(self):
        self.barrier()
        self.torch.cuda.synchronize()

    def max(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def gather_floats(self, vals):
        """-> list over ranks of the list `vals`"""
        if self.world == 1:
            return [list(vals)]
        mine = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.cdev)
        allr = [self.torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allr, mine)
        return [[float(x) for x in t] for t in allr]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def roofline_from_records(recs, S, bf16, config_name, _lib, x3=False):
    lin_kind = _lib.K_LINEAR_BF16 if bf16 else _lib.K_LINEAR
    kk = 3 * HN if x3 else HN  # bf16x3: one contraction of length 3K (xh wh + xl wh + xh wl)
    # the 1024 x 1024 layers: one launch over all S rows
    hits = [r for r in recs if r["kind"] == lin_kind and r["n_pad"] == HN and r["k_pad"] == kk]
    chained = [r for r in recs if r["kind"] == lin_kind and r["n_pad"] == HN and r["k_pad"] == 6 * kk] if bf16 else []
    nlay = 1
    if chained and (not hits or max(r["M"] for r in chained) >= max(r["M"] for r in hits)):
        hits, nlay = chained, 6  # bf16 / bf16x3 modes: the six hidden layers in ONE launch (m360_mlp_chain_bf16[x3]_safe), recorded as one kernel
    if not hits:
        return None
    rows = max(r["M"] for r in hits)
    durs = [r["ms"] for r in hits if r["M"] == rows]
    flops = 2.0 * rows * HN * kk * nlay  # what the matrix pipe executes (bf16x3: three times the layer's algorithmic FLOPs)
    avg_ms = sum(durs) / len(durs)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch counter bytes from separate rocprofv3 --pmc passes
    if os.path.exists(tpath) and not bf16 and config_name in ("c2", "c4"):
        tj = json.load(open(tpath))
        if tj.get("kernel_source_sha256") == kernel_source_sha():
            traffic = tj.get("linear_f32_mfma_1024x1024_bytes_per_launch")
            traffic_note = ("fabric-side counter bytes per launch (FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, gfx950 corrections): "
                            "Infinity-Cache hits included, so W re-streamed per tile counts; algorithmic bytes are 4.30 GB")
        else:
            traffic_note = "profiles/traffic.json was measured on different kernel sources (stale): not reported"
    elif os.path.exists(tpath) and bf16 and config_name in ("c2", "c5"):
        # the reduced-precision workloads, each with its own counter passes (tools/gpu_session.sh pmc16): c2_bf16 (legacy key bf16_ring_kernel),
        # c5_bf16 (configs[4]'s per-GPU shape), c2_bf16x3
        tj = json.load(open(tpath))
        key = f"{config_name}_{'bf16x3' if x3 else 'bf16'}"
        tb = (tj.get("by_workload") or {}).get(key) or (tj.get("bf16_ring_kernel") if key == "c2_bf16" else None)
        if tb and tb.get("kernel_source_sha256") == kernel_source_sha(TRAFFIC_SOURCES_BF16):
            traffic = tb.get("bytes_per_launch")
            if tb.get("layers_per_launch", 1) != nlay or tb.get("rows", rows) != rows:
                traffic = None
            traffic_note = ("fabric-side counter bytes per launch (FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, gfx950 corrections); "
                            + ("the algorithmic bytes of the six-layer launch are rows in, rows out and six weight matrices; the five hidden "
                               "activations in between are counted here whenever they cross the fabric (Infinity-Cache hits included)" if nlay == 6 else
                               "the activation tile is read by the 4 column tiles of an XCD: L2 hits are not counted, Infinity-Cache hits are"))
        elif tb:
            traffic_note = f"profiles/traffic.json ({key}) was measured on different kernel sources (stale): not reported"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    kname = ("linear_bf16_w16_kernel<X3> (bf16x3: 3 MFMA passes per product)" if x3 else "linear_bf16_w16_kernel") if bf16 else "linear_f32_hd_kernel"
    shape = "six 1024x1024 layers in one launch (m360_mlp_chain_bf16)" if nlay == 6 else "1024x1024 layer"
    # algorithmic bytes of the dominant launch: rows in + rows out + the weight matrices (bf16x3: [hi | lo] rows, three weight blocks)
    el = 2 if bf16 else 4
    algorithmic = rows * HN * el * (2 if x3 else 1) * 2 + nlay * HN * kk * el
    roofline = {"bound": "mfma", "kernel": f"{kname} ({shape}, M={rows}" + (f" of the {S} rows" if rows != S else "") + ")", "achieved": round(achieved, 2),
                "algorithmic_bytes": algorithmic,
                "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "launches": len(durs), "avg_launch_ms": round(avg_ms, 4),
                "median_launch_ms": round(statistics.median(durs), 4), "flops_per_launch": flops}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    if x3:
        roofline["layer_algorithmic_tflops"] = round(achieved / 3.0, 2)  # 2 M N K of the fp32 layer / launch time
    return roofline


def hbm_kernels_from_records(recs, n_rays, samples, bf16, _lib, x3=False):
    """HBM-bound kernels of the path: algorithmic bytes / mean launch duration.  The finishers read the
    last layer's partial head sums [S, slots, heads] fp32 for the rows the fused epilogue covered, activation rows else."""
    S = n_rays * samples
    el = 2 if (bf16 and not x3) else 4  # bf16x3 rows are [hi | lo] pairs: 4 bytes per value, like fp32
    lib = _lib.lib()
    enc = [r for r in recs if r["kind"] == _lib.K_ENCODE]
    in_pad = enc[0]["n_pad"] if enc else 64  # row length of the MLP input the encoder writes (58 channels, zero-padded)

    def finish_in_bytes(width, heads):
        # rows whose head products the last layer's epilogue formed: the finisher reads their partial sums [slots][heads] fp32
        # (fp32 kernel: 2 slots per 256 columns; bf16 / bf16x3 rendering forward, store_y = 0: the ring kernel's 2 per 256 -
        # m360_linear_heads_slots_bf16 answers for the kernel the library picks), the other rows' activations
        fused = int(lib.m360_linear_heads_fused_rows(S, width, int(bf16)))
        slots = int(lib.m360_linear_heads_slots_bf16(width, width, 2 if x3 else 1, 0)) if bf16 else int(lib.m360_linear_heads_slots(width, 0))
        return fused * slots * heads * 4 + (S - fused) * width * el

    out = {}
    heads_recs = [r for r in recs if r["kind"] == _lib.K_LINEAR_HEADS and r["n_pad"] == HN]
    if heads_recs:  # one launch over all rows, or one per row block (the shorter last block left out)
        rows = max(r["M"] for r in heads_recs)
        fused_last = [r["ms"] for r in heads_recs if r["M"] == rows]
        ms = sum(fused_last) / len(fused_last)
        out["nerf_last_layer_fused_heads"] = {"avg_launch_ms": round(ms, 4), "rows": rows, "tflops": round(2.0 * rows * HN * HN / ms / 1e9, 1),
                                              "launches": len(fused_last), "note": "MFMA-bound; listed for completeness"}
    for kind, name, nbytes in (
            # feature rows: fp32 [in_pad] and the bf16 mode's [hi | lo] pairs: 4 bytes per value; bf16x3: x6 rows (three bf16 terms, six blocks: 12)
            (_lib.K_ENCODE, "encode_features", S * in_pad * (12 if x3 else 4) + n_rays * (48 + 4 * (samples + 1))),
            (_lib.K_PROP_FINISH, "prop_finish", finish_in_bytes(HP, 1) + n_rays * (4 * (samples + 1) + 12 + 4 * samples + 4 * (samples + 1))),
            (_lib.K_NERF_FINISH, "nerf_finish", finish_in_bytes(HN, 4) + n_rays * (4 * (samples + 1) + 12 + 20))):
        d = [r["ms"] for r in recs if r["kind"] == kind and r["M"] == S]  # full chunks only (a frame ends with a partial one)
        if d:
            ms = sum(d) / len(d)
            out[name] = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes": nbytes,
                         "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                         "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3),
                         # ... and of what a float4 copy reaches on this chip (MI355X_MICROARCH.md: 6.29 TB/s measured, 79 % of the spec)
                         "frac_of_measured_copy_6p29TBps": round(nbytes / (ms * 1e-3) / 1e9 / 6290.0, 3), "launches": len(d)}
            assert out[name]["frac_of_8TBps"] <= 1.0, f"{name}: {out[name]} is above the HBM peak - the byte count is wrong"
    return out


def chain_delta(model, before=None):
    """Counters of the bf16 mode's layer chain (model.chain_status(): launches, launches repaired by the gated re-run, waits that ran
    out, workgroups off their XCD) accumulated since `before`; None outside the bf16 mode.  Informational: outputs are right either way,
    but a run with recoveries > 0 measured the repaired path."""
    if getattr(model, "mlp_dtype", "fp32") != "bf16":
        return None
    now = model.chain_status()
    if before is None:
        return now
    # (the counters live in the stream's scratch buffer and restart at 0 when a larger one replaces it: a delta is never negative)
    d = {k: (now[k] - before.get(k, 0) if now[k] >= before.get(k, 0) else now[k]) for k in ("launches", "recoveries", "timeouts", "xcc_mismatch")}
    d["chain_error"] = d["recoveries"] > 0
    return d


MLP_NAMES = {"fp32": "fp32", "bf16": "bf16 (fp32 accumulate)",
             "bf16x3": "bf16x3 (two bf16 terms per value, xh wh + xl wh + xh wl on the bf16 MFMA, fp32 accumulate)"}
FRAME_POSE = [[1.0, 0.0, 0.0, 0.05], [0.0, 1.0, 0.0, -0.02], [0.0, 0.0, 1.0, 0.1]]


def frame_size(args):
    if not args.frame_size:
        return FRAME_W, FRAME_H
    w, h = (int(v) for v in args.frame_size.lower().split("x"))
    return w, h


def frame_pipeline(model, comm, frames, warmup, overlap, width=FRAME_W, height=FRAME_H, chunks=FRAME_CHUNKS):
    """`frames` frames of width x height rays rendered by all ranks together: per frame every rank generates the rays of
    its own block of chunks (m360_generate_rays_span), renders them into the send block of a PixelGather and the
    pixels are all-gathered.  overlap=True: the all-gather + assembly of frame i run on a side stream under the compute
    of frame i + 1 (two send / receive slots); overlap=False: everything on one stream.
    -> dict with wall seconds (max over ranks) and per-rank medians."""
    import torch

    from mipnerf360_amd.distributed import PixelGather, partition_efficiency, render_local_block
    from mipnerf360_amd.intern.ray import generate_rays
    dev, world = comm.dev, comm.world
    n = width * height
    pg = PixelGather(n, chunks, dev, slots=2)
    pose = torch.tensor(FRAME_POSE, dtype=torch.float32, device=dev)
    focal = 0.9 * width
    outs = [(torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)) for _ in range(2)]
    side = torch.cuda.Stream(device=dev) if (overlap and world > 1) else None
    done = [None, None]
    total = warmup + frames
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(total)]

    def one_frame(i):
        s = i % 2
        cur = torch.cuda.current_stream(dev)
        if side is not None and done[s] is not None:
            cur.wait_event(done[s])  # the gather that read send[s] two frames ago must be through
        ev[i][0].record()
        local = generate_rays(pose, height, width, focal, 0.0, 1.0, True, span=pg.span)
        render_local_block(model, None, chunks, pg, slot=s, local_rays=local)
        ev[i][1].record()
        if world > 1:
            if side is not None:
                side.wait_event(ev[i][1])
                with torch.cuda.stream(side):
                    ev[i][2].record()
                    pg.gather(s)
                    pg.assemble(s, out=outs[s])
                    ev[i][3].record()
                    done[s] = ev[i][3]
            else:
                ev[i][2].record()
                pg.gather(s)
                pg.assemble(s, out=outs[s])
                ev[i][3].record()
        else:
            pg.gather(s)  # one rank: a local copy into the receive block
            pg.assemble(s, out=outs[s])

    for i in range(warmup):
        one_frame(i)
    comm.fence()
    t0 = time.perf_counter()
    for i in range(warmup, total):
        one_frame(i)
    comm.fence()
    elapsed = comm.max(time.perf_counter() - t0)
    compute = [e[0].elapsed_time(e[1]) for e in ev[warmup:]]
    gather = [e[2].elapsed_time(e[3]) for e in ev[warmup:]] if world > 1 else [0.0]
    per_rank = comm.gather_floats([statistics.median(compute), statistics.median(gather), float(pg.span[1] - pg.span[0])])
    last = outs[(total - 1) % 2]
    finite = bool(torch.isfinite(last[0]).all() and torch.isfinite(last[1]).all() and torch.isfinite(last[2]).all())
    # the assembled last frame as one number (sum of the pixels' bit patterns): overlapped == serial == one rank, bit for bit
    bits_sum = int(sum(int(t.contiguous().view(torch.int32).to(torch.int64).sum()) for t in last))
    n_chunks = (n + chunks - 1) // chunks
    return {"frames": frames, "rays_per_frame": n, "seconds": round(elapsed, 4), "seconds_per_frame": round(elapsed / max(frames, 1), 4),
            "rays_per_s": round(n * frames / elapsed, 1), "n_chunks": n_chunks,
            "chunks_per_rank": (n_chunks + world - 1) // world,
            "partition_efficiency_bound": round(partition_efficiency(n, chunks, world), 4),
            "overlap": bool(side is not None), "finite": finite, "frame_bits_sum": bits_sum,
            "per_rank": [{"rank": r, "rays": int(v[2]), "compute_ms_median": round(v[0], 2),
                          "all_gather_ms_median": round(v[1], 3)} for r, v in enumerate(per_rank)]}


def timed_forward(model, rays, steps, warm, _lib, torch):
    """`steps` public forwards under no_grad after `warm` untimed ones -> (ms per step, event records of the timed steps)."""
    with torch.no_grad():
        for _ in range(warm):
            model(rays)
        prof = _lib.Prof(48 * max(steps, 1))
        model.set_prof(prof)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(rays)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    model.set_prof(None)
    recs = prof.records()
    prof.close()
    return ms, recs


def named_workloads(sd_np, dev, _lib):
    """The other single-GPU configurations of BASELINE.json, timed in this process after the headline's timed region, a few
    steps each: the driver's default run then carries a number for every single-GPU config.  Every entry
    has its own dtype, rays/s, ms/step and the dominant kernel's roofline fraction from HIP-event records of its own steps.
    None of this touches the headline fields."""
    import torch

    from mipnerf360_amd import ops, synthetic
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    from mipnerf360_amd.intern.ray import Rays
    from mipnerf360_amd.model import mipNeRF360
    out = {}
    t_all = time.perf_counter()

    def guarded(name, fn, *a):
        """An entry that fails (an out-of-memory on a shared box, ...) reports its error instead of costing the run its headline line."""
        try:
            fn(name, *a)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            out[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
            torch.cuda.empty_cache()

    def forward_entry(name, cfg, mlp_dtype, steps, warm):
        n_rays, samples, _, metric, workload = CONFIGS[cfg]
        m = mipNeRF360(randomized=False, num_samples=samples, hidden_proposal=HP, hidden_nerf=HN, white_bkgd=False, device=dev,
                       mlp_dtype=mlp_dtype)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        m.eval()
        r = synthetic.make_rays("garden", n_rays, seed=1)
        rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])
        chain0 = chain_delta(m)
        ms, recs = timed_forward(m, rays, steps, warm, _lib, torch)
        with torch.no_grad():
            rgb, d, a = m(rays)
        finite = bool(torch.isfinite(rgb).all() and torch.isfinite(d).all() and torch.isfinite(a).all())
        x3 = mlp_dtype == "bf16x3"
        roof = roofline_from_records(recs, n_rays * samples, mlp_dtype != "fp32", cfg, _lib, x3)
        out[name] = {"config": cfg, "workload": workload.format(mlp=MLP_NAMES[mlp_dtype]), "dtype": mlp_dtype if mlp_dtype != "fp32" else "f32",
                     "metric": metric, "rays_per_s": round(n_rays / ms * 1e3, 1), "ms_per_step": round(ms, 3), "steps": steps,
                     "warmup": warm, "finite": finite,
                     "whole_path_tflops": round(n_rays / ms * 1e3 * FLOPS_PER_SAMPLE * samples / 1e12, 1),
                     "roofline": None if roof is None else {k: roof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "traffic", "algorithmic_bytes", "traffic_note") if k in roof}}
        if chain0 is not None:
            out[name]["chain"] = chain_delta(m, chain0)
        out[name]["hbm_kernels"] = hbm_kernels_from_records(recs, n_rays, samples, mlp_dtype != "fp32", _lib, x3)
        del m, rays
        torch.cuda.empty_cache()

    guarded("c5_bf16", forward_entry, "c5", "bf16", 8, 2)        # BASELINE configs[4]'s per-GPU shape: 8192 x 256, bf16 MLP
    guarded("c2_bf16", forward_entry, "c2", "bf16", 20, 3)       # configs[1]'s shape with the opt-in bf16 MLP
    guarded("c2_bf16x3", forward_entry, "c2", "bf16x3", 10, 2)   # ... and with two bf16 terms per value (inside the fp32 tolerance)

    # ---- one iteration of the reference's training loop body (train.py:53-82: two proposal updates, one NeRF update, AdamW), in fp32 and
    # (round 5) with the student in bf16: bf16 tape and gradients in flight, fp32 accumulation, fp32 master weights + AdamW
    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def training_entry(name, mlp_dtype):
        n_rays, samples = CONFIGS["c2"][0], CONFIGS["c2"][1]
        b16 = mlp_dtype == "bf16"
        torch.cuda.reset_peak_memory_stats()
        model = mipNeRF360(randomized=False, num_samples=samples, hidden_proposal=HP, hidden_nerf=HN, device=dev, mlp_dtype=mlp_dtype)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        model.train()
        r = synthetic.make_rays("garden", n_rays, seed=1)
        rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])
        pixels = torch.rand(n_rays, 3, device=dev)
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

        # (bf16: 8 updates each - a timed region starts behind a synchronize, and the host needs ~1 ms to queue the first update's launches)
        prop_ms, nerf_ms = timed(prop_step, 8 if b16 else 1), timed(nerf_step, 8 if b16 else 2)
        it_ms = 2 * prop_ms + nerf_ms
        finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
        del model, opt
        torch.cuda.empty_cache()
        M = n_rays * samples
        flops = 2.0 * M * HN * HN
        if b16:
            dz = torch.randn(M, HN, device=dev).bfloat16()
            x = torch.relu(torch.randn(M, HN, device=dev)).bfloat16()
            wt = ops.pack_linear_bf16_transposed(torch.randn(HN, HN, device=dev) / 32)
            dx = torch.empty(M, HN, device=dev, dtype=torch.bfloat16)
            # dW = dZ^T X with its split reduction - what six of the seven 1024-wide layers run (their bias gradient comes out of the ReLU-mask kernel
            # beside it); the top layer also forms db on the matrix pipe: wgrad_with_bias_ms
            wgrad_ms = timed(lambda: ops.linear_wgrad_bf16(dz, x, want_bias=False), 5)
            wgrad_bias_ms = timed(lambda: ops.linear_wgrad_bf16(dz, x), 5)
            dgrad_ms = timed(lambda: ops.linear_dgrad_bf16(dz, wt, None, out=dx), 5)        # the GEMM alone (MFMA-bound)
            dgrad_serial_ms = timed(lambda: ops.linear_dgrad_bf16(dz, wt, x, out=dx), 5)    # + the ReLU mask pass behind it, one stream, full rate
            peak, wk, dk = PEAK_BF16_MFMA_TFLOPS, "tn16w::linear_tn_bf16_w_kernel (one wave per SIMD, 128 x 128 wave tiles, one read / LDS-DMA piece per MFMA gap) + tn16_reduce_kernel (dW = dZ^T X on v_mfma_f32_16x16x32_bf16, operands transposed by ds_read_b64_tr_b16; with_bias: + db on the matrix pipe and its reduction)", \
                "linear_bf16_w16_kernel on the transposed bf16 packing (dX = dZ W; the ReLU mask is its own HBM-bound kernel: relu_mask_1024)"
        else:
            dz = torch.randn(M, HN, device=dev)
            x = torch.relu(torch.randn(M, HN, device=dev))
            wt = ops.pack_linear_transposed(torch.randn(HN, HN, device=dev) / 32)
            dx = torch.empty(M, HN, device=dev)
            wgrad_ms = timed(lambda: ops.linear_wgrad(dz, x), 3)
            dgrad_ms = timed(lambda: ops.linear_dgrad(dz, wt, x, out=dx), 3)
            peak, wk, dk = PEAK_F32_MFMA_TFLOPS, "linear_tn_kernel + tn_reduce_kernel (dW = dZ^T X, bias gradient fused)", "linear_f32_mfma_persist_kernel<RELU_MASK> (dX = (dZ W) * [a > 0])"
        del dz, x, dx, wt
        torch.cuda.empty_cache()
        tg = training_gemm_traffic("bf16" if b16 else "fp32")
        el_ = 2 if b16 else 4

        def with_traffic(entry, key, algorithmic):
            entry["algorithmic_bytes"] = algorithmic
            if key in tg:
                entry["traffic"] = tg[key]["bytes_per_launch"]
                entry["traffic_note"] = "fabric-side counter bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950 corrections; Infinity-Cache hits included)"
            else:
                entry["traffic"] = None
            return entry
        out[name] = {
            "config": "c2", "dtype": "bf16" if b16 else "f32", "finite": finite,
            "workload": f"one iteration of the reference's training loop body (train.py:53-82) at 4096 rays x 128 samples, full width, {MLP_NAMES[mlp_dtype]}: "
                        "two proposal updates (forward both nets, Loss_prop, backward of the proposal net, AdamW) + one NeRF update "
                        "(forward both nets, Loss_nerf + 0.01 Loss_dist, backward of the NeRF net, AdamW); tape-keeping forwards, "
                        "hand-written backward (m360_prop_backward / m360_nerf_backward), torch.optim.AdamW on fp32 master weights",
            "iteration_ms": round(it_ms, 2), "prop_update_ms": round(prop_ms, 2), "nerf_update_ms": round(nerf_ms, 2),
            "train_rays_per_s": round(n_rays / it_ms * 1e3, 1),
            "nerf_update_tflops": round(M * (423424 + 3 * 14807040) / nerf_ms / 1e9, 1),
            # algorithmic bytes: dW reads dZ and X once and writes [1024, 1024] fp32 (+ its split partials, which are the kernel's own business);
            # dX reads dZ and the 1024 x 1024 weights and writes dX (fp32: + the stored activation for the fused ReLU mask)
            "wgrad_1024x1024": with_traffic({"ms": round(wgrad_ms, 3), "tflops": round(flops / wgrad_ms / 1e9, 1), "frac": round(flops / wgrad_ms / 1e9 / peak, 4), "kernel": wk,
                                             **({"with_bias_ms": round(wgrad_bias_ms, 3), "with_bias_frac": round(flops / wgrad_bias_ms / 1e9 / peak, 4)} if b16 else {})},
                                            "wgrad", 2 * M * HN * el_ + HN * HN * 4),
            "dgrad_1024x1024": with_traffic({"ms": round(dgrad_ms, 3), "tflops": round(flops / dgrad_ms / 1e9, 1), "frac": round(flops / dgrad_ms / 1e9 / peak, 4), "kernel": dk},
                                            "dgrad", (2 if b16 else 3) * M * HN * el_ + HN * HN * el_),
            **({"relu_mask_1024": {
                # dX *= [a > 0]: its own kernel with its own bound (HBM: dX in, the stored activation in, dX out = 6 bytes per element).  Timed as
                # (GEMM + mask on one stream) - (GEMM alone): the mask at full rate.  In the training iteration it runs THROTTLED on a second stream
                # beside the weight gradient (m360_hyper_t.side), where the pair costs ~0.12 ms more than the weight gradient alone
                "ms": round(dgrad_serial_ms - dgrad_ms, 3), "bound": "hbm", "algorithmic_bytes": 3 * M * HN * 2,
                "achieved_GBps": round(3 * M * HN * 2 / max(dgrad_serial_ms - dgrad_ms, 1e-6) / 1e6, 1),
                "frac_of_8TBps": round(3 * M * HN * 2 / max(dgrad_serial_ms - dgrad_ms, 1e-6) / 1e6 / PEAK_HBM_GBPS, 3),
                "dgrad_plus_mask_serial_ms": round(dgrad_serial_ms, 3), "traffic": (tg.get("relu_mask") or {}).get("bytes_per_launch"),
                "kernel": "relu_mask_bf16_kernel (16-byte pieces, whole rows per workgroup)"}} if b16 else {}),
            "peak": peak, "unit": "TFLOP/s", "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}

    guarded("c2_training_iteration", training_entry, "fp32")
    guarded("c2_training_iteration_bf16", training_entry, "bf16")

    # ---- a frame at the reference's CLI default chunk size (config.py:49: chunks = 128; configs[0]'s scene: lego 400 x 400, 64 samples per ray,
    # white background): 1250 chunks, each contracted by ITS OWN norm, launched 32 at a time (m360_hyper_t.norm_group_rays) - bit-identical to one
    # launch sequence per chunk (tests), here with a number
    def chunks128_entry(name):
        from mipnerf360_amd.intern.ray import generate_rays
        fm = mipNeRF360(randomized=False, num_samples=64, hidden_proposal=HP, hidden_nerf=HN, white_bkgd=True, device=dev)
        fm.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        fm.eval()
        pose = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 4.0]], device=dev)
        frays = generate_rays(pose, 400, 400, 555.6, 2.0, 6.0, False)
        fm.render_rays(Rays(*[f[:8192] for f in frays]), chunks=128)  # warm-up: packing, workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rgb, dist, acc = fm.render_rays(frays, chunks=128)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        flops_ray = FLOPS_PER_SAMPLE * 64
        out[name] = {
            "config": "c1", "dtype": "f32", "workload": "nerf_synthetic/lego-like 400 x 400 frame (160 000 rays, synthetic pinhole pose, near 2 / far 6, white background), 64 samples per ray, "
                                                        "rendered in the reference's default chunks of 128 rays (config.py:49) = 1250 chunks of their own contraction norm, 32 chunks per launch "
                                                        "sequence; proposal 4x256 + NeRF 8x1024 MLPs in fp32 on MFMA, random-init Kaiming weights (BASELINE.json configs[0]'s scene on the GPU)",
            "rays_per_s": round(160000 / sec, 1), "ms_per_step": round(sec * 1e3, 1), "steps": 1, "chunks": 128, "n_chunks": 1250,
            "finite": bool(torch.isfinite(rgb).all() and torch.isfinite(dist).all() and torch.isfinite(acc).all()),
            "whole_path_tflops": round(160000 / sec * flops_ray / 1e12, 1), "frac_of_fp32_mfma_peak": round(160000 / sec * flops_ray / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        del fm, frays, rgb, dist, acc
        torch.cuda.empty_cache()

    guarded("c1_frame_chunks128", chunks128_entry)
    out["seconds"] = round(time.perf_counter() - t_all, 2)
    return out


def worker(args):
    n_rays, samples, mlp_default, metric, workload = CONFIGS[args.config]
    mlp_dtype = args.mlp_dtype or mlp_default
    frame_cfg = args.config == "c4"
    steps = args.steps if args.steps is not None else (3 if frame_cfg else 50)
    warmup = args.warmup if args.warmup is not None else (1 if frame_cfg else 5)
    if args.cpu_rays is None:
        args.cpu_rays = n_rays if (args.config == "c2" and mlp_dtype != "bf16") else 0
    if args.frame_steps is None:
        args.frame_steps = 1 if (args.config == "c2" and mlp_dtype == "fp32") else 0

    import numpy as np
    import torch

    from mipnerf360_amd import _lib, synthetic
    from mipnerf360_amd.intern.ray import Rays
    from mipnerf360_amd.model import mipNeRF360

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if args.plain_rows or args.no_chain:  # per-call bits of m360_hyper_t.tuning; the host mirror keeps them for this thread
        from mipnerf360_amd import ops as _ops
        if args.plain_rows:
            _ops.set_paired_rows(False)
        if args.no_chain:
            _ops.set_hidden_chain(False)
    ndev = torch.cuda.device_count()
    # one GPU per rank.  With fewer devices than ranks the ranks wrap around: RCCL then refuses the duplicate device
    # with its own error (the gloo diagnostics backend lets ranks share a GPU).
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    comm = Comm(world, rank, dev, args.backend)

    sd_np = synthetic.make_state_dict(HP, HN, seed=0)
    bf16 = mlp_dtype in ("bf16", "bf16x3")  # the bf16 matrix pipe (roofline peak); bf16x3 keeps 16 significant bits
    x3 = mlp_dtype == "bf16x3"
    model = mipNeRF360(randomized=False, num_samples=samples, hidden_proposal=HP, hidden_nerf=HN, white_bkgd=False,
                       device=dev, mlp_dtype=mlp_dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.eval()  # rendering: weights are packed once, outside the timed region
    if world > 1:
        from mipnerf360_amd.distributed import check_replicas
        check_replicas(model)  # one 32-byte all-reduce at setup: every rank holds the same weights / sample counts

    chain0 = chain_delta(model)
    line = {"metric": metric, "value": None, "unit": "rays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if frame_cfg else "weak",
            "vs_baseline": None, "dtype": mlp_dtype if bf16 else "f32", "data": "synthetic"}
    S = n_rays * samples
    prof = None

    if frame_cfg:
        # ---------------------------------------------------------------- c4: frames sharded over the ranks
        fw, fh = frame_size(args)
        n_frame = fw * fh
        chunks_mine = (n_frame + FRAME_CHUNKS - 1) // FRAME_CHUNKS
        prof = _lib.Prof(40 * (chunks_mine // world + 2) * (steps + warmup) * 2 + 64)
        model.set_prof(prof)
        res = frame_pipeline(model, comm, steps, warmup, overlap=True, width=fw, height=fh)
        model.set_prof(None)
        recs = prof.records()
        serial = frame_pipeline(model, comm, steps, warmup, overlap=False, width=fw, height=fh) if world > 1 else None
        if args.frame_size:
            workload = f"DIAGNOSTIC frame size {fw}x{fh} instead of 1237x822; " + workload
        elapsed = res["seconds"]
        line.update(value=res["rays_per_s"], ms_per_step=round(1e3 * elapsed / max(steps, 1), 3))
        line["config"] = {"workload": workload.format(mlp=MLP_NAMES[mlp_dtype]), "name": "c4",
                          "rays_per_frame": n_frame, "samples_per_ray": samples, "chunks": FRAME_CHUNKS,
                          "n_chunks": res["n_chunks"], "chunks_per_rank": res["chunks_per_rank"],
                          "partition_efficiency_bound": res["partition_efficiency_bound"],
                          "parallelism": f"whole chunks sharded over {world} GPU(s), replicated weights" +
                                         (", one RCCL all-gather of the pixel block per frame overlapped with the next "
                                          "frame's compute" if world > 1 else ""),
                          "flops_per_ray": FLOPS_PER_SAMPLE * samples,
                          "whole_path_tflops": round(res["rays_per_s"] * FLOPS_PER_SAMPLE * samples / 1e12, 2)}
        line["frame"] = {"overlapped": res, "serial": serial}
    else:
        # ---------------------------------------------------------------- c2 / c5: one batch per rank and step
        rays_np = synthetic.make_rays("garden", n_rays, seed=1 + rank)
        rays = Rays(*[torch.from_numpy(rays_np[k]).to(dev) for k in synthetic.RAY_FIELDS])
        pg = None
        if world > 1:
            from mipnerf360_amd.distributed import PixelGather
            pg = PixelGather(world * n_rays, n_rays, dev)  # one chunk per rank: preallocated send / receive blocks
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

        def step(i=None):
            if i is not None:
                ev[i][0].record()
            with torch.no_grad():  # rendering, as in render_image (model.py:261); with grad enabled the mirrors keep a training tape
                rgb, d, a = model(rays)  # the public forward: (rgb[B,3], distance[B], acc[B])
            if i is not None:
                ev[i][1].record()
            if pg is not None:  # the path's one exchange step: 20 B per ray
                for dst, src in zip(pg.local_outputs(), (rgb, d, a)):
                    dst.copy_(src)
                pg.gather()
            if i is not None:
                ev[i][2].record()
            return rgb, d, a

        for _ in range(warmup):
            step()
        prof = _lib.Prof(48 * max(steps, 1))  # caller-owned HIP-event recorder: every kernel of the stage drivers
        model.set_prof(prof)
        comm.fence()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = step(i)
        comm.fence()
        elapsed = time.perf_counter() - t0
        model.set_prof(None)
        compute_ms = [e[0].elapsed_time(e[1]) for e in ev]
        gather_ms = [e[1].elapsed_time(e[2]) for e in ev]
        step_ms = [e[0].elapsed_time(e[2]) for e in ev]
        elapsed = comm.max(elapsed)
        per_rank = None
        if world > 1:
            allr = comm.gather_floats([statistics.median(compute_ms), statistics.median(gather_ms)])
            per_rank = [{"rank": r, "compute_ms_median": round(t[0], 3), "all_gather_ms_median": round(t[1], 3)}
                        for r, t in enumerate(allr)]
        recs = prof.records()
        value = world * n_rays * steps / elapsed
        med_ms = statistics.median(step_ms) if step_ms else 0.0
        line.update(value=round(value, 1), ms_per_step=round(1e3 * elapsed / max(steps, 1), 3),
                    ms_per_step_median=round(med_ms, 3),
                    value_at_median_step=round(world * n_rays / (med_ms * 1e-3), 1) if med_ms else None)
        line["config"] = {"workload": workload.format(mlp=MLP_NAMES[mlp_dtype]),
                          "name": args.config, "rays_per_gpu": n_rays, "samples_per_ray": samples,
                          "parallelism": f"rays sharded over {world} GPU(s), replicated weights" +
                                         (f", RCCL all-gather of the [{n_rays},5] pixel block per step" if world > 1 else ""),
                          "flops_per_ray": FLOPS_PER_SAMPLE * samples,
                          "whole_path_tflops": round(value * FLOPS_PER_SAMPLE * samples / 1e12, 2)}
        if args.plain_rows:
            line["config"]["plain_rows"] = True
        if args.no_chain:
            line["config"]["no_chain"] = True
        if per_rank:
            line["per_rank"] = per_rank

    # ---- per-kernel numbers from the event records (HIP events on the launch stream, inside the timed region)
    prof.close()
    line["roofline"] = roofline_from_records(recs, S, bf16, args.config, _lib, x3)
    if chain0 is not None:  # bf16 mode: what the layer chain's self-checks saw over warm-up + timed region (this rank)
        line["chain"] = chain_delta(model, chain0)
    line["hbm_kernels"] = hbm_kernels_from_records(recs, n_rays, samples, bf16, _lib, x3)
    line["rccl"] = comm.info

    if not frame_cfg and args.frame_steps > 0:
        # BASELINE configs[2] as written ("hierarchical 64+128 samples") beside the headline, and configs[3] for N > 1: the
        # same processes render one frame together (strong scaling).  64 proposal + 128 NeRF samples per ray is the build's
        # `num_samples_fine` extension (the reference draws as many NeRF as proposal samples, intern/ray.py:147).
        # An extra leg: the headline above is complete.  What CAN fail on one rank alone - building the frame model, its buffers, the event
        # recorder (out of memory) - is tried first, and the ranks AGREE on the outcome (one MIN all-reduce that every rank takes part in,
        # whatever happened to it): all render or none does, nobody is left waiting in a collective for a rank that gave up.
        # From there on an exception with N > 1 is not caught: it ends this process, the launcher ends the others, the run exits non-zero.
        fmodel = fprof = None
        setup_error = None
        try:
            fw, fh = frame_size(args)
            if rank == args.fail_frame_setup_on_rank:
                raise MemoryError("injected by --fail-frame-setup-on-rank (test hook)")
            fmodel = mipNeRF360(randomized=False, num_samples=64, num_samples_fine=128, hidden_proposal=HP, hidden_nerf=HN,
                                white_bkgd=False, device=dev, mlp_dtype=mlp_dtype)
            fmodel.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
            fmodel.eval()
            fprof = _lib.Prof(40 * ((fw * fh + FRAME_CHUNKS - 1) // FRAME_CHUNKS // world + 2) * args.frame_steps + 64)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            setup_error = f"{type(e).__name__}: {e}"[:500]
        all_ok = comm.min(0.0 if setup_error else 1.0) > 0.5

        def frame_leg():
            fmodel.set_prof(fprof)
            fr = frame_pipeline(fmodel, comm, args.frame_steps, 0, overlap=False, width=fw, height=fh)
            fmodel.set_prof(None)
            frecs = fprof.records()
            fprof.close()
            fr["samples_per_ray"] = "64 proposal + 128 NeRF (BASELINE configs[2]: 'hierarchical 64+128 samples')"
            fr["flops_per_ray"] = 423424 * 64 + 14807040 * 128
            fr["whole_path_tflops"] = round(fr["rays_per_s"] * fr["flops_per_ray"] / 1e12, 2)
            froof = roofline_from_records(frecs, FRAME_CHUNKS * 128, bf16, "c2", _lib, x3)
            fr["roofline"] = None if froof is None else {k: froof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches")}
            fr["workload"] = (f"DIAGNOSTIC frame size {fw}x{fh} instead of 1237x822; " if args.frame_size else "") + \
                CONFIGS["c4"][4].format(mlp=MLP_NAMES[mlp_dtype]).replace("x 128 samples/ray", "x (64 proposal + 128 NeRF) samples/ray")
            return fr

        if not all_ok:
            line["strong_scaling_frame"] = {"error": setup_error or "another rank could not set the frame leg up (see its stderr)"}
        elif world > 1:
            line["strong_scaling_frame"] = frame_leg()  # collectives inside: a failure ends the run (non-zero exit), it is not papered over
        else:
            try:
                line["strong_scaling_frame"] = frame_leg()
            except Exception as e:  # noqa: BLE001  (one rank: nobody waits for us)
                import traceback
                traceback.print_exc()
                line["strong_scaling_frame"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        del fmodel, fprof
        torch.cuda.empty_cache()

    if rank != 0:
        comm.close()
        return

    if args.config == "c2" and mlp_dtype == "fp32" and world == 1 and not args.no_named:
        line["named_workloads"] = named_workloads(sd_np, dev, _lib)
        if "strong_scaling_frame" in line and "error" not in line["strong_scaling_frame"]:  # BASELINE configs[2] is the frame leg above: listed with the others
            fr = line["strong_scaling_frame"]
            line["named_workloads"]["c3_frame_64+128"] = {
                "config": "c3", "dtype": "f32", "workload": fr["workload"], "rays_per_s": fr["rays_per_s"],
                "ms_per_step": round(1e3 * fr["seconds_per_frame"], 1), "steps": fr["frames"], "roofline": fr["roofline"],
                "whole_path_tflops": fr["whole_path_tflops"], "see": "strong_scaling_frame"}

    if not frame_cfg and world == 1 and args.cpu_rays > 0:
        n_cpu = min(args.cpu_rays, n_rays)
        cpu_rps, cpu_times, o = cpu_baseline(sd_np, rays_np, n_cpu, samples)
        line["cpu_baseline"] = {"value": round(cpu_rps, 2), "unit": "rays/s", "cores": torch.get_num_threads(),
                                "kind": "port",
                                "sample": f"{'all' if n_cpu == n_rays else 'first'} {n_cpu} rays of the same {n_rays}x{samples} batch "
                                          f"as one chunk, {len(cpu_times)} pass(es) ({', '.join(f'{t:.1f}' for t in cpu_times)} s) "
                                          f"(oracle/ref_path.py, vectorised torch-CPU fp32, closed-form Jacobian; the "
                                          f"unmodified reference measured 17.9-19.1 rays/s on 8 cores, BASELINE.md)",
                                "host_cpus": os.cpu_count()}
        # parity + PSNR: with the whole batch on the CPU the timed forward's own output is compared (same chunk, same
        # contraction norm); a sub-batch is re-rendered as its own chunk on the GPU
        if n_cpu == n_rays:
            g_rgb, g_dist, g_acc = out
        else:
            sub = Rays(*[f[:n_cpu].contiguous() for f in rays])
            with torch.no_grad():
                g_rgb, g_dist, g_acc = model(sub)
        mse = float(((g_rgb.cpu() - o[0]) ** 2).mean())
        line["parity"] = {"max_abs_rgb": float((g_rgb.cpu() - o[0]).abs().max()),
                          "max_abs_acc": float((g_acc.cpu() - o[2]).abs().max()),
                          "max_abs_dist": float((g_dist.cpu() - o[1]).abs().max()),
                          "psnr_vs_cpu_db": round(-10.0 * float(np.log10(max(mse, 1e-20))), 2), "rays": n_cpu,
                          "note": "HIP forward of the timed region vs the CPU oracle on the same rays as ONE chunk"}
    print(json.dumps(line), flush=True)
    comm.close()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args, argv))
    if args.dry_launch:
        print(json.dumps({"launch": None, "ranks": 1, "note": "single process: nothing to launch"}), flush=True)
        return
    worker(args)


if __name__ == "__main__":
    main()
