#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of a profiling run (gpurun_out/<tag>/{stats,pmc_fetch,pmc_write,pmc_sq}) into the
small summaries committed under profiles/<tag>/ and into profiles/traffic.json (read by bench.py's roofline).

The raw run is produced on the GPU box by:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>/stats -- python3 bench.py --steps 10 --warmup 3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -d gpurun_out/<tag>/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-rays 0
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  ... -d gpurun_out/<tag>/pmc_write -- (same)
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ... -d gpurun_out/<tag>/pmc_sq -- (same)
PMC passes are separate runs, never combined with tracing domains other than --kernel-trace.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402  (sha256 over the dominant kernel's source files)
M_BENCH, N_BENCH = 4096 * 128, 1024


def one(pattern):
    """the NEWEST match: gpurun merges every call's files into gpurun_out/<tag>, so older runs of the same step lie next to it"""
    hits = glob.glob(pattern)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return max(hits, key=os.path.getmtime)


def durations(d):
    return {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in
            csv.DictReader(open(one(f"{d}/*/*_kernel_trace.csv")))}


def dominant_counters(d):
    """mean counter value over the dominant launches (ReLU 1024x1024 layers on M=524288: half-tile kernel, > 6 ms)."""
    dur = durations(d)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f"{d}/*/*_counter_collection.csv"))):
        if "linear_f32_hd_kernel<1" in r["Kernel_Name"] and dur[r["Dispatch_Id"]] > 6e6:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    return {k: (sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v)) for k, v in agg.items()}


def counter_summary(d, out):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(one(f"{d}/*/*_counter_collection.csv"))):
        k = (r["Kernel_Name"].split("(")[0][-70:], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,mean_value\n")
        for (k, c), (n, s) in sorted(agg.items()):
            f.write(f"\"{k}\",{c},{n},{s / n:.6e}\n")


def step_trace(d, out):
    """per-launch durations of the LAST bench step (kernel order as launched)."""
    rows = sorted(csv.DictReader(open(one(f"{d}/*/*_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # a forward begins with the proposal stage's prologue (round 4; sample_t_kernel before) and ends with the NeRF finisher
    # (t_to_s_kernel before round 4)
    first, last_k = ("stage_prologue_kernel", "nerf_finish_kernel") if any("stage_prologue_kernel" in n for n in names) else ("sample_t_kernel", "t_to_s_kernel")
    starts = [i for i, n in enumerate(names) if first in n and any(last_k in m for m in names[i:])]
    ends = [min(j for j, n in enumerate(names) if last_k in n and j > i) for i in starts]
    # rendering forwards only (18 launches since round 4, 26 before): the named workloads of the same process also train
    keep = [k for k, (s_, e_) in enumerate(zip(starts, ends)) if e_ - s_ + 1 <= 26 and any("linear_f32" in n for n in names[s_:e_ + 1])]
    starts, ends = [starts[k] for k in keep], [ends[k] for k in keep]
    spans = [int(rows[e]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"]) for s, e in zip(starts, ends)]
    med = sorted(spans)[len(spans) // 2]
    full = [k for k, sp in enumerate(spans) if 0.9 * med < sp < 1.1 * med]  # the 4096-ray bench steps (not a frame's partial chunk, not c5)
    last, end = starts[full[-1]], ends[full[-1]]
    with open(out, "w") as f:
        f.write("order,kernel,duration_us\n")
        total = 0
        for i, r in enumerate(rows[last:end + 1]):
            dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            total += dt
            short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("m360::", "")[:60]
            f.write(f"{i},{short},{dt:.1f}\n")
        span = (int(rows[end]["End_Timestamp"]) - int(rows[last]["Start_Timestamp"])) / 1e3
        f.write(f"sum_of_kernels,,{total:.1f}\nwall_span_first_to_last,,{span:.1f}\n")
    return open(out).read()


def _bench_line(path):
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def markdown(tag):
    """The measured numbers of profiles/<tag>/ as the markdown block DESIGN.md carries between its
    `<!-- profiles:begin -->` / `<!-- profiles:end -->` markers (tests/test_host_cpu.py checks they are identical: numbers
    in the docs are regenerated from profiles/, never remembered)."""
    d = os.path.join(ROOT, "profiles", tag)
    out = [f"*Generated by `python tools/summarize_profiles.py --markdown {tag}` from `profiles/{tag}/` - do not edit by hand.*", ""]
    b = _bench_line(os.path.join(d, "bench.json"))
    if b:
        r = b["roofline"]
        out += [f"**Headline (`profiles/{tag}/bench.json`, 1 x MI355X, {b['config']['name']}, dtype {b['dtype']}):** "
                f"{b['value']:.0f} rays/s, {b['ms_per_step']:.3f} ms/step (median {b.get('ms_per_step_median', 0):.3f}) = "
                f"{b['config']['whole_path_tflops']:.2f} TFLOP/s whole path = "
                f"{b['config']['whole_path_tflops'] / r['peak']:.3f} of the {r['peak']:.1f} TF fp32-MFMA roofline; dominant kernel "
                f"`{r['kernel']}`: {r['avg_launch_ms']:.4f} ms mean over {r['launches']} launches = {r['achieved']:.1f} TF = "
                f"**{r['frac']:.4f}** of peak (HIP events on the launch stream, inside the timed region).", ""]
        if b.get("parity"):
            p_ = b["parity"]
            out += [f"Parity in the same run (all {p_['rays']} rays as ONE chunk against the CPU oracle): max |d rgb| {p_['max_abs_rgb']:.2e}, "
                    f"|d acc| {p_['max_abs_acc']:.2e}, |d distance| {p_.get('max_abs_dist', float('nan')):.2e}, PSNR {p_['psnr_vs_cpu_db']} dB.  "
                    f"CPU baseline ({b['cpu_baseline']['kind']}): {b['cpu_baseline']['value']} rays/s on {b['cpu_baseline']['cores']} threads.", ""]
        if b.get("strong_scaling_frame"):
            f_ = b["strong_scaling_frame"]
            out += [f"Frame leg of the same run (BASELINE configs[2]/[3] at N = {b['n_gpus']}{', ' + f_['samples_per_ray'] if f_.get('samples_per_ray') else ''}): "
                    f"{f_['rays_per_frame']} rays in {f_['n_chunks']} chunks, "
                    f"{f_['seconds_per_frame']:.3f} s per frame = {f_['rays_per_s']:.0f} rays/s.", ""]
        if b.get("hbm_kernels"):
            hk = b["hbm_kernels"]
            out += ["HBM-bound kernels of the same run (algorithmic bytes / mean launch duration, HIP events): " +
                    "; ".join(f"`{k}` {v['avg_launch_ms'] * 1e3:.1f} us = {v['achieved_GBps']:.0f} GB/s = {v['frac_of_8TBps']:.3f} of 8 TB/s"
                              for k, v in hk.items() if "frac_of_8TBps" in v) + ".", ""]
        if b.get("named_workloads"):
            nw = b["named_workloads"]
            out += [f"`named_workloads` of the same run ({nw.get('seconds', '?')} s in total, a few steps each):", ""]
            for k, v in nw.items():
                if not isinstance(v, dict):
                    continue
                if "iteration_ms" in v:
                    out.append(f"* `{k}` (train.py:53-82 at 4096 x 128, {v.get('dtype', 'f32')}): **{v['iteration_ms']:.1f} ms per iteration = {v['train_rays_per_s']:.0f} rays/s** "
                               f"(proposal update {v['prop_update_ms']:.1f} ms x 2, NeRF update {v['nerf_update_ms']:.1f} ms = {v['nerf_update_tflops']:.1f} TF sustained); "
                               f"wgrad 1024x1024 {v['wgrad_1024x1024']['ms']:.3f} ms = {v['wgrad_1024x1024']['tflops']:.1f} TF = {v['wgrad_1024x1024']['frac']:.3f}"
                               + (f" (with the bias gradient, top layer only: {v['wgrad_1024x1024']['with_bias_ms']:.3f} ms = {v['wgrad_1024x1024']['with_bias_frac']:.3f})" if "with_bias_ms" in v["wgrad_1024x1024"] else "") + ", "
                               
                               f"dgrad {v['dgrad_1024x1024']['ms']:.3f} ms = {v['dgrad_1024x1024']['tflops']:.1f} TF = {v['dgrad_1024x1024']['frac']:.3f} of {v['peak']:.1f}"
                               + (f" (the GEMM; its ReLU mask is a kernel of its own: {v['relu_mask_1024']['ms']:.3f} ms = {v['relu_mask_1024']['achieved_GBps']:.0f} GB/s = "
                                  f"{v['relu_mask_1024']['frac_of_8TBps']:.3f} of 8 TB/s at full rate, GEMM + mask on one stream {v['relu_mask_1024']['dgrad_plus_mask_serial_ms']:.3f} ms; "
                                  f"in the iteration it runs throttled beside the weight gradient)" if v.get("relu_mask_1024") else "")
                               + (f"; counter traffic: wgrad {v['wgrad_1024x1024']['traffic'] / 1e9:.2f} GB against {v['wgrad_1024x1024']['algorithmic_bytes'] / 1e9:.2f} GB algorithmic, "
                                  f"dgrad {v['dgrad_1024x1024']['traffic'] / 1e9:.2f} against {v['dgrad_1024x1024']['algorithmic_bytes'] / 1e9:.2f}"
                                  if v["wgrad_1024x1024"].get("traffic") and v["dgrad_1024x1024"].get("traffic") else "")
                               + f"; peak memory {v['peak_mem_gb']:.1f} GB")
                else:
                    r_ = v.get("roofline") or {}
                    line = f"* `{k}` (dtype {v['dtype']}): **{v['rays_per_s']:.0f} rays/s, {v['ms_per_step']:.3f} ms/step**"
                    if r_:
                        line += f", dominant kernel {r_.get('achieved', 0):.0f} TF = {r_.get('frac', 0):.3f} of {r_.get('peak', 0):.0f}"
                        if r_.get("traffic"):
                            line += f", counter traffic {r_['traffic'] / 1e9:.2f} GB per launch against {r_.get('algorithmic_bytes', 0) / 1e9:.2f} GB algorithmic"
                    elif "frac_of_fp32_mfma_peak" in v:
                        line += f" ({v.get('n_chunks', '?')} chunks of {v.get('chunks', '?')} rays), whole path {v['whole_path_tflops']:.1f} TF = {v['frac_of_fp32_mfma_peak']:.3f} of the fp32-MFMA roofline"
                    ch = v.get("chain")
                    if ch:
                        line += f"; layer chain: {ch['launches']} launches, {ch['recoveries']} repaired, {ch['timeouts']} waits ran out, {ch['xcc_mismatch']} workgroups off their XCD"
                    hkn = v.get("hbm_kernels") or {}
                    if any("frac_of_8TBps" in x for x in hkn.values()):
                        line += "; HBM-bound kernels: " + ", ".join(f"`{k2}` {x['avg_launch_ms'] * 1e3:.1f} us = {x['frac_of_8TBps']:.3f} of 8 TB/s" for k2, x in hkn.items() if "frac_of_8TBps" in x)
                    out.append(line)
            out.append("")
    dom = os.path.join(d, "dominant_kernel_launches.csv")
    if os.path.exists(dom):
        row = list(csv.DictReader(open(dom)))[0]
        out += [f"rocprofv3 `--kernel-trace --stats` of the same command (`rocprofv3_kernel_stats_bench.csv`, `dominant_kernel_launches.csv`): "
                f"{row['kernel']} {row['shape']}: {int(row['launches'])} launches, mean {float(row['mean_ns']) / 1e6:.4f} ms "
                f"(min {float(row['min_ns']) / 1e6:.4f}, max {float(row['max_ns']) / 1e6:.4f}) = {row['tflops_at_mean']} TF.", ""]
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        if tag in t.get("source", ""):
            out += [f"Counter traffic of the dominant kernel (`profiles/traffic.json`, separate `--pmc` passes, gfx950 corrections): fetch "
                    f"{t['fetch_bytes'] / 1e9:.2f} GB + write {t['write_bytes'] / 1e9:.2f} GB = {t['linear_f32_mfma_1024x1024_bytes_per_launch'] / 1e9:.2f} GB per launch "
                    f"against {t['algorithmic_bytes'] / 1e9:.2f} GB algorithmic ({t['linear_f32_mfma_1024x1024_bytes_per_launch'] / t['algorithmic_bytes']:.2f}x: "
                    f"fabric-side counters, Infinity-Cache hits included - W is re-streamed per 128-row tile); MFMA pipe busy "
                    f"{100 * t['mfma_busy_fraction']:.1f} % at {t['effective_clock_ghz']:.3f} GHz.", ""]
            for key, b in (t.get("by_workload") or ({"c2_bf16": t["bf16_ring_kernel"]} if t.get("bf16_ring_kernel") else {})).items():
                out += [f"The same for the dominant kernel of `{key}` (`{b['kernel']}`, `rocprofv3_pmc_{key}_summary.json`): fetch "
                        f"{b['fetch_bytes'] / 1e9:.2f} GB + write {b['write_bytes'] / 1e9:.2f} GB = {b['bytes_per_launch'] / 1e9:.2f} GB per launch against "
                        f"{b['algorithmic_bytes'] / 1e9:.2f} GB algorithmic ({b['bytes_per_launch'] / b['algorithmic_bytes']:.2f}x); MFMA pipe busy "
                        f"{100 * b['mfma_busy_fraction']:.1f} % at {b['effective_clock_ghz']:.3f} GHz; LDS bank-conflict cycles "
                        f"{b['lds_bank_conflict_cycles']:.0f} over {b['lds_instructions'] / 1e6:.1f} M LDS instructions.", ""]
    tr = os.path.join(d, "last_step_kernel_trace.csv")
    if os.path.exists(tr):
        rows = []
        for ln in open(tr).read().splitlines()[1:]:  # kernel names hold commas (template arguments): split from both ends
            order, rest = ln.split(",", 1)
            kernel, dur = rest.rsplit(",", 1)
            rows.append({"order": order, "kernel": kernel.strip('"'), "duration_us": dur})
        out += ["Per-launch durations of the last step under rocprofv3 (`last_step_kernel_trace.csv`):", "",
                "| # | kernel | us | | # | kernel | us |", "|---|---|---|---|---|---|---|"]
        body = [r_ for r_ in rows if r_["order"].isdigit()]
        half = (len(body) + 1) // 2
        for i in range(half):
            a = body[i]
            c = body[i + half] if i + half < len(body) else None
            out.append(f"| {a['order']} | `{a['kernel'][:52]}` | {a['duration_us']} | | " +
                       (f"{c['order']} | `{c['kernel'][:52]}` | {c['duration_us']} |" if c else " | | |"))
        tail = {r_["order"]: r_["duration_us"] for r_ in rows if not r_["order"].isdigit()}
        out += ["", f"Sum of kernels {tail.get('sum_of_kernels', '?')} us, wall span first-to-last {tail.get('wall_span_first_to_last', '?')} us.", ""]
    for name, label in (("bench_bf16.json", "`--mlp-dtype bf16` (configs[1] shape, opt-in bf16 MLP)"), ("bench_c5.json", "`--config c5` (configs[4] per-GPU shape, bf16)"),
                        ("bench_bf16x3.json", "`--mlp-dtype bf16x3` (configs[1] shape, two bf16 terms per value)"), ("bench_c5_bf16x3.json", "`--config c5 --mlp-dtype bf16x3` (configs[4] per-GPU shape, two bf16 terms per value)"),
                        ("bench_c4.json", "`--config c4` (1237 x 822 frames)"),
                        ("bench_c4_bf16.json", "`--config c4 --mlp-dtype bf16` (1237 x 822 frames, opt-in bf16 MLP)"),
                        ("bench_c4_bf16x3.json", "`--config c4 --mlp-dtype bf16x3` (1237 x 822 frames, two bf16 terms per value)")):
        x = _bench_line(os.path.join(d, name))
        if x:
            r = x.get("roofline") or {}
            extra = ""
            if x.get("parity"):
                extra = f"; max |d rgb| vs the fp32 CPU oracle on all {x['parity']['rays']} rays {x['parity']['max_abs_rgb']:.2e}"
            if r.get("layer_algorithmic_tflops"):
                extra += f"; {r['layer_algorithmic_tflops']:.1f} TF in units of the fp32 layer's FLOPs"
            out.append(f"* {label}: **{x['value']:.0f} rays/s, {x['ms_per_step']:.3f} ms/step**, dtype {x['dtype']}, dominant kernel "
                       f"{r.get('achieved', 0):.0f} TF = {r.get('frac', 0):.3f} of {r.get('peak', 0):.0f}{extra} (`profiles/{tag}/{name}`)")
    hk = os.path.join(d, "hbm_kernels_sq_counters.json")
    if os.path.exists(hk):
        h = json.load(open(hk)).get("fp32", {})
        parts = []
        for nm in ("stage_prologue_kernel", "encode_features_wave_kernel", "prop_finish_kernel", "nerf_finish_kernel"):
            e = h.get(nm)
            if e and e.get("per_wave"):
                pw = e["per_wave"]
                us = [v for k, v in e.items() if k.startswith("launch_us_under_pmc")]
                parts.append(f"`{nm}` {sum(us) / len(us):.1f} us, {e['SQ_WAVES']:.0f} waves: per wave {4 * pw['SQ_WAVE_CYCLES']:.0f} cycles resident, {pw['SQ_INSTS_VALU']:.0f} vector + "
                             f"{pw['SQ_INSTS_SALU']:.0f} scalar + {pw.get('SQ_INSTS_LDS', 0):.0f} LDS + {pw.get('SQ_INSTS_VMEM_RD', 0) + pw.get('SQ_INSTS_VMEM_WR', 0):.0f} memory instructions, "
                             f"waiting on an operand {100 * e['waiting_fraction_of_wave_cycles']:.0f} %")
        if parts:
            out += ["", "SQ counters of the path's small kernels (`hbm_kernels_sq_counters.json`, separate `--pmc` passes; SQ cycle counters are quad-cycles, x 4 here): " + "; ".join(parts) + ".", ""]
    ts = os.path.join(d, "train_step_bf16.json")
    if os.path.exists(ts):
        t = _bench_line(ts)
        if t:
            out += [f"bf16 training iteration alone (`train_step_bf16.json`, `tools/train_step_bench.py --mlp-dtype bf16`): {t['iteration_ms']} ms (proposal update {t['prop_step_ms']} ms, NeRF update "
                    f"{t['nerf_step_ms']} ms = {t['nerf_step_tflops']} TF); dW of a 1024^2 layer {t['wgrad_ms']} ms ({t['wgrad_nobias_ms']} without the bias gradient), dX {t['dgrad_ms']} ms "
                    f"({t['dgrad_nomask_ms']} without the ReLU mask pass); peak memory {t['peak_mem_gb']} GB; kernel times of the iteration: `rocprofv3_kernel_stats_train_bf16.csv`.", ""]
    return "\n".join(out) + "\n"


def readme_state(tag):
    """The few numbers README.md and INTEGRATION.md quote, as ONE generated paragraph (their `<!-- state:begin <tag> -->` blocks): one number
    per quantity, from profiles/<tag>/bench.json alone - the same file DESIGN.md's block is made from."""
    b = _bench_line(os.path.join(ROOT, "profiles", tag, "bench.json"))
    if not b:
        return f"*(no profiles/{tag}/bench.json)*\n"
    r, nw, p_ = b["roofline"], b.get("named_workloads", {}), b.get("parity", {})
    f_ = b.get("strong_scaling_frame", {})
    parts = [f"*Generated by `python tools/summarize_profiles.py --readme {tag}` from `profiles/{tag}/bench.json` (one default `python bench.py` run on 1 x MI355X; "
             f"boxes of the pool differ by +-2 %).*", "",
             f"fp32 headline (BASELINE configs[1], 4096 rays x 128 samples): **{b['value'] / 1e3:.1f} k rays/s**, {b['ms_per_step']:.2f} ms per step = "
             f"{100 * b['config']['whole_path_tflops'] / r['peak']:.1f} % of the fp32 matrix-core roofline for the whole forward; dominant kernel "
             f"{r['achieved']:.1f} TFLOP/s = {r['frac']:.3f} of peak; max |error| vs the CPU oracle on all {p_.get('rays', '?')} rays "
             f"{p_.get('max_abs_rgb', float('nan')):.1e} (rgb); CPU oracle {b['cpu_baseline']['value']:.0f} rays/s on {b['cpu_baseline']['cores']} threads "
             f"(the unmodified reference: 18-19 rays/s on 8 cores, BASELINE.md)."]
    if f_ and "rays_per_s" in f_:
        parts.append(f"A 1237 x 822 frame at 64 + 128 samples (configs[2]): {f_['seconds_per_frame']:.2f} s = {f_['rays_per_s'] / 1e3:.1f} k rays/s.")
    for k, label in (("c2_bf16", "bf16 MLP, configs[1] shape"), ("c5_bf16", "bf16 MLP, configs[4] per-GPU shape 8192 x 256"), ("c2_bf16x3", "bf16x3 (two bf16 terms per value, inside the fp32 tolerance)")):
        v = nw.get(k)
        if isinstance(v, dict) and "rays_per_s" in v:
            rr = v.get("roofline") or {}
            parts.append(f"{label}: {v['rays_per_s'] / 1e3:.0f} k rays/s, {v['ms_per_step']:.2f} ms per step, dominant kernel {rr.get('frac', 0):.3f} of {rr.get('peak', 0):.0f} TF.")
    for k, label in (("c2_training_iteration", "fp32"), ("c2_training_iteration_bf16", "bf16")):
        v = nw.get(k)
        if isinstance(v, dict) and "iteration_ms" in v:
            parts.append(f"One iteration of `train.py:53-82` at 4096 x 128 in {label}: {v['iteration_ms']:.1f} ms ({v['train_rays_per_s'] / 1e3:.1f} k rays/s).")
    return "\n".join(parts) + "\n"


def sync_docs(tag):
    """Rewrite the generated blocks of DESIGN.md (`profiles:begin`), README.md and INTEGRATION.md (`state:begin`) in place."""
    import re
    for name, marker, gen in (("DESIGN.md", "profiles", markdown), ("README.md", "state", readme_state), ("INTEGRATION.md", "state", readme_state)):
        path = os.path.join(ROOT, name)
        text = open(path).read()
        pat = re.compile(r"<!-- " + marker + r":begin \w+ -->\n.*?<!-- " + marker + r":end -->", re.S)
        if not pat.search(text):
            print(f"{name}: no {marker} block")
            continue
        text = pat.sub(lambda _m: f"<!-- {marker}:begin {tag} -->\n" + gen(tag) + f"<!-- {marker}:end -->", text)
        open(path, "w").write(text)
        print(f"{name}: {marker} block regenerated from profiles/{tag}/")


def sha_at_measurement(src, key, name="kernel_source_sha256_at_measurement.json"):
    """The SHA-256 of the kernel sources as stamped on the GPU box when the counters were collected (tools/gpu_session.sh prof /
    b16pmc).  Never recomputed here: a summary made after the sources changed must not look current to bench.py."""
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(key)


# the reduced-precision workloads of the driver line with counter passes of their own (tools/gpu_session.sh pmc16):
# key -> (rows of the dominant launch, bf16x3?, what the launch is)
WORKLOADS_16 = {"c2_bf16": (4096 * 128, False), "c5_bf16": (8192 * 256, False), "c2_bf16x3": (4096 * 128, True)}


def workload_counters(src, key):
    """Counters of the dominant launches of one reduced-precision workload from three separate --pmc passes over its bench.py command
    (tools/gpu_session.sh pmc16: pmc_<key>_{fetch,write,sq}; round 5's b16pmc directories are read for c2_bf16), or None when those
    passes were not run.  Dominant = the ReLU launches of the one-wave ring kernel that take at least half as long as the longest one:
    the six-layer chain launch in the bf16 mode, the 1024 x 1024 hidden layers in the bf16x3 mode."""
    import collections
    rows, x3 = WORKLOADS_16[key]

    def per_dispatch(d):
        fs = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")
        if not fs:
            return None
        acc = collections.defaultdict(dict)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if "linear_bf16_w16_kernel<1," not in r["Kernel_Name"]:   # ACT = ReLU
                continue
            acc[r["Dispatch_Id"]]["_chain"] = 1.0 if "<1, 128," in r["Kernel_Name"] else 0.0   # the chain launch carries ABL = 128 (its own instantiation)
            acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            acc[r["Dispatch_Id"]]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        xs = list(acc.values())
        if not xs:
            return None
        top = max(x["_ns"] for x in xs)
        return [x for x in xs if x["_ns"] >= 0.5 * top]

    def dirs(kind):
        new = f"pmc_{key}_{kind}"
        return new if glob.glob(f"{src}/{new}/*/*_counter_collection.csv") or key != "c2_bf16" else f"pmc_b16_{kind}"
    S, F, W = per_dispatch(dirs("sq")), per_dispatch(dirs("fetch")), per_dispatch(dirs("write"))
    if not S or not F or not W:
        return None
    mean = lambda xs, k: sum(x[k] for x in xs) / len(xs)  # noqa: E731
    chain = any(x.get("_chain") for x in S)
    fetch_b, write_b = mean(F, "FETCH_SIZE") * 1024 * 2, mean(W, "WRITE_SIZE") * 1024
    g = mean(S, "GRBM_GUI_ACTIVE") / 8
    nlay = 6 if chain else 1
    el = 4 if x3 else 2   # bf16x3 rows are [hi | lo] pairs
    kk = 3 * N_BENCH if x3 else N_BENCH
    stamp = "kernel_source_sha256_at_measurement_b16.json"
    return {"kernel": (f"w16::linear_bf16_w16_kernel<ReLU, CHAIN>: six 1024x1024 layers in one launch, M={rows}" if chain else
                       f"w16::linear_bf16_w16_kernel<ReLU{', X3' if x3 else ''}> 1024x1024, M={rows}"),
            "rows": rows, "bytes_per_launch": round(fetch_b + write_b), "fetch_bytes": round(fetch_b), "write_bytes": round(write_b),
            # rows in, rows out, the weight matrices (the chain's five hidden activations in between never HAVE to leave the die)
            "algorithmic_bytes": 2 * rows * N_BENCH * el + nlay * N_BENCH * kk * 2, "layers_per_launch": nlay,
            "launch_ms_under_pmc": round(mean(S, "_ns") / 1e6, 4), "launches_averaged": len(S), "effective_clock_ghz": round(g / mean(S, "_ns"), 3),
            "mfma_busy_fraction": round(mean(S, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / g, 4),
            "lds_bank_conflict_cycles": mean(S, "SQ_LDS_BANK_CONFLICT"), "lds_instructions": mean(S, "SQ_INSTS_LDS"),
            "sq_counters_mean": {k: mean(S, k) for k in S[0] if not k.startswith("_")},
            "method": f"three separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters) over the workload's bench.py command "
                      f"(tools/gpu_session.sh pmc16: {key}), mean over its dominant launches; bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024",
            "kernel_source_sha256": sha_at_measurement(src, "bf16", stamp)}


def training_gemm_counters(src):
    """Counter traffic of the training GEMMs of one 1024 x 1024 layer on 524 288 rows (tools/gpu_session.sh pmctrain: FETCH_SIZE and WRITE_SIZE passes
    over `tools/train_step_bench.py --gemm-only`, bf16 and fp32) -> traffic.json: training_gemms, or None when those passes were not run."""
    import collections
    pick = {"bf16": {"wgrad": ("linear_tn_bf16_w_kernel",), "dgrad": ("linear_bf16_w16_kernel<0, 0, false, false, false, 0, false, false, false, false, false>",),
                     "relu_mask": ("relu_mask_bf16_kernel",)},
            "fp32": {"wgrad": ("linear_tn_kernel",), "dgrad": ("linear_f32_mfma_persist_kernel",)}}

    def per_kernel(d, counter, names):
        fs = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")
        if not fs:
            return None
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if r["Counter_Name"] != counter:
                continue
            for key, pats in names.items():
                if any(p_ in r["Kernel_Name"] for p_ in pats):
                    vals[key].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        return vals
    out = {}
    for dk, names in pick.items():
        F, W = per_kernel(f"pmc_train_{dk}_fetch", "FETCH_SIZE", names), per_kernel(f"pmc_train_{dk}_write", "WRITE_SIZE", names)
        if not F or not W:
            continue
        out[dk] = {}
        for key in names:
            if not F.get(key) or not W.get(key):
                continue
            # the long launches only (the weight gradient's split reduction and other small launches of the same family are left out)
            top = max(ns for _, ns in F[key])
            f_ = [v for v, ns in F[key] if ns >= 0.5 * top]
            topw = max(ns for _, ns in W[key])
            w_ = [v for v, ns in W[key] if ns >= 0.5 * topw]
            fetch_b, write_b = sum(f_) / len(f_) * 1024 * 2, sum(w_) / len(w_) * 1024
            out[dk][key] = {"bytes_per_launch": round(fetch_b + write_b), "fetch_bytes": round(fetch_b), "write_bytes": round(write_b), "launches_averaged": len(f_),
                            "launch_ms_under_pmc": round(sum(ns for _, ns in F[key] if ns >= 0.5 * top) / len(f_) / 1e6, 4)}
    if not out:
        return None
    out["kernel_source_sha256"] = sha_at_measurement(src, "train", "kernel_source_sha256_at_measurement_train.json")
    out["method"] = ("two separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) over `tools/train_step_bench.py [--mlp-dtype bf16] --gemm-only --iters 3` "
                     "(tools/gpu_session.sh pmctrain), mean over each kernel's long launches; bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024")
    return out


def hbm_kernel_counters(tag):
    """SQ counters of the path's HBM- / latency-bound kernels (tools/gpu_session.sh finpmc: separate --pmc passes over bench.py): per kernel
    the mean launch duration under the counters, waves, and per WAVE: resident cycles, cycles waiting on an instruction's operands, cycles
    issuing, instruction counts by kind -> profiles/<tag>/hbm_kernels_sq_counters.json (what are the 16 / 31 / 40 us made
    of?).  SQ_* cycle counters are summed over the waves of all XCDs; per-wave = / SQ_WAVES."""
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    names = ("stage_prologue_kernel", "encode_features_wave_kernel", "prop_finish_kernel", "nerf_finish_kernel", "norm_partial_from_t_kernel")
    out = {}
    for mode, dirs in (("fp32", ("pmc_fin_a", "pmc_fin_b")), ("bf16", ("pmc_fin_a16",))):
        per = {}
        for d in dirs:
            fs = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")
            if not fs:
                continue
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
                for nm in names:
                    if nm in r["Kernel_Name"]:
                        # full 4096-ray chunks only: the grid of a 4096 x 128 launch
                        acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
                        acc[nm]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                        acc[nm]["_grid"].append(int(r.get("Grid_Size", 0) or 0))
            for nm, c in acc.items():
                e = per.setdefault(nm, {})
                for k, v in c.items():
                    if k == "_grid":
                        continue
                    # a counter row per (dispatch, counter): _ns repeats per counter - the mean is unaffected
                    e[k if k != "_ns" else f"launch_us_under_pmc[{d}]"] = round((sum(v) / len(v)) / (1e3 if k == "_ns" else 1.0), 3)
                e.setdefault("dispatches", {})[d] = len(c["_ns"]) // max(1, len([k for k in c if not k.startswith("_")]))
        for nm, e in per.items():
            w = e.get("SQ_WAVES")
            if w:
                e["per_wave"] = {k: round(e[k] / w, 1) for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
                                                             "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU") if k in e}
                if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_INST_ANY" in e:
                    e["waiting_fraction_of_wave_cycles"] = round(e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 3)
        out[mode] = per
    out["method"] = ("separate rocprofv3 --kernel-trace --pmc passes over `bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0` (fp32: two passes of 8 "
                     "counters; bf16: one), mean over the launches of each kernel; SQ cycle / instruction counters are sums over all waves")
    json.dump(out, open(f"{dst}/hbm_kernels_sq_counters.json", "w"), indent=1)
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--markdown":
        sys.stdout.write(markdown(sys.argv[2]))
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--readme":
        sys.stdout.write(readme_state(sys.argv[2]))
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--sync-docs":
        sync_docs(sys.argv[2])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--hbm-kernels":
        print(json.dumps(hbm_kernel_counters(sys.argv[2])))
        return
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(one(f"{src}/stats/*/*_kernel_stats.csv"), f"{dst}/rocprofv3_kernel_stats_bench.csv")
    for extra in ("bench.json", "bench_under_rocprof.log", "bench_bf16.json", "bench_c5.json", "bench_bf16x3.json", "bench_c5_bf16x3.json", "rocprofv3_kernel_stats_bench_bf16x3.csv", "rocprofv3_kernel_stats_bench_bf16.csv", "bench_c4.json", "bench_c4_bf16.json", "bench_c4_bf16x3.json",
                  "bench_gloo2.json", "pytest_gpu.log", "smoke.log", "nccl2.err", "nccl2.out", "train_step_bf16.json", "train_step_fp32.json", "rocprofv3_kernel_stats_train_bf16.csv",
                  "g20_recipe_in_bf16.json", "g20_recipe_in_fp32.json"):
        if os.path.exists(f"{src}/{extra}"):
            shutil.copy(f"{src}/{extra}", f"{dst}/{extra}")
    for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
        counter_summary(f"{src}/{d}", f"{dst}/rocprofv3_{d}_summary.csv")
    print(step_trace(f"{src}/stats", f"{dst}/last_step_kernel_trace.csv"))
    # the dominant launches alone (ReLU 1024x1024 layers on M = 524288): what bench.py's roofline.avg_launch_ms must agree with
    dur = [v for k, v in durations(f"{src}/stats").items()]
    rows = [r for r in csv.DictReader(open(one(f"{src}/stats/*/*_kernel_trace.csv"))) if "linear_f32_hd_kernel<1" in r["Kernel_Name"]]
    big = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    big = [d for d in big if d > 6e6]
    with open(f"{dst}/dominant_kernel_launches.csv", "w") as f:
        f.write("kernel,shape,launches,mean_ns,min_ns,max_ns,tflops_at_mean\n")
        f.write(f"linear_f32_hd_kernel<ReLU>,M=524288 N=1024 K=1024,{len(big)},{sum(big) / len(big):.0f},{min(big)},{max(big)},"
                f"{2.0 * M_BENCH * N_BENCH * N_BENCH / (sum(big) / len(big)) / 1e3:.1f}\n")
    f, w, s = dominant_counters(f"{src}/pmc_fetch"), dominant_counters(f"{src}/pmc_write"), dominant_counters(f"{src}/pmc_sq")
    fetch_b = f["FETCH_SIZE"][0] * 1024 * 2  # KB -> B; x2: gfx950 tallies the 128-B requests of a wide coalesced stream at 64 B
    write_b = w["WRITE_SIZE"][0] * 1024      # exact for 16-B-per-lane streaming stores
    alg = M_BENCH * N_BENCH * 4 * 2 + N_BENCH * N_BENCH * 4 + N_BENCH * 4
    clk = s["GRBM_GUI_ACTIVE"][0] / 8 / s["GRBM_GUI_ACTIVE"][1]            # GHz (sum over 8 XCDs / ns)
    busy = s["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024 / (s["GRBM_GUI_ACTIVE"][0] / 8)
    info = {"linear_f32_mfma_1024x1024_bytes_per_launch": round(fetch_b + write_b), "fetch_bytes": round(fetch_b),
            "write_bytes": round(write_b), "algorithmic_bytes": alg, "launch_ms_under_pmc": round(f["FETCH_SIZE"][1] / 1e6, 3),
            "launches_averaged": f["FETCH_SIZE"][2], "effective_clock_ghz": round(clk, 3),
            "mfma_busy_fraction": round(busy, 4), "source": f"profiles/{tag}/rocprofv3_pmc_*_summary.csv",
            "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 3 --warmup 1 "
                      "--cpu-rays 0`, mean over the ReLU 1024x1024 launches; bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 "
                      "(gfx950 corrections of MI355X_MICROARCH.md, HBM section); clock = GRBM_GUI_ACTIVE/8/duration; "
                      "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE/8)",
            "sq_counters_mean": {k: v[0] for k, v in s.items()},
            # bench.py trusts this file only while the kernel sources it was measured on are unchanged (stamped at measurement time)
            "kernel_source_sha256": sha_at_measurement(src, "fp32")}
    info["by_workload"] = {}
    for key in WORKLOADS_16:
        wc = workload_counters(src, key)
        if wc:
            info["by_workload"][key] = wc
            json.dump(wc, open(f"{dst}/rocprofv3_pmc_{key}_summary.json", "w"), indent=1)
    tg = training_gemm_counters(src)
    if tg:
        info["training_gemms"] = tg
        json.dump(tg, open(f"{dst}/rocprofv3_pmc_training_gemms_summary.json", "w"), indent=1)
    if "c2_bf16" in info["by_workload"]:
        info["bf16_ring_kernel"] = info["by_workload"]["c2_bf16"]   # (the key rounds 4-5 used)
    json.dump(info, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({k: info[k] for k in list(info)[:9]}, indent=1))


if __name__ == "__main__":
    main()
