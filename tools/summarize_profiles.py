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
    hits = glob.glob(pattern)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[0]


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
    starts = [i for i, n in enumerate(names) if "sample_t_kernel" in n]
    ends = [min(j for j, n in enumerate(names) if "t_to_s_kernel" in n and j > i) for i in starts]
    spans = [int(rows[e]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"]) for s, e in zip(starts, ends)]
    full = [k for k, sp in enumerate(spans) if sp > 0.8 * max(spans)]  # the 4096-ray bench steps (not the 1024-ray parity pass)
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


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(one(f"{src}/stats/*/*_kernel_stats.csv"), f"{dst}/rocprofv3_kernel_stats_bench.csv")
    for extra in ("bench.json", "bench_under_rocprof.log"):
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
            # bench.py trusts this file only while the kernel sources it was measured on are unchanged
            "kernel_source_sha256": kernel_source_sha()}
    json.dump(info, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({k: info[k] for k in list(info)[:9]}, indent=1))


if __name__ == "__main__":
    main()
