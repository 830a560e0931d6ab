#!/usr/bin/env python3
"""Reads the kernel trace of tools/diag/launch_gap_probe.hip: gap between the end of writer<MODE> (n bytes) and the start of the tiny kernel behind it.
    python tools/diag/launch_gap_probe.py <..._kernel_trace.csv>"""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "writer" in r["Kernel_Name"] or "tiny" in r["Kernel_Name"]]
sizes = [64, 4096, 64 << 10, 1 << 20, 2 << 20, 8 << 20, 32 << 20, 128 << 20]
modes = ["plain stores", "nt stores", "sc0 sc1 stores", "loads only"]
i = 0
print("bytes written by the first kernel | mode | median gap to the dependent tiny kernel behind it (us) | median gap tiny -> next writer (us) | writer duration (us)")
for b in sizes:
    for m in modes:
        gaps, back, dur = [], [], []
        for rep in range(20):
            w, t = rows[i], rows[i + 1]
            assert "writer" in w["Kernel_Name"] and "tiny" in t["Kernel_Name"], (w["Kernel_Name"], t["Kernel_Name"])
            gaps.append((int(t["Start_Timestamp"]) - int(w["End_Timestamp"])) / 1e3)
            dur.append((int(w["End_Timestamp"]) - int(w["Start_Timestamp"])) / 1e3)
            if i + 2 < len(rows) and rep < 19:
                back.append((int(rows[i + 2]["Start_Timestamp"]) - int(t["End_Timestamp"])) / 1e3)
            i += 2
        print(f"{b:>10d} | {m:14s} | {statistics.median(gaps):6.2f} | {statistics.median(back):6.2f} | {statistics.median(dur):8.2f}")
