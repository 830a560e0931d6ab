#!/usr/bin/env python3
"""Reads the kernel trace of tools/diag/launch_gap_probe2.hip.   python tools/diag/launch_gap_probe2.py <..._kernel_trace.csv>"""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "work" in r["Kernel_Name"] or "fat" in r["Kernel_Name"]]
names = ["small-LDS long -> tiny (1 workgroup)", "small-LDS long -> small-LDS 2048 workgroups", "small-LDS long -> 64 KB-LDS 2048 workgroups",
         "160 KB-LDS long -> small-LDS 2048 workgroups", "160 KB-LDS long -> 160 KB-LDS 256 workgroups", "small-LDS long -> 200-live-register kernel"]
i = 0
print("stream | pair | first kernel (us) | gap first.end -> second.start (us, median of 12) | second kernel (us) | gap second.end -> next first.start")
for stream in ("null stream", "non-blocking stream"):
    for w in range(6):
        g, b, d1, d2 = [], [], [], []
        for rep in range(12):
            a, c = rows[i], rows[i + 1]
            g.append((int(c["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
            d1.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
            d2.append((int(c["End_Timestamp"]) - int(c["Start_Timestamp"])) / 1e3)
            if rep < 11:
                b.append((int(rows[i + 2]["Start_Timestamp"]) - int(c["End_Timestamp"])) / 1e3)
            i += 2
        print(f"{stream:20s} | {names[w]:48s} | {statistics.median(d1):7.1f} | {statistics.median(g):6.2f} | {statistics.median(d2):7.1f} | {statistics.median(b):6.2f}")
