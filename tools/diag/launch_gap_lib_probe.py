#!/usr/bin/env python3
"""Gaps between consecutive dispatches of a rocprofv3 kernel trace (any program).   python tools/diag/launch_gap_lib_probe.py <..._kernel_trace.csv> [substring]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if sub in r["Kernel_Name"]:
        print(f"gap {((s - prev) / 1e3 if prev else 0):7.2f} us | {(e - s) / 1e3:8.1f} us | {r['Kernel_Name'][:70]}")
    prev = e
