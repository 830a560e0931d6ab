#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace csv: per kernel name, mean duration, and for the ReLU-mask launches how much of each lies inside a weight-gradient launch."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
dur = defaultdict(list)
for n, s, e in ks:
    dur[n.split("(")[0][-60:]].append(e - s)
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{n:60s} n={len(d):4d} mean {sum(d) / len(d) / 1e3:9.1f} us  min {min(d) / 1e3:9.1f}")
wg = [(s, e) for n, s, e in ks if "linear_tn_bf16_w_kernel" in n]
mk = [(s, e) for n, s, e in ks if "relu_mask_bf16" in n and e - s > 200000]
inside = []
for s, e in mk:
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in wg)
    inside.append(ov / (e - s))
if inside:
    print("big mask launches:", len(mk), "mean fraction inside a weight-gradient launch:", sum(inside) / len(inside))
