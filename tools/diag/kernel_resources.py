#!/usr/bin/env python3
"""Register / scratch use of the kernels in libm360.so (from the code objects' metadata notes): `python tools/diag/kernel_resources.py [substr]`.
A kernel of the hot path must show private_segment_fixed_size 0 (no scratch) unless DESIGN.md says otherwise."""
import os, re, subprocess, sys, tempfile
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mipnerf360_amd", "libm360.so")
data = open(so, "rb").read()
rows = []
with tempfile.TemporaryDirectory() as d:
    for n, m in enumerate(list(re.finditer(b"\x7fELF", data))[1:]):
        f = os.path.join(d, f"co{n}.elf")
        open(f, "wb").write(data[m.start():])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
        cur = {}
        for line in out.splitlines():
            line = line.strip()
            for key in (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count", ".group_segment_fixed_size"):
                if line.startswith(key + ":"):
                    if key == ".name" and cur.get(".name"):
                        rows.append(cur); cur = {}
                    cur[key] = line.split(":", 1)[1].strip()
        if cur.get(".name"):
            rows.append(cur)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    name = subprocess.run(["c++filt", r[".name"]], capture_output=True, text=True).stdout.strip()
    if pat in name:
        print(f"vgpr {str(r.get('.vgpr_count')):>4} agpr {str(r.get('.agpr_count')):>4} sgpr {str(r.get('.sgpr_count')):>4} scratch {str(r.get('.private_segment_fixed_size')):>5} "
              f"spill {str(r.get('.vgpr_spill_count')):>4} lds {str(r.get('.group_segment_fixed_size')):>7}  {name[:170]}")
