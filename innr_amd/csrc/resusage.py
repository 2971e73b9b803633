#!/usr/bin/env python3
"""Compact view of hipcc -Rpass-analysis=kernel-resource-usage output (innr_amd/lib/asm/resource_usage.txt)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "../lib/asm/resource_usage.txt"
rows, cur = [], None
for line in open(path):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None:
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, name in zip(rows, names):
    name = re.sub(r"\(.*", "", name).replace("innr::", "").replace("void ", "")
    print(f"{name[:64]:64s} VGPR={r.get('VGPRs', '?'):>4} AGPR={r.get('AGPRs', '?'):>4} SGPR={r.get('TotalSGPRs', '?'):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]', '?'):>5} occ={r.get('Occupancy [waves/SIMD]', '?'):>2} "
          f"LDS={r.get('LDS Size [bytes/block]', '?')}")
