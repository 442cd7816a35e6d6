#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / LDS report for one HIP source: kernel_resources.py valley_amd/csrc/x.hip [filter [-Dflags ...]]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]                                        # further compiler flags (-DVLY_...)
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        cur = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in sorted(rows.items()):
    if flt in k:
        print(f"{k:60s} vgpr={v.get('VGPRs'):>4} agpr={v.get('AGPRs'):>4} scratch={v.get('ScratchSize [bytes/lane]'):>4} spill={v.get('VGPRs Spill'):>3} "
              f"lds={v.get('LDS Size [bytes/block]'):>7} occ={v.get('Occupancy [waves/SIMD]')}")
