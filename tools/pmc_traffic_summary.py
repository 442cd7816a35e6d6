#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into per-kernel, per-launch HBM
bytes.  Units and gfx950 correction per MI355X_MICROARCH.md §HBM: both counters are in KiB-like
kilobyte units (x1024); FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read on
gfx950, so it is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import json
import os
import re
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(d, f"{c}_counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            name = re.sub(r">\(.*", ">", name) if ">(" in name else re.sub(r"\(.*", "", name)
            agg[name][c].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    if "at::native" in k or "rocclr" in k:
        continue
    fe, wr = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    out[k] = {"launches": len(fe), "fetch_bytes_per_launch": round(2 * 1024 * sum(fe) / max(len(fe), 1)),
              "write_bytes_per_launch": round(1024 * sum(wr) / max(len(wr), 1))}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), x1024",
           "kernels": out}, open(os.path.join(d, "traffic.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(k[:70], v)
