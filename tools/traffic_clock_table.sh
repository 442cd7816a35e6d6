#!/bin/bash
# does cutting the dominant GEMM's 3.4x fetch amplification buy clock?  13B gate|up (2688 x 27648 x 5120, SwiGLU,
# persistent 4-wave kernel, hint 197) under tile-group heights {default, 1 = n-fastest, 11 = m-fastest} x weight layout {row-major,
# block-ordered}: wall time, TFLOP/s, sustained clock (GRBM_GUI_ACTIVE / 8 XCDs / wall) and L2<->fabric fetch (FETCH_SIZE x 2 x 1024,
# gfx950) per launch — ONE rocprofv3 pass per arm (two counters, kernel trace only).   -> gpurun_out/traffic_clock.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; out=$R/gpurun_out/traffic_clock; mkdir -p $out; : > $R/gpurun_out/traffic_clock.txt
cd /tmp; export TMPDIR=/tmp
for gm in 0 1 11 4; do for packed in 0 1; do
  tag=gm${gm}_p${packed}
  VLY_TILE_GM=$gm rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $out -o $tag -- python $R/tools/gemm_one.py 2688 27648 5120 tile 197 2 8 $packed > /dev/null 2> $out/$tag.err
  python - $out $tag $gm $packed <<'PY' >> $R/gpurun_out/traffic_clock.txt
import csv, glob, sys
out, tag, gm, packed = sys.argv[1:5]
c = {}
for f in glob.glob(f"{out}/**/{tag}_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_p4" in r["Kernel_Name"]:
            c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
d = []
for f in glob.glob(f"{out}/**/{tag}_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_p4" in r["Kernel_Name"]:
            d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2]
if d and c:
    us = med(d) / 1e3
    print(f"gm {gm:>2s} packed {packed}: {us:7.1f} us  {2 * 2688 * 27648 * 5120 / us / 1e6:7.1f} TFLOP/s  clock {med(c['GRBM_GUI_ACTIVE']) / 8 / us / 1e3:5.2f} GHz  "
          f"fetch {med(c['FETCH_SIZE']) * 2 * 1024 / 1e6:7.1f} MB per launch (algorithmic A + W 310.6 MB)")
else:
    print(f"gm {gm} packed {packed}: no data")
PY
done; done
cat $R/gpurun_out/traffic_clock.txt
