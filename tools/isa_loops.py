#!/usr/bin/env python3
"""List the loops of one kernel in a hipcc -save-temps .s file with their instruction mix:
isa_loops.py file.s <substring of the mangled kernel name>.  (MFMA / ds_read / LDS-DMA / scratch counts, waitcnts, barriers.)"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    if key not in m.group(1):
        continue
    body = m.group(2).split('\n')
    print(m.group(1)[:110], len(body), "lines")
    labels = {}
    for n, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = n
    for n, l in enumerate(body):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            seg = body[labels[mm.group(1)]:n]
            cnt = lambda k: sum(1 for x in seg if k in x)
            print("  loop", mm.group(1), labels[mm.group(1)], n, "mfma", cnt("v_mfma"), "ds_read", cnt("ds_read"), "ldsdma", cnt("global_load_lds") + sum(1 for x in seg if "buffer_load" in x and " lds" in x),
                  "scratch", cnt("scratch_"), "barrier", cnt("s_barrier"), "accvgpr_mov", cnt("v_accvgpr"),
                  "waits", [x.split()[1] + " " + " ".join(x.split()[2:3]) for x in seg if "s_waitcnt" in x][:12])
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(body))
