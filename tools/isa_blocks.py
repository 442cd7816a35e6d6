#!/usr/bin/env python3
"""Basic-block instruction histogram of one kernel in a hipcc -S listing (epilogue / K-loop instruction diets).
usage: isa_blocks.py file.s KERNEL_SUBSTRING [min_block_size]"""
import collections
import re
import sys


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:\s*(;.*)?$", l)]
    for k, (i, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        yield name, lines[i:end]


def main():
    path, key = sys.argv[1], sys.argv[2]
    mins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    meta = {}
    cur = None
    for l in lines:                                   # resource metadata
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1)
        m = re.match(r"\s+\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", l)
        if m and cur:
            meta.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    for name, body in kernels(lines):
        if key not in name:
            continue
        print("==", name[:110])
        blocks, cb = [], None
        for l in body:
            m = re.match(r"^(\.LBB\S+):", l)
            if m:
                cb = [m.group(1), collections.Counter(), 0]
                blocks.append(cb)
                continue
            t = l.strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            if cb is None:
                cb = ["entry", collections.Counter(), 0]
                blocks.append(cb)
            cb[1][t.split()[0]] += 1
            cb[2] += 1
        tot = collections.Counter()
        for b in blocks:
            tot.update(b[1])
            if b[2] >= mins:
                print(f"  {b[0]:12s} {b[2]:4d}", dict(b[1].most_common(9)))
        print("  total", sum(tot.values()), {k: v for k, v in tot.most_common(14)})
    for k, v in meta.items():
        if key in k:
            print(k[:100], v)


if __name__ == "__main__":
    main()
