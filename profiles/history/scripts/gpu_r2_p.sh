#!/bin/bash
# round-2 GPU call P: full suite on the build with the 4-wave tiles, default bench (live traffic + cpu baseline), rocprofv3 stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -rP 2>&1 | grep -v "^$" > gpurun_out/p_pytest_full.log
tail -4 gpurun_out/p_pytest_full.log; grep -n "c3 end to end\|c4 tower\|FAILED\|Error" gpurun_out/p_pytest_full.log | head
timeout 1200 python bench.py > gpurun_out/p_bench_c3_default.json 2> gpurun_out/p_err1.txt
head -c 1400 gpurun_out/p_bench_c3_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none > $R/gpurun_out/p_bench_c3_prof.json 2> $R/gpurun_out/p_prof.err
cd $R
python - <<'PY'
# idle time between consecutive kernels (launch gaps) over the profiled run
import csv, glob
f = glob.glob("gpurun_out/p_prof_c3/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 3):i + 1]) for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 < g < 100_000]
print("kernels", len(rows), "busy_ms", busy / 1e6, "gaps<100us: n", len(small), "sum_ms", sum(small) / 1e6, "median_us", sorted(small)[len(small) // 2] / 1e3 if small else None,
      "overlap(neg) n", sum(1 for g in gaps if g <= 0))
import collections
by = collections.Counter()
for i, g in enumerate(gaps):
    if 0 < g < 100_000:
        by[rows[i + 1][2][:60]] += g
for k, v in by.most_common(8):
    print("  gap before", k, round(v / 1e6, 2), "ms")
PY
find gpurun_out/p_prof_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/p_prof_c3 -name "*.db" -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/p_prof_c3/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("gemm", "norm", "attn", "rope", "skinny")):
        print(n[:80].replace("void (anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
