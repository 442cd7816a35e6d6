#!/bin/bash
# round-2 GPU call I: full suite on the final build, default bench (live traffic + cpu baseline), rocprofv3 stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -rP 2>&1 | grep -v "^$" > gpurun_out/i_pytest_full.log
tail -4 gpurun_out/i_pytest_full.log; grep -n "c3 end to end\|c4 tower\|FAILED\|Error" gpurun_out/i_pytest_full.log | head
timeout 1200 python bench.py > gpurun_out/i_bench_c3_default.json 2> gpurun_out/i_err1.txt
head -c 1400 gpurun_out/i_bench_c3_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/i_prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none > $R/gpurun_out/i_bench_c3_prof.json 2> $R/gpurun_out/i_prof.err
cd $R
find gpurun_out/i_prof_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/i_prof_c3 -name "*.db" -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/i_prof_c3/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("gemm", "norm", "attn", "rope", "skinny")):
        print(n[:80].replace("void (anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
