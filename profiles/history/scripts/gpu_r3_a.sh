#!/bin/bash
# round-3 GPU call A: parity after the epilogue rewrite, A/B against the round-2 GEMM, default bench with --also, rocprofv3 stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
O=gpurun_out/a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/ab_lib.py run r2,base 32768,4096,1024,1,197 32896,3072,1024,0,198 32768,1024,1024,0,197 32768,1024,4096,0,197 \
  2688,27648,5120,2,197 2688,15360,5120,0,198 2688,5120,13824,0,198 2688,5120,5120,0,198 2816,27648,5120,2,197 8192,8192,8192,0,197 > $O/ab_epilogue.jsonl 2> $O/ab_err.txt
cat $O/ab_epilogue.jsonl; tail -3 $O/ab_err.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_err.txt
head -c 3000 $O/bench_default.json; echo; tail -3 $O/bench_err.txt
timeout 300 python tools/vit_time.py 128 > $O/vit_128.json 2>> $O/vit_err.txt
VALLEY_VIT_CHUNK=64 VALLEY_ROW_SPLIT_MIN=16384 timeout 300 python tools/vit_time.py 128 > $O/vit_128_chunk64_split.json 2>> $O/vit_err.txt
VALLEY_VIT_CHUNK=64 timeout 300 python tools/vit_time.py 128 > $O/vit_128_chunk64.json 2>> $O/vit_err.txt
cat $O/vit_128*.json; tail -3 $O/vit_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --also none > $R/$O/bench_c3_prof.json 2> $R/$O/prof.err
cd $R
find $O/prof_c3 -name "*kernel_trace.csv" -delete; find $O/prof_c3 -name "*.db" -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/a/prof_c3/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if float(r["Percentage"]) > 0.3:
        print(n[:90].replace("void (anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
