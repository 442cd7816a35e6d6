#!/bin/bash
# The final artifacts of a round, from ONE GPU lease:  usage  gpurun -- 'bash tools/refresh_profiles.sh <tag>'   (e.g. v1)
#   gpurun_out/final_<tag>/gpu_suite.txt            tail of `pytest tests -m gpu`
#   .../bench_default.json                          the driver's command: python bench.py --steps 20 --warmup 5
#   .../c3_kernel_stats.csv + bench_c3_under_rocprof.json     rocprofv3 --kernel-trace --stats of the c3 step
#   .../decode_kernel_stats.csv                     the same for the hipGraph decode step
#   .../pmc_*.txt                                   PMC counters of the dominant GEMMs (tools/pmc_gemm.sh), separate passes
# Copy what should be judged into profiles/r0N/ (this directory is scratch).
TAG=${1:-v1}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/final_$TAG; mkdir -p $O
export TMPDIR=/tmp
if [ "${REFRESH_SKIP_TESTS:-0}" != 1 ]; then
  ( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/gpu_suite.txt 2>&1
  tail -3 $O/gpu_suite.txt
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --also none > $O/bench_c3_under_rocprof.json 2>> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -o dec -- python $R/bench.py --config c5 --decode 64 --warmup 4 --no-cpu-baseline --traffic none --also none > $O/bench_decode_under_rocprof.json 2>> $O/bench_default.err
cd $R
find $O/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c3_kernel_stats.csv
find $O/prof_decode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/decode_kernel_stats.csv
rm -rf $O/prof_c3 $O/prof_decode
if [ "${REFRESH_PMC:-1}" = 1 ]; then        # PMC counters of the two dominant GEMM call sites (one rocprofv3 pass per counter group)
  bash tools/pmc_gemm.sh swiglu_13b 2688 27648 5120 tile 197 2 > /dev/null 2>&1 && cp gpurun_out/pmc_swiglu_13b/summary.txt $O/pmc_swiglu_13b.txt
  bash tools/pmc_gemm.sh vit_fc1 32768 4096 1024 tile 197 1 > /dev/null 2>&1 && cp gpurun_out/pmc_vit_fc1/summary.txt $O/pmc_vit_fc1.txt
fi
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('c3', d['value'], d['ms_per_step'], d['stages']['vit_frac_of_bf16_peak'], d['stages']['prefill_frac_of_bf16_peak'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic_over_algorithmic'))
for k,v in d.get('also',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
head -12 $O/c3_kernel_stats.csv | cut -c1-140; head -8 $O/decode_kernel_stats.csv | cut -c1-140
