#!/bin/bash
# round-3 GPU call E: PMC of the attention kernels, full suite on the cleaned-up GEMM source, default bench + rocprofv3 stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/e
mkdir -p $O
bash tools/pmc_cmd.sh vitattn_v1 vit_attn python $R/tools/vit_attn_time.py 128 > /dev/null 2>&1; cp gpurun_out/pmc_vitattn_v1/summary.txt $O/pmc_vit_attn_v1.txt
VLY_VIT_ATTN=2 bash tools/pmc_cmd.sh vitattn_v2 vit_attn python $R/tools/vit_attn_time.py 128 > /dev/null 2>&1; cp gpurun_out/pmc_vitattn_v2/summary.txt $O/pmc_vit_attn_v2.txt
bash tools/pmc_cmd.sh llattn llama_attn python $R/tools/attn_one.py llama 8 336 40 > /dev/null 2>&1; cp gpurun_out/pmc_llattn/summary.txt $O/pmc_llama_attn.txt
cat $O/pmc_vit_attn_v1.txt $O/pmc_vit_attn_v2.txt $O/pmc_llama_attn.txt
tail -5 gpurun_out/pmc_vitattn_v1/p1.err
rm -rf gpurun_out/pmc_vitattn_v1 gpurun_out/pmc_vitattn_v2 gpurun_out/pmc_llattn
timeout 1800 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_err.txt
head -c 1200 $O/bench_default.json; echo; tail -3 $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --also none > $R/$O/bench_c3_prof.json 2> $R/$O/prof.err
cd $R
find $O/prof_c3 -name "*kernel_trace.csv" -delete; find $O/prof_c3 -name "*.db" -delete
